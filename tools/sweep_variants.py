"""Sweep scatter variants / B / remap at one N (development aid)."""
import sys
import time
from pathlib import Path

import ctypes
import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vkradixsort_amd as vrs  # noqa: E402
from vkradixsort_amd import capi  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 8
    combos = sys.argv[2].split(",") if len(sys.argv) > 2 else ["32:0"]
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    keys = np.random.RandomState(1).randint(0, 2 ** 32, size=n, dtype=np.uint32)
    ref = None
    with vrs.GPUContext(0) as gpu:
        mm = ctypes.c_uint64(0)
        gpu.check(gpu.lib.vrs_debug_atomic_rank_selftest(gpu.handle, 4096, 12345, ctypes.byref(mm)))
        print("atomic rank selftest mismatches:", mm.value, flush=True)
        src = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, vrs.Buffer.BufferSettings(4 * n), keys)
        for combo in combos:
            parts = [int(x) for x in combo.split(":")]
            parts += [0, 1, 0, 1][len(parts) - 1:]
            B, variant, fused, wgs, remap = parts[:5]
            persistent = fused
            gpu.setTuning(capi.VRS_TUNE_FUSED_PREFIX, fused)
            gpu.setTuning(capi.VRS_TUNE_SCATTER_VARIANT, variant)
            gpu.setTuning(capi.VRS_TUNE_XCD_REMAP, remap)
            m = vrs.MultiRadixSort(NUM_BLOCKS_PER_WORKGROUP=B, keys=keys, quiet=True)
            m.setup(gpu)
            times = []
            try:
                for r in range(reps + 2):
                    m.m_buffers[0].copyFrom(src)
                    gpu.waitIdle()
                    if r == 2:
                        gpu.profileReset()
                        gpu.profileEnable(True)
                    t0 = time.perf_counter()
                    m.enqueueSort()
                    gpu.waitIdle()
                    if r >= 2:
                        times.append(time.perf_counter() - t0)
            except vrs.VrsError as e:
                print(combo, "ERROR", e)
                gpu.profileEnable(False)
                continue
            gpu.profileEnable(False)
            out = m.download()
            if ref is None:
                ref = np.sort(keys)
            ok = bool(np.array_equal(out, ref))
            t = min(times)
            line = f"N={n} B={B} variant={variant} fusedprefix={persistent} wgs={wgs} remap={remap} exact={ok} min={t*1e3:.3f}ms {n/t/1e9:.2f} Gkeys/s {48*n/t/8e12*100:.1f}%roof"
            for kid, name in capi.KERNEL_NAMES.items():
                cnt, ms = gpu.profileQuery(kid)
                if cnt:
                    line += f" | {name}: {ms/cnt*1e3:.1f}us"
            print(line, flush=True)
            m.releaseBuffers()
            m.m_pass.release()
            while gpu.getActiveIndex() != 0:
                gpu.incrementActiveIndex()


if __name__ == "__main__":
    main()
