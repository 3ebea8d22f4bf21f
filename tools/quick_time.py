"""Quick per-kernel timing of the four-pass sort (development aid; bench.py is the contract bench)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vkradixsort_amd as vrs  # noqa: E402
from vkradixsort_amd import capi  # noqa: E402


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 8
    Bs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [32]
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    remaps = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1]
    keys = np.random.RandomState(1).randint(0, 2 ** 32, size=n, dtype=np.uint32)
    with vrs.GPUContext(0) as gpu:
        print(gpu.deviceInfo())
        for B in Bs:
            for remap in remaps:
                gpu.setTuning(capi.VRS_TUNE_XCD_REMAP, remap)
                m = vrs.MultiRadixSort(NUM_BLOCKS_PER_WORKGROUP=B, keys=keys, quiet=True)
                m.setup(gpu)
                src = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, vrs.Buffer.BufferSettings(4 * n), keys)
                times = []
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
                gpu.profileEnable(False)
                out = m.download()
                ok = bool(np.all(out[1:] >= out[:-1]))
                t = min(times)
                line = f"N={n} B={B} remap={remap} W={m.m_pass.m_pushConstants.g_num_workgroups} sorted={ok} min={t*1e3:.3f}ms med={np.median(times)*1e3:.3f}ms {n/t/1e9:.2f} Gkeys/s {48*n/t/1e12:.3f} TB/s(48B/key)"
                for kid, name in capi.KERNEL_NAMES.items():
                    cnt, ms = gpu.profileQuery(kid)
                    if cnt:
                        line += f" | {name}: {ms/cnt*1e3:.1f}us x{cnt//reps}"
                print(line, flush=True)
                src.release()
                m.releaseBuffers()
                m.m_pass.release()


if __name__ == "__main__":
    main()
