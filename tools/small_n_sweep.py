"""Crossover of vrs_sort_keys_u32's single-launch form (one single_radixsort workgroup) against the multi-block passes
at small N -> CSV (profiles/r02_small_n_crossover.csv).  usage: small_n_sweep.py [out.csv]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vkradixsort_amd as vrs  # noqa: E402
from vkradixsort_amd import capi  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    sizes = [100, 500, 1000, 2000, 3000, 4096, 5000, 6000, 8000, 10000, 15000, 20000, 50000, 100000]
    lines = ["n,single_launch_us,multi_block_us,faster"]
    with vrs.GPUContext(0) as gpu:
        S = vrs.Buffer.BufferSettings
        for n in sizes:
            keys = np.random.RandomState(n).randint(0, 2 ** 32, size=n, dtype=np.uint32)
            src = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), keys)
            k0, k1 = vrs.Buffer(gpu, S(4 * n)), vrs.Buffer(gpu, S(4 * n))
            res = {}
            for name, thr in (("single", 1 << 30), ("multi", 0)):
                gpu.setTuning(capi.VRS_TUNE_SINGLE_MAX_KEYS, thr)
                ts = []
                for r in range(60):
                    k0.copyFrom(src)
                    gpu.waitIdle()
                    t0 = time.perf_counter()
                    gpu.check(gpu.lib.vrs_sort_keys_u32(gpu.handle, k0.handle, k1.handle, n))
                    gpu.waitIdle()
                    ts.append(time.perf_counter() - t0)
                res[name] = float(np.median(ts[10:])) * 1e6
                o = np.empty(n, np.uint32)
                k0.downloadWithStagingBuffer(o)
                assert np.array_equal(o, np.sort(keys)), (name, n)
            gpu.setTuning(capi.VRS_TUNE_SINGLE_MAX_KEYS, 4096)
            lines.append(f"{n},{res['single']:.1f},{res['multi']:.1f},{'single' if res['single'] < res['multi'] else 'multi'}")
            for b in (src, k0, k1):
                b.release()
    text = "\n".join(lines) + "\n"
    print(text)
    if out:
        Path(out).write_text(text)


if __name__ == "__main__":
    main()
