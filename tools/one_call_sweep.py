"""N sweep of the one-call sort: contract passes vs the one-read form (wall time per sort, keys resident, no events).
   python tools/one_call_sweep.py > profiles/r01_one_call_crossover.csv"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vkradixsort_amd as vrs  # noqa: E402
from vkradixsort_amd import capi  # noqa: E402


def main():
    sizes = [int(float(x)) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else
                                     "1e5,2e5,5e5,1e6,2e6,5e6,1e7,2e7,5e7,1e8,2e8".split(","))]
    print("n,contract_ms,one_read_ms,contract_gkeys_s,one_read_gkeys_s,speedup")
    with vrs.GPUContext(0) as gpu:
        lib = gpu.lib
        for n in sizes:
            keys = np.random.RandomState(1).randint(0, 2 ** 32, size=n, dtype=np.uint32)
            S = vrs.Buffer.BufferSettings(4 * n)
            src = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S, keys)
            k0, k1 = vrs.Buffer(gpu, S), vrs.Buffer(gpu, S)
            res = []
            for min_keys in (0, 1):
                gpu.setTuning(capi.VRS_TUNE_ONE_CALL_MIN_KEYS, min_keys)
                best = 1e9
                for r in range(12):
                    k0.copyFrom(src)
                    gpu.waitIdle()
                    t0 = time.perf_counter()
                    gpu.check(lib.vrs_sort_keys_u32(gpu.handle, k0.handle, k1.handle, n))
                    gpu.waitIdle()
                    if r >= 2:
                        best = min(best, time.perf_counter() - t0)
                res.append(best)
            out = np.empty(n, np.uint32)
            k0.downloadWithStagingBuffer(out)
            assert np.all(out[1:] >= out[:-1])
            print(f"{n},{res[0]*1e3:.4f},{res[1]*1e3:.4f},{n/res[0]/1e9:.2f},{n/res[1]/1e9:.2f},{res[0]/res[1]:.3f}", flush=True)
            for b in (src, k0, k1):
                b.release()
        gpu.setTuning(capi.VRS_TUNE_ONE_CALL_MIN_KEYS, capi.ONE_CALL_MIN_KEYS_DEFAULT)


if __name__ == "__main__":
    main()
