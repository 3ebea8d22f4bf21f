"""Hybrid form vs LSD form of vrs_sort_keys_u32 by size, bench conditions (K pre-staged batches back to back, no events):
   python tools/lab/hybrid_sweep.py [N ...]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import vkradixsort_amd as vrs  # noqa: E402
from vkradixsort_amd import capi  # noqa: E402

sizes = [int(float(a)) for a in sys.argv[1:]] or [5 * 10 ** 6, 10 ** 7, 2 * 10 ** 7, 3 * 10 ** 7, 4 * 10 ** 7]
S = vrs.Buffer.BufferSettings
with vrs.GPUContext(0) as gpu:
    for n in sizes:
        K = max(8, min(48, int(4e8 // n)))
        keys = np.random.RandomState(1).randint(0, 2 ** 32, size=n, dtype=np.uint32)
        src = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), keys)
        batches = [vrs.Buffer(gpu, S(4 * n)) for _ in range(K)]
        tmp = vrs.Buffer(gpu, S(4 * n))
        res = {}
        for name, hyb in (("lsd", 0), ("hybrid", 1)):
            gpu.setTuning(capi.VRS_TUNE_HYBRID, hyb)
            gpu.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 1 << 22)
            best = 1e9
            for rep in range(4):
                for b in batches:
                    b.copyFrom(src)
                gpu.waitIdle()
                t0 = time.perf_counter()
                for b in batches:
                    gpu.check(gpu.lib.vrs_sort_keys_u32(gpu.handle, b.handle, tmp.handle, n))
                gpu.waitIdle()
                if rep:
                    best = min(best, (time.perf_counter() - t0) / K)
            out = np.empty(n, np.uint32)
            batches[-1].downloadWithStagingBuffer(out)
            res[name] = (best * 1e3, bool(np.array_equal(out, np.sort(keys))))
        gpu.setTuning(capi.VRS_TUNE_HYBRID, 1)
        gpu.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, capi.HYBRID_MIN_KEYS_DEFAULT)
        print(f"N={n:>10d} K={K:2d}  lsd {res['lsd'][0]:.4f} ms ({n / res['lsd'][0] / 1e6:.1f} Gkeys/s, exact={res['lsd'][1]})   "
              f"hybrid {res['hybrid'][0]:.4f} ms ({n / res['hybrid'][0] / 1e6:.1f} Gkeys/s, exact={res['hybrid'][1]})", flush=True)
        for b in batches + [src, tmp]:
            b.release()
