"""Lab: key + payload pairs by size -- the LSD passes, the counted hybrid form and the stable pool form side by side (back to back,
K batches, ms per sort).   python tools/lab/pairs_crossover.py"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import vkradixsort_amd as vrs  # noqa: E402
from vkradixsort_amd import capi  # noqa: E402

S = vrs.Buffer.BufferSettings


def timed(gpu, n, K=8, reps=3):
    keys = np.random.RandomState(1).randint(0, 2 ** 32, size=n, dtype=np.uint32)
    src = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), keys)
    iota = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), np.arange(n, dtype=np.uint32))
    ks = [vrs.Buffer(gpu, S(4 * n)) for _ in range(K)]
    vs = [vrs.Buffer(gpu, S(4 * n)) for _ in range(K)]
    kt, vt = vrs.Buffer(gpu, S(4 * n)), vrs.Buffer(gpu, S(4 * n))
    best = 1e9
    for _ in range(reps + 1):
        for i in range(K):
            ks[i].copyFrom(src)
            vs[i].copyFrom(iota)
        gpu.waitIdle()
        t0 = time.perf_counter()
        for i in range(K):
            gpu.check(gpu.lib.vrs_sort_pairs_u32(gpu.handle, ks[i].handle, kt.handle, vs[i].handle, vt.handle, n))
        gpu.waitIdle()
        best = min(best, (time.perf_counter() - t0) / K)
    for b in ks + vs + [kt, vt, src, iota]:
        b.release()
    return best * 1e3


with vrs.GPUContext(0) as gpu:
    print("n          lsd      counted  pool")
    for n in [int(float(a)) for a in sys.argv[1:]] or (6 * 10 ** 6, 8 * 10 ** 6, 10 ** 7, 12 * 10 ** 6, 16 * 10 ** 6, 2 * 10 ** 7, 25 * 10 ** 6, 3 * 10 ** 7, 5 * 10 ** 7, 10 ** 8, 15 * 10 ** 7):
        row = []
        for mode in ("lsd", "counted", "pool"):
            gpu.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, (1 << 30) if mode == "lsd" else (1 << 22))
            gpu.setTuning(capi.VRS_TUNE_MSD_POOL_PAIRS, 1 if mode == "pool" else 0)
            row.append(timed(gpu, n))
        print(f"{n:10d} {row[0]:8.4f} {row[1]:8.4f} {row[2]:8.4f}", flush=True)
