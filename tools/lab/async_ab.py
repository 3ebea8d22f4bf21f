"""A/B on one box: the enqueue-only and the blocking form of the one-call sorts, back to back over pre-staged batches, alternating.
usage: async_ab.py [n] [reps] [pairs]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import vkradixsort_amd as vrs  # noqa: E402
from vkradixsort_amd import capi  # noqa: E402

S = vrs.Buffer.BufferSettings
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
pairs = len(sys.argv) > 3 and sys.argv[3] == "pairs"
keys = np.random.RandomState(1).randint(0, 2 ** 32, size=n, dtype=np.uint32)
with vrs.GPUContext(0) as gpu:
    lib = gpu.lib
    src = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), keys)
    bat = [vrs.Buffer(gpu, S(4 * n)) for _ in range(reps)]
    vals = [vrs.Buffer(gpu, S(4 * n)) for _ in range(reps)] if pairs else None
    k1 = vrs.Buffer(gpu, S(4 * n))
    v1 = vrs.Buffer(gpu, S(4 * n)) if pairs else None
    for rnd in range(4):
        for mode in (1, 0):
            gpu.setTuning(capi.VRS_TUNE_ASYNC_SORT, mode)
            for b in bat:
                b.copyFrom(src)
            gpu.waitIdle()
            t0 = time.perf_counter()
            for i, b in enumerate(bat):
                if pairs:
                    gpu.check(lib.vrs_sort_pairs_u32(gpu.handle, b.handle, k1.handle, vals[i].handle, v1.handle, n))
                else:
                    gpu.check(lib.vrs_sort_keys_u32(gpu.handle, b.handle, k1.handle, n))
            gpu.waitIdle()
            dt = (time.perf_counter() - t0) / reps
            print(f"round {rnd} n={n} {'pairs' if pairs else 'keys'} async={mode}: {dt * 1e3:.4f} ms/sort", flush=True)
    print("sorted:", all(b.verifyKeys(n)[0] == 0 for b in bat))
