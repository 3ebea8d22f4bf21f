"""Soak of the pool form at full size: many sorts of 10^8 keys back to back, every output verified on the device (ascending + the
input's order-independent fingerprints).   python tools/lab/pool_soak.py [sorts] [n]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import vkradixsort_amd as vrs  # noqa: E402

S = vrs.Buffer.BufferSettings
sorts = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10 ** 8
with vrs.GPUContext(0) as gpu:
    srcs = [vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), np.random.RandomState(s).randint(0, 2 ** 32, size=n, dtype=np.uint32)) for s in (11, 12, 13)]
    prints = [b.verifyKeys(n)[1:] for b in srcs]
    work = [vrs.Buffer(gpu, S(4 * n)) for _ in range(6)]
    tmp = vrs.Buffer(gpu, S(4 * n))
    bad = 0
    for it in range(0, sorts, len(work)):
        for i, w in enumerate(work):
            w.copyFrom(srcs[(it + i) % 3])
        for w in work:
            gpu.check(gpu.lib.vrs_sort_keys_u32(gpu.handle, w.handle, tmp.handle, n))
        for i, w in enumerate(work):
            r = w.verifyKeys(n)
            if r[0] != 0 or r[1:] != prints[(it + i) % 3]:
                bad += 1
                print("BAD sort", it + i, r, flush=True)
    import ctypes
    a, b = ctypes.c_uint64(), ctypes.c_uint64()
    gpu.check(gpu.lib.vrs_one_call_pool_sorts(gpu.handle, ctypes.byref(a), ctypes.byref(b)))
    print(f"pool soak: {sorts} sorts of {n} keys, {bad} bad; pool form took {a.value}, refused {b.value}", flush=True)
