"""A/B of vrs_sort_pairs_u64 (uint64 keys + uint32 payloads) between library builds: VRS_LIB=<lib.so> pairs64_ab.py TAG [N] [reps]"""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from vkradixsort_amd import capi  # noqa: E402
if os.environ.get("VRS_LIB"):
    capi.LIB_PATH = Path(os.environ["VRS_LIB"]).resolve()
import vkradixsort_amd as vrs  # noqa: E402

tag = sys.argv[1]
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 5 * 10 ** 7
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
rs = np.random.RandomState(1)
keys = (rs.randint(0, 2 ** 32, n, dtype=np.uint64) << np.uint64(32)) | rs.randint(0, 2 ** 32, n, dtype=np.uint64)
vals = np.arange(n, dtype=np.uint32)
S = vrs.Buffer.BufferSettings
with vrs.GPUContext(0) as gpu:
    src = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(8 * n), keys)
    vsrc = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), vals)
    k0, k1, v0, v1 = vrs.Buffer(gpu, S(8 * n)), vrs.Buffer(gpu, S(8 * n)), vrs.Buffer(gpu, S(4 * n)), vrs.Buffer(gpu, S(4 * n))
    ts = []
    for r in range(reps + 1):
        k0.copyFrom(src)
        v0.copyFrom(vsrc)
        gpu.waitIdle()
        t0 = time.perf_counter()
        gpu.check(gpu.lib.vrs_sort_pairs_u64(gpu.handle, k0.handle, k1.handle, v0.handle, v1.handle, n))
        gpu.waitIdle()
        ts.append(time.perf_counter() - t0)
    print(f"{tag:16s} N={n}: best {min(ts[1:]) * 1e3:.4f} ms, median {sorted(ts[1:])[len(ts) // 2 - 1] * 1e3:.4f} ms, sorted={k0.verifyKeys(n)[0] == 0 if hasattr(k0, 'verifyKeys') else '?'}", flush=True)
