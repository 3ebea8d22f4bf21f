"""Per-kernel times (the context's own HIP events) of the one-call sort of another key kind.
usage: [VRS_LIB=lib.so] kind_breakdown.py KIND [N] [K]     KIND = keys_u64 | pairs_u32 | pairs_u64 | keys_u32"""
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

kind = sys.argv[1]
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10 ** 8
K = int(sys.argv[3]) if len(sys.argv) > 3 else 6
wide = kind.endswith("u64")
pairs = kind.startswith("pairs")
rs = np.random.RandomState(1)
keys = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32)
if wide:
    keys = (keys.astype(np.uint64) << np.uint64(32)) | keys[::-1].astype(np.uint64)
kb = keys.itemsize
with vrs.GPUContext(0) as gpu:
    S = vrs.Buffer.BufferSettings
    lib = gpu.lib
    if os.environ.get("VRS_NO_HYBRID"):
        gpu.setTuning(capi.VRS_TUNE_HYBRID, 0)
    src = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(kb * n), keys)
    k0, k1 = vrs.Buffer(gpu, S(kb * n)), vrs.Buffer(gpu, S(kb * n))
    if pairs:
        vsrc = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), np.arange(n, dtype=np.uint32))
        v0, v1 = vrs.Buffer(gpu, S(4 * n)), vrs.Buffer(gpu, S(4 * n))

    def once():
        if pairs and wide:
            gpu.check(lib.vrs_sort_pairs_u64(gpu.handle, k0.handle, k1.handle, v0.handle, v1.handle, n))
        elif pairs:
            gpu.check(lib.vrs_sort_pairs_u32(gpu.handle, k0.handle, k1.handle, v0.handle, v1.handle, n))
        elif wide:
            gpu.check(lib.vrs_sort_keys_u64(gpu.handle, k0.handle, k1.handle, n))
        else:
            gpu.check(lib.vrs_sort_keys_u32(gpu.handle, k0.handle, k1.handle, n))

    best = 1e9
    for r in range(K + 2):
        k0.copyFrom(src)
        if pairs:
            v0.copyFrom(vsrc)
        gpu.waitIdle()
        t0 = time.perf_counter()
        once()
        gpu.waitIdle()
        if r >= 2:
            best = min(best, time.perf_counter() - t0)
    gpu.profileReset()
    gpu.profileEnableMask(0xFF)
    for r in range(K):
        k0.copyFrom(src)
        if pairs:
            v0.copyFrom(vsrc)
        once()
    gpu.waitIdle()
    gpu.profileEnable(False)
    line = f"{kind} N={n}: {best * 1e3:.4f} ms/sort {n / best / 1e9:.1f} G/s"
    for kid, name in capi.KERNEL_NAMES.items():
        cnt, ms = gpu.profileQuery(kid)
        if cnt:
            line += f" | {name} {ms / cnt * 1e3:.1f}us x{cnt / K:g}"
    res = np.empty(n, keys.dtype)
    k0.downloadWithStagingBuffer(res)
    line += f" | sorted={bool(np.all(res[1:] >= res[:-1]))}"
    print(line, flush=True)
