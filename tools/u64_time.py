"""Time uint64 sorts through the stage API with a chosen contract B (development aid)."""
import ctypes
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vkradixsort_amd as vrs  # noqa: E402
from vkradixsort_amd import capi  # noqa: E402

S = vrs.Buffer.BufferSettings
n = 10 ** 8
rs = np.random.RandomState(1)
keys = (rs.randint(0, 2 ** 32, n, dtype=np.uint64) << np.uint64(32)) | rs.randint(0, 2 ** 32, n, dtype=np.uint64)
with vrs.GPUContext(0) as gpu:
    lib = gpu.lib
    src = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(8 * n), keys)
    b = [vrs.Buffer(gpu, S(8 * n)), vrs.Buffer(gpu, S(8 * n))]
    for B, variant in [(16, 0), (32, 0), (64, 0)]:
        gpu.setTuning(capi.VRS_TUNE_SCATTER_VARIANT, variant)
        W = lib.vrs_workgroup_count(n, B)
        h = vrs.Buffer(gpu, S(W * 1024))
        best = 1e9
        for r in range(5):
            b[0].copyFrom(src)
            gpu.waitIdle()
            if r == 1:
                gpu.profileReset(); gpu.profileEnable(True)
            t0 = time.perf_counter()
            for i in range(8):
                pc = vrs.PushConstants(n, 8 * i, W, B)
                gpu.check(lib.vrs_multi_radixsort_histograms_u64(gpu.handle, b[i % 2].handle, h.handle, ctypes.byref(pc)))
                gpu.check(lib.vrs_multi_radixsort_u64(gpu.handle, b[i % 2].handle, b[(i + 1) % 2].handle, h.handle, ctypes.byref(pc)))
            gpu.waitIdle()
            best = min(best, time.perf_counter() - t0)
        gpu.profileEnable(False)
        out = np.empty(n, np.uint64)
        b[0].downloadWithStagingBuffer(out)
        line = f"u64 N={n} B={B} variant={variant}: {best*1e3:.3f} ms {n/best/1e9:.1f} Gkeys/s {192*n/best/8e12*100:.1f}%roof sorted={bool(np.all(out[1:]>=out[:-1]))}"
        for kid, name in capi.KERNEL_NAMES.items():
            cnt, ms = gpu.profileQuery(kid)
            if cnt:
                line += f" | {name}: {ms/cnt*1e3:.1f}us"
        print(line, flush=True)
        h.release()
