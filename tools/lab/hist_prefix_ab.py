"""contract stages (histograms + sort, four passes) with the prefix made by the histogram stage (VRS_TUNE_HIST_PREFIX) on / off,
alternating on one box: python tools/lab/hist_prefix_ab.py N [K] [min_rows]"""
import ctypes, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import vkradixsort_amd as vrs
from vkradixsort_amd import capi
n = int(float(sys.argv[1])); K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
S = vrs.Buffer.BufferSettings

keys = np.random.RandomState(1).randint(0, 2 ** 32, size=n, dtype=np.uint32)
ref = np.sort(keys)
with vrs.GPUContext(0) as gpu:
    W = gpu.lib.vrs_workgroup_count(n, B)
    src = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), keys)
    a = [vrs.Buffer(gpu, S(4 * n)) for _ in range(K)]
    tmp, table = vrs.Buffer(gpu, S(4 * n)), vrs.Buffer(gpu, S(4 * 256 * W))
    pcs = [capi.PushConstants(n, 8 * i, W, B) for i in range(4)]
    def sort(buf):
        x, y = buf, tmp
        for pc in pcs:
            gpu.check(gpu.lib.vrs_multi_radixsort_histograms(gpu.handle, x.handle, table.handle, ctypes.byref(pc)))
            gpu.check(gpu.lib.vrs_multi_radixsort(gpu.handle, x.handle, y.handle, table.handle, ctypes.byref(pc)))
            x, y = y, x
    for rep in range(5):
        for mode in (0,):
            for b in a: b.copyFrom(src)
            gpu.waitIdle(); t0 = time.perf_counter()
            for b in a: sort(b)
            gpu.waitIdle(); dt = (time.perf_counter() - t0) / K
            got = np.empty(n, np.uint32); a[K - 1].downloadWithStagingBuffer(got)
            print(f"n={n} B={B}: {dt * 1e3:.4f} ms/sort {'ok' if np.array_equal(got, ref) else 'BAD'}", flush=True)
