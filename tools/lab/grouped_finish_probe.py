"""What a rank of an 8 x 10^8-key sort does with its received keys: n keys of `top_bytes` top bytes, grouped by top byte --
vrs_msd_finish_grouped_u32 against vrs_sort_keys_u32_ranged (wall time of K back-to-back calls + the context's kernel events).
usage: grouped_finish_probe.py [n] [top_bytes] [K]"""
import ctypes
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import vkradixsort_amd as vrs  # noqa: E402
from vkradixsort_amd import capi  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
first = 0x40
rs = np.random.RandomState(3)
keys = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32)
keys = (keys & np.uint32(0x00FFFFFF)) | (rs.randint(first, first + T, size=n).astype(np.uint32) << np.uint32(24))
grouped = keys[np.argsort(keys >> np.uint32(24), kind="stable")]
S = vrs.Buffer.BufferSettings
with vrs.GPUContext(0) as gpu:
    lib = gpu.lib
    src = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), grouped)
    bufs = [vrs.Buffer(gpu, S(4 * n)) for _ in range(K)]
    out = vrs.Buffer(gpu, S(4 * n))

    def run(which, events):
        for b in bufs:
            b.copyFrom(src)
        gpu.waitIdle()
        gpu.profileReset()
        gpu.profileEnableMask(0xFF if events else 0)
        t0 = time.perf_counter()
        for b in bufs:
            if which == "grouped":
                gpu.check(lib.vrs_msd_finish_grouped_u32(gpu.handle, b.handle, out.handle, n, first, T))
            else:
                gpu.check(lib.vrs_sort_keys_u32_ranged(gpu.handle, b.handle, out.handle, n, first << 24))
        gpu.waitIdle()
        dt = time.perf_counter() - t0
        gpu.profileEnable(False)
        return dt

    for which in ("grouped", "ranged"):
        run(which, False)
        best = min(run(which, False) for _ in range(3))
        run(which, True)
        line = f"{which:8s} n={n} top_bytes={T}: {best / K * 1e3:.4f} ms per call"
        for kid, name in capi.KERNEL_NAMES.items():
            cnt, ms = gpu.profileQuery(kid)
            if cnt:
                line += f" | {name} {ms / cnt * 1e3:.1f}us x{cnt / K:g}"
        res = np.empty(n, np.uint32)
        (out if which == "grouped" else bufs[-1]).downloadWithStagingBuffer(res)
        took = ctypes.c_int(-1)
        if which == "grouped":
            lib.vrs_msd_finish_status(gpu.handle, ctypes.byref(took))
        line += f" | sorted={bool(np.all(res[1:] >= res[:-1]))} took={took.value}"
        print(line, flush=True)
