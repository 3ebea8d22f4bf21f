"""A/B timing under bench.py's conditions: K pre-staged batches sorted back to back by vrs_sort_keys_u32 (each sort
reads a buffer that nothing touched recently), then the same with every kernel instrumented.
usage: VRS_LIB=<lib.so> [VRS_LIB_LENIENT=1] ab_bench.py TAG [N] [K] [groups]"""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from vkradixsort_amd import capi  # noqa: E402
if os.environ.get("VRS_LIB"):
    capi.LIB_PATH = Path(os.environ["VRS_LIB"]).resolve()
if os.environ.get("VRS_LIB_LENIENT"):  # timing an older build of the library: drop the entry points it does not have yet
    import ctypes
    _probe = ctypes.CDLL(str(capi.LIB_PATH))
    capi._SIGNATURES[:] = [sig for sig in capi._SIGNATURES if hasattr(_probe, sig[0])]
import vkradixsort_amd as vrs  # noqa: E402

tag = sys.argv[1]
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10 ** 8
K = int(sys.argv[3]) if len(sys.argv) > 3 else 12
groups = int(sys.argv[4]) if len(sys.argv) > 4 else 0
keys = np.random.RandomState(1).randint(0, 2 ** 32, size=n, dtype=np.uint32)
with vrs.GPUContext(0) as gpu:
    S = vrs.Buffer.BufferSettings
    src = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), keys)
    batches = [vrs.Buffer(gpu, S(4 * n)) for _ in range(K)]
    tmp = vrs.Buffer(gpu, S(4 * n))
    if groups:
        gpu.setTuning(8, groups)
    if os.environ.get("VRS_HYBRID_MIN"):
        gpu.setTuning(12, int(float(os.environ["VRS_HYBRID_MIN"])))
    if os.environ.get("VRS_RESERVE"):
        gpu.setTuning(16, int(os.environ["VRS_RESERVE"]))
    if os.environ.get("VRS_ONE_CALL_MIN"):  # 0: vrs_sort_keys_u32 runs the contract stages (4 x [histogram, prefix, scatter])
        gpu.setTuning(4, int(float(os.environ["VRS_ONE_CALL_MIN"])))
    if os.environ.get("VRS_POOL"):
        gpu.setTuning(17, int(os.environ["VRS_POOL"]))
    if os.environ.get("VRS_TOP_BITS"):
        gpu.setTuning(capi.VRS_TUNE_MSD_POOL_TOP_BITS, int(os.environ["VRS_TOP_BITS"]))
    if os.environ.get("VRS_SUB_BITS"):
        gpu.setTuning(capi.VRS_TUNE_MSD_POOL_SUB_BITS, int(os.environ["VRS_SUB_BITS"]))
    if os.environ.get("VRS_FUSED_PLAN"):
        gpu.setTuning(10, int(os.environ["VRS_FUSED_PLAN"]))

    def rearm():
        for b in batches:
            b.copyFrom(src)
        gpu.waitIdle()

    def run(mask):
        rearm()
        gpu.profileReset()
        gpu.profileEnableMask(mask)
        gpu.waitIdle()
        t0 = time.perf_counter()
        for b in batches:
            gpu.check(gpu.lib.vrs_sort_keys_u32(gpu.handle, b.handle, tmp.handle, n))
        gpu.waitIdle()
        dt = time.perf_counter() - t0
        gpu.profileEnable(False)
        return dt

    run(0)
    best = min(run(0) for _ in range(3))
    run(0xFF)
    line = f"{tag:28s} N={n} K={K}: {best / K * 1e3:.4f} ms/sort {n * K / best / 1e9:.1f} Gkeys/s"
    for kid, name in capi.KERNEL_NAMES.items():
        try:
            cnt, ms = gpu.profileQuery(kid)
        except vrs.VrsError:  # an older library without this kernel id
            continue
        if cnt:
            line += f" | {name} {ms / cnt * 1e3:.1f}us x{cnt // K}"
    out = np.empty(n, np.uint32)
    batches[-1].downloadWithStagingBuffer(out)
    line += f" | sorted={bool(np.all(out[1:] >= out[:-1]))}"
    print(line, flush=True)
