import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import vkradixsort_amd as vrs
from vkradixsort_amd import capi
S = vrs.Buffer.BufferSettings
with vrs.GPUContext(0) as ctx:
    lib = ctx.lib
    ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 1 << 22)
    for span, floor_key in ((27, 0xA3000000), (29, 0x60000000), (27, 0), (30, 0x40000000)):
        for fast in (0, 2):
            ctx.setTuning(capi.VRS_TUNE_HYBRID_FAST_COUNT, fast)
            n = (1 << 23) + 3
            rs = np.random.RandomState(span)
            k = (np.uint32(floor_key) + rs.randint(0, 1 << span, size=n, dtype=np.uint32)).astype(np.uint32)
            k0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), k)
            k1 = vrs.Buffer(ctx, S(4 * n))
            ctx.check(lib.vrs_sort_keys_u32_ranged(ctx.handle, k0.handle, k1.handle, n, floor_key))
            out = np.empty(n, np.uint32)
            k0.downloadWithStagingBuffer(out)
            ref = np.sort(k)
            bad = np.nonzero(out != ref)[0]
            print(span, hex(floor_key), "fast", fast, "ok" if bad.size == 0 else f"BAD {bad.size} first {bad[0]} out {out[bad[0]]:#x} ref {ref[bad[0]]:#x} sorted={bool(np.all(out[1:]>=out[:-1]))} perm={bool(np.array_equal(np.sort(out), ref))}")
            k0.release(); k1.release()
