import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import vkradixsort_amd as vrs
from vkradixsort_amd import capi
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 7
keys = np.random.RandomState(3).randint(0, 2 ** 32, size=n, dtype=np.uint32)
S = vrs.Buffer.BufferSettings
with vrs.GPUContext(0) as gpu:
    gpu.setTuning(capi.VRS_TUNE_MSD_POOL_TOP_BITS, 8)
    gpu.setTuning(capi.VRS_TUNE_MSD_POOL_SUB_BITS, 8)
    b = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), keys)
    t = vrs.Buffer(gpu, S(4 * n))
    gpu.check(gpu.lib.vrs_sort_keys_u32(gpu.handle, b.handle, t.handle, n))
    gpu.waitIdle()
    out = np.empty(n, dtype=np.uint32)
    b.downloadWithStagingBuffer(out)
ref = np.sort(keys)
bad = np.nonzero(out != ref)[0]
print("mismatches", bad.size, "of", n)
for i in bad[:12]:
    print(i, hex(out[i]), hex(ref[i]))
if bad.size:
    print("low16 equal where bad:", float(((out[bad] & 0xFFFF) == (ref[bad] & 0xFFFF)).mean()), "high16 equal:", float(((out[bad] >> 16) == (ref[bad] >> 16)).mean()))
    print("sorted multiset equal:", bool(np.array_equal(np.sort(out), ref)))
