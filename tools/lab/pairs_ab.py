import os, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, "/root/repo")
from vkradixsort_amd import capi
if os.environ.get("VRS_LIB"):
    capi.LIB_PATH = Path(os.environ["VRS_LIB"]).resolve()
import vkradixsort_amd as vrs
tag=sys.argv[1]; n=int(float(sys.argv[2])); K=int(sys.argv[3])
S=vrs.Buffer.BufferSettings
keys=np.random.RandomState(1).randint(0,2**32,size=n,dtype=np.uint32); vals=np.arange(n,dtype=np.uint32)
with vrs.GPUContext(0) as gpu:
    src=vrs.Buffer.fillDeviceWithStagingBuffer(gpu,S(4*n),keys); vsrc=vrs.Buffer.fillDeviceWithStagingBuffer(gpu,S(4*n),vals)
    kb=[vrs.Buffer(gpu,S(4*n)) for _ in range(K)]; vb=[vrs.Buffer(gpu,S(4*n)) for _ in range(K)]
    kt,vt=vrs.Buffer(gpu,S(4*n)),vrs.Buffer(gpu,S(4*n))
    if os.environ.get('VRS_PACKED'): gpu.setTuning(capi.VRS_TUNE_MSD_POOL_PAIRS_PACKED, int(os.environ['VRS_PACKED']))
    best=1e9
    for rep in range(4):
        for i in range(K): kb[i].copyFrom(src); vb[i].copyFrom(vsrc)
        gpu.waitIdle(); t0=time.perf_counter()
        for i in range(K): gpu.check(gpu.lib.vrs_sort_pairs_u32(gpu.handle,kb[i].handle,kt.handle,vb[i].handle,vt.handle,n))
        gpu.waitIdle(); dt=(time.perf_counter()-t0)/K
        if rep: best=min(best,dt)
    print(f"{tag:10s} pairs n={n}: {best*1e3:.4f} ms/sort", flush=True)
    if os.environ.get("VRS_CHECK"):
        ok=np.empty(n,dtype=np.uint32); ov=np.empty(n,dtype=np.uint32)
        kb[0].downloadWithStagingBuffer(ok); vb[0].downloadWithStagingBuffer(ov)
        good = bool((np.diff(ok.astype(np.int64))>=0).all()) and bool((keys[ov]==ok).all())
        eq = ok[1:]==ok[:-1]; good = good and bool((ov[1:][eq] > ov[:-1][eq]).all())
        print(f"{tag:10s} check: {'ok' if good else 'BAD'}", flush=True)
