"""Lab: the pool form with another cut of its buckets (VRS_TUNE_MSD_POOL_TOP_BITS), keys and pairs against numpy.   python tools/lab/top_bits_check.py [6|7|8]"""
import sys, numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent.parent))
import vkradixsort_amd as vrs
from vkradixsort_amd import capi
S=vrs.Buffer.BufferSettings
with vrs.GPUContext(0) as gpu:
    gpu.setTuning(capi.VRS_TUNE_MSD_POOL_TOP_BITS, int(sys.argv[1]) if len(sys.argv) > 1 else 7)
    gpu.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 1<<22)
    for n in (5000001, 30000001, 100000000):
        for rep in range(2):
            keys=np.random.RandomState(n%97+rep).randint(0,2**32,size=n,dtype=np.uint32)
            k0=vrs.Buffer.fillDeviceWithStagingBuffer(gpu,S(4*n),keys); k1=vrs.Buffer(gpu,S(4*n))
            gpu.check(gpu.lib.vrs_sort_keys_u32(gpu.handle,k0.handle,k1.handle,n))
            out=np.empty(n,np.uint32); k0.downloadWithStagingBuffer(out)
            print('keys',n,rep,np.array_equal(out,np.sort(keys)))
            k0.release(); k1.release()
        keys=np.random.RandomState(5).randint(0,2**32,size=n,dtype=np.uint32)&np.uint32(0xFFFF0FFF)
        vals=np.arange(n,dtype=np.uint32)
        k0=vrs.Buffer.fillDeviceWithStagingBuffer(gpu,S(4*n),keys); k1=vrs.Buffer(gpu,S(4*n))
        v0=vrs.Buffer.fillDeviceWithStagingBuffer(gpu,S(4*n),vals); v1=vrs.Buffer(gpu,S(4*n))
        gpu.check(gpu.lib.vrs_sort_pairs_u32(gpu.handle,k0.handle,k1.handle,v0.handle,v1.handle,n))
        ov=np.empty(n,np.uint32); v0.downloadWithStagingBuffer(ov)
        print('pairs',n,np.array_equal(ov,np.argsort(keys,kind='stable').astype(np.uint32)))
        for b in (k0,k1,v0,v1): b.release()
    import ctypes
    a,b=ctypes.c_uint64(),ctypes.c_uint64(); gpu.check(gpu.lib.vrs_one_call_pool_sorts(gpu.handle,ctypes.byref(a),ctypes.byref(b))); print('pool sorts',a.value,'refused',b.value)
