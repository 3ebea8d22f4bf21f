import sys, ctypes
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import vkradixsort_amd as vrs
from vkradixsort_amd import capi
S = vrs.Buffer.BufferSettings
mode = sys.argv[1]
dev = torch.device("cuda", 0)
def run(gpu, sizes):
    for n in sizes:
        k = torch.randint(-2**31, 2**31, (n,), dtype=torch.int32, device=dev)
        tmp = torch.empty_like(k)
        torch.cuda.synchronize()
        k0 = vrs.Buffer(gpu, S(4*n), device_ptr=k.data_ptr()); k1 = vrs.Buffer(gpu, S(4*n), device_ptr=tmp.data_ptr())
        gpu.check(gpu.lib.vrs_sort_keys_u32(gpu.handle, k0.handle, k1.handle, n))
        gpu.waitIdle()
        u = k.to(torch.int64) & 0xFFFFFFFF
        ok = bool((u[1:] >= u[:-1]).all().item())
        print(mode, n, "sorted", ok, flush=True)
        k0.release(); k1.release()
if mode == "borrowed":
    with vrs.GPUContext(0, stream=torch.cuda.current_stream().cuda_stream) as gpu:
        run(gpu, [10**8, 10**8, 3000000, 10**8, 10**8, 5000000, 5000000])
elif mode == "own_blocking":
    with vrs.GPUContext(0) as gpu:
        gpu.setTuning(capi.VRS_TUNE_ASYNC_SORT, 0)
        run(gpu, [10**8, 10**8, 3000000, 10**8, 10**8])
else:
    with vrs.GPUContext(0) as gpu:
        run(gpu, [10**8, 10**8, 3000000, 10**8, 10**8])
