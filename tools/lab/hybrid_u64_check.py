"""Hybrid form of vrs_sort_keys_u64: correctness vs numpy, which form ran, time against the LSD form, per-kernel times.
usage: hybrid_u64_check.py [n,n,...] [dist,dist,...] [hybrid_min_keys]"""
import ctypes
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import vkradixsort_amd as vrs  # noqa: E402
from vkradixsort_amd import capi  # noqa: E402

S = vrs.Buffer.BufferSettings


def make(n, dist, rs):
    k = (rs.randint(0, 2 ** 32, n, dtype=np.uint64) << np.uint64(32)) | rs.randint(0, 2 ** 32, n, dtype=np.uint64)
    if dist == "44bit":  # the reference's SORT_64_BIT generator
        k >>= np.uint64(20)
    elif dist == "low32":
        k &= np.uint64(0xFFFFFFFF)
    elif dist == "sorted":
        k.sort()
    return k


def main():
    sizes = [int(float(x)) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [10 ** 8]
    dists = sys.argv[2].split(",") if len(sys.argv) > 2 else ["uniform", "44bit", "low32", "sorted"]
    rs = np.random.RandomState(3)
    with vrs.GPUContext(0) as gpu:
        lib = gpu.lib
        if len(sys.argv) > 3:
            gpu.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, int(float(sys.argv[3])))
        for n in sizes:
            for dist in dists:
                keys = make(n, dist, rs)
                ref = np.sort(keys)
                src = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(8 * n), keys)
                k0, k1 = vrs.Buffer(gpu, S(8 * n)), vrs.Buffer(gpu, S(8 * n))
                line = f"u64 n={n} {dist}:"
                for hybrid in (1, 0):
                    gpu.setTuning(capi.VRS_TUNE_HYBRID, hybrid)
                    h0 = ctypes.c_uint64()
                    gpu.check(lib.vrs_one_call_hybrid_sorts(gpu.handle, ctypes.byref(h0)))
                    ts = []
                    for r in range(6):
                        k0.copyFrom(src)
                        gpu.waitIdle()
                        t0 = time.perf_counter()
                        gpu.check(lib.vrs_sort_keys_u64(gpu.handle, k0.handle, k1.handle, n))
                        gpu.waitIdle()
                        ts.append(time.perf_counter() - t0)
                    h1 = ctypes.c_uint64()
                    gpu.check(lib.vrs_one_call_hybrid_sorts(gpu.handle, ctypes.byref(h1)))
                    out = np.empty(n, np.uint64)
                    k0.downloadWithStagingBuffer(out)
                    line += f" | hybrid={hybrid}: exact={bool(np.array_equal(out, ref))} took_hybrid={h1.value - h0.value}/6 min={min(ts[2:]) * 1e3:.3f}ms"
                gpu.profileReset()
                gpu.profileEnable(True)
                gpu.setTuning(capi.VRS_TUNE_HYBRID, 1)
                k0.copyFrom(src)
                gpu.check(lib.vrs_sort_keys_u64(gpu.handle, k0.handle, k1.handle, n))
                gpu.waitIdle()
                gpu.profileEnable(False)
                for kid, name in capi.KERNEL_NAMES.items():
                    cnt, ms = gpu.profileQuery(kid)
                    if cnt:
                        line += f" | {name} {ms / cnt * 1e3:.1f}us x{cnt}"
                print(line, flush=True)
                for b in (src, k0, k1):
                    b.release()


if __name__ == "__main__":
    main()
