"""Hybrid form of vrs_sort_pairs_u32: stability / correctness vs numpy, which form ran, time against the LSD form, per-kernel
times.   usage: hybrid_pairs_check.py [n,n,...] [dist,dist,...] [hybrid_min_keys]"""
import ctypes
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import vkradixsort_amd as vrs  # noqa: E402
from vkradixsort_amd import capi  # noqa: E402
from hybrid_check import make  # noqa: E402

S = vrs.Buffer.BufferSettings


def main():
    sizes = [int(float(x)) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [10 ** 8]
    dists = sys.argv[2].split(",") if len(sys.argv) > 2 else ["uniform", "sorted", "28bit", "dups", "24bit", "hot_bucket"]
    rs = np.random.RandomState(3)
    with vrs.GPUContext(0) as gpu:
        lib = gpu.lib
        if len(sys.argv) > 3:
            gpu.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, int(float(sys.argv[3])))
        for n in sizes:
            for dist in dists:
                keys = make(n, dist, rs)
                vals = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32)
                order = np.argsort(keys, kind="stable")
                rk, rv = keys[order], vals[order]
                ksrc = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), keys)
                vsrc = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), vals)
                k0, k1, v0, v1 = (vrs.Buffer(gpu, S(4 * n)) for _ in range(4))
                line = f"pairs n={n} {dist}:"
                for hybrid in (1, 0):
                    gpu.setTuning(capi.VRS_TUNE_HYBRID, hybrid)
                    h0 = ctypes.c_uint64()
                    gpu.check(lib.vrs_one_call_hybrid_sorts(gpu.handle, ctypes.byref(h0)))
                    ts = []
                    for r in range(7):
                        k0.copyFrom(ksrc)
                        v0.copyFrom(vsrc)
                        gpu.waitIdle()
                        t0 = time.perf_counter()
                        gpu.check(lib.vrs_sort_pairs_u32(gpu.handle, k0.handle, k1.handle, v0.handle, v1.handle, n))
                        gpu.waitIdle()
                        ts.append(time.perf_counter() - t0)
                    h1 = ctypes.c_uint64()
                    gpu.check(lib.vrs_one_call_hybrid_sorts(gpu.handle, ctypes.byref(h1)))
                    ok_, ov_ = np.empty(n, np.uint32), np.empty(n, np.uint32)
                    k0.downloadWithStagingBuffer(ok_)
                    v0.downloadWithStagingBuffer(ov_)
                    ok = bool(np.array_equal(ok_, rk)) and bool(np.array_equal(ov_, rv))
                    line += f" | hybrid={hybrid}: exact={ok} took_hybrid={h1.value - h0.value}/7 min={min(ts[2:]) * 1e3:.3f}ms"
                gpu.profileReset()
                gpu.profileEnable(True)
                gpu.setTuning(capi.VRS_TUNE_HYBRID, 1)
                k0.copyFrom(ksrc)
                v0.copyFrom(vsrc)
                gpu.check(lib.vrs_sort_pairs_u32(gpu.handle, k0.handle, k1.handle, v0.handle, v1.handle, n))
                gpu.waitIdle()
                gpu.profileEnable(False)
                for kid, name in capi.KERNEL_NAMES.items():
                    cnt, ms = gpu.profileQuery(kid)
                    if cnt:
                        line += f" | {name} {ms / cnt * 1e3:.1f}us x{cnt}"
                print(line, flush=True)
                for b in (ksrc, vsrc, k0, k1, v0, v1):
                    b.release()


if __name__ == "__main__":
    main()
