"""Hybrid form of vrs_sort_keys_u32 (MSD partition + LDS-local sort): correctness vs numpy over sizes / distributions, which
form ran, and time against the LSD form.   usage: hybrid_check.py [quick | sizes n,n,... [dist,dist,...] [hybrid_min_keys]]"""
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
    k = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32)
    if dist == "sorted":
        k.sort()
    elif dist == "reverse":
        k = np.sort(k)[::-1].copy()
    elif dist == "28bit":
        k >>= 4
    elif dist == "24bit":
        k >>= 8
    elif dist == "max_keys":
        k = np.where(k % 3 == 0, np.uint32(0xFFFFFFFF), k).astype(np.uint32)
    elif dist == "dups":
        k = (k & np.uint32(0xFFFC0000)) | (k & np.uint32(0xFF))  # 256 distinct low parts per bucket: many ties
    elif dist == "hot_bucket":
        k[: n // 300] = (k[: n // 300] & np.uint32(0x3FFFF)) | np.uint32(0x12340000)  # one bucket with 0.33 % of the keys
    return k


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    sizes = [int(float(x)) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 and sys.argv[1] == "sizes" else None
    dists = sys.argv[3].split(",") if len(sys.argv) > 3 else ["uniform"]
    min_keys = int(float(sys.argv[4])) if len(sys.argv) > 4 else None
    rs = np.random.RandomState(3)
    with vrs.GPUContext(0) as gpu:
        lib = gpu.lib
        if min_keys is not None:
            gpu.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, min_keys)
        cases = [((1 << 24), "uniform"), ((1 << 24) + 12345, "uniform"), (30000001, "uniform"), (30000001, "sorted"), (30000001, "reverse"),
                 (20000003, "28bit"), (25000000, "max_keys"), (25000000, "dups"), (25000000, "hot_bucket"), (10 ** 8, "uniform"), (2 * 10 ** 8 + 77, "uniform")]
        if quick:
            cases = [(40000000, "uniform"), (50000000, "uniform"), (60000000, "uniform"), (70000000, "uniform"), (80000000, "uniform"),
                     (90000000, "uniform"), (10 ** 8, "uniform"), (10 ** 8, "sorted"), (10 ** 8, "reverse"), (10 ** 8, "28bit"),
                     (10 ** 8, "dups"), (102000000, "uniform"), (10 ** 8, "24bit"), (10 ** 8, "hot_bucket")]
        if sizes:
            cases = [(n, d) for n in sizes for d in dists]
        for n, dist in cases:
            keys = make(n, dist, rs)
            ref = np.sort(keys)
            src = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), keys)
            k0, k1 = vrs.Buffer(gpu, S(4 * n)), vrs.Buffer(gpu, S(4 * n))
            line = f"n={n} {dist}:"
            for hybrid in (1, 0):
                gpu.setTuning(capi.VRS_TUNE_HYBRID, hybrid)
                h0 = ctypes.c_uint64()
                gpu.check(lib.vrs_one_call_hybrid_sorts(gpu.handle, ctypes.byref(h0)))
                ts = []
                for r in range(7):
                    k0.copyFrom(src)
                    gpu.waitIdle()
                    t0 = time.perf_counter()
                    gpu.check(lib.vrs_sort_keys_u32(gpu.handle, k0.handle, k1.handle, n))
                    gpu.waitIdle()
                    ts.append(time.perf_counter() - t0)
                h1 = ctypes.c_uint64()
                gpu.check(lib.vrs_one_call_hybrid_sorts(gpu.handle, ctypes.byref(h1)))
                out = np.empty(n, np.uint32)
                k0.downloadWithStagingBuffer(out)
                ok = bool(np.array_equal(out, ref))
                line += f" | hybrid={hybrid}: exact={ok} took_hybrid={h1.value - h0.value}/7 min={min(ts[2:]) * 1e3:.3f}ms"
                if not ok:
                    bad = np.flatnonzero(out != ref)
                    line += f" FIRST BAD {bad[0]} of {bad.size}"
            gpu.profileReset()
            gpu.profileEnable(True)
            gpu.setTuning(capi.VRS_TUNE_HYBRID, 1)
            k0.copyFrom(src)
            gpu.check(lib.vrs_sort_keys_u32(gpu.handle, k0.handle, k1.handle, n))
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
