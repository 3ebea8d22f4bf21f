"""BASELINE.json single-GPU configs through both paths of the one-call entry points: wall time per sort, keys resident,
no events, best of `reps`; every result checked against numpy.   python tools/config_matrix.py > profiles/r01_config_matrix.json"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vkradixsort_amd as vrs  # noqa: E402
from vkradixsort_amd import capi  # noqa: E402


def run(gpu, keys, vals, reps):
    lib, n = gpu.lib, keys.size
    kb = keys.itemsize
    S = vrs.Buffer.BufferSettings
    src = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(kb * n), keys)
    k0, k1 = vrs.Buffer(gpu, S(kb * n)), vrs.Buffer(gpu, S(kb * n))
    bufs = [src, k0, k1]
    if vals is not None:
        vsrc = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), vals)
        v0, v1 = vrs.Buffer(gpu, S(4 * n)), vrs.Buffer(gpu, S(4 * n))
        bufs += [vsrc, v0, v1]
    order = np.argsort(keys, kind="stable")
    out = {}
    for name, min_keys in (("contract_passes", 0), ("one_call", 1 << 20)):
        gpu.setTuning(capi.VRS_TUNE_ONE_CALL_MIN_KEYS, min_keys)
        best = 1e9
        for r in range(reps + 2):
            k0.copyFrom(src)
            if vals is not None:
                v0.copyFrom(vsrc)
            gpu.waitIdle()
            t0 = time.perf_counter()
            if vals is not None and kb == 8:
                gpu.check(lib.vrs_sort_pairs_u64(gpu.handle, k0.handle, k1.handle, v0.handle, v1.handle, n))
            elif vals is not None:
                gpu.check(lib.vrs_sort_pairs_u32(gpu.handle, k0.handle, k1.handle, v0.handle, v1.handle, n))
            elif kb == 8:
                gpu.check(lib.vrs_sort_keys_u64(gpu.handle, k0.handle, k1.handle, n))
            else:
                gpu.check(lib.vrs_sort_keys_u32(gpu.handle, k0.handle, k1.handle, n))
            gpu.waitIdle()
            if r >= 2:
                best = min(best, time.perf_counter() - t0)
        res = np.empty(n, keys.dtype)
        k0.downloadWithStagingBuffer(res)
        ok = bool(np.array_equal(res, keys[order]))
        if vals is not None:
            rv = np.empty(n, np.uint32)
            v0.downloadWithStagingBuffer(rv)
            ok = ok and bool(np.array_equal(rv, vals[order]))
        out[name] = {"ms": round(best * 1e3, 4), "G_per_s": round(n / best / 1e9, 2), "bit_exact": ok}
    gpu.setTuning(capi.VRS_TUNE_ONE_CALL_MIN_KEYS, capi.ONE_CALL_MIN_KEYS_DEFAULT)
    for b in bufs:
        b.release()
    return out


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rs = np.random.RandomState(1)
    result = {}
    with vrs.GPUContext(0) as gpu:
        result["device"] = gpu.deviceInfo()[0]
        k7 = rs.randint(0, 2 ** 32, size=10 ** 7, dtype=np.uint32)
        result["configs[1] 10^7 uint32 keys"] = run(gpu, k7, None, reps)
        k8 = rs.randint(0, 2 ** 32, size=10 ** 8, dtype=np.uint32)
        result["configs[2] 10^8 uint32 keys"] = run(gpu, k8, None, reps)
        result["configs[3] 10^8 uint32 key + uint32 payload pairs"] = run(gpu, k8, np.arange(10 ** 8, dtype=np.uint32), reps)
        result["10^8 reference-style 28-bit keys"] = run(gpu, k8 >> np.uint32(4), None, reps)
        result["10^8 sorted uint32 keys"] = run(gpu, np.sort(k8), None, reps)
        k64 = (k8.astype(np.uint64) << np.uint64(32)) | k8[::-1].astype(np.uint64)
        result["10^8 uint64 keys (SORT_64_BIT)"] = run(gpu, k64, None, max(3, reps // 2))
        n5 = 5 * 10 ** 7
        result["5 x 10^7 uint64 key + uint32 payload pairs"] = run(gpu, k64[:n5].copy(), np.arange(n5, dtype=np.uint32), max(3, reps // 2))
        # beyond 1.03e8 elements: buckets of 6657-13312 pairs / 64-bit keys (the 1024-thread local sort of the hybrid form)
        k2 = np.concatenate([k8, rs.randint(0, 2 ** 32, size=10 ** 8, dtype=np.uint32)])
        result["2 x 10^8 uint32 key + uint32 payload pairs"] = run(gpu, k2, np.arange(2 * 10 ** 8, dtype=np.uint32), 3)
        k64b = (k2.astype(np.uint64) << np.uint64(32)) | k2[::-1].astype(np.uint64)
        del k2
        result["2 x 10^8 uint64 keys"] = run(gpu, k64b, None, 3)
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
