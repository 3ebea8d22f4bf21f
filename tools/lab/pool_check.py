"""Pool form of vrs_sort_keys_u32 (hybrid form without the counting read): correctness vs numpy over sizes / distributions,
whether the form was taken or refused, and time against the counted form (back-to-back sorts, like bench.py's timed region).
usage: pool_check.py [cases | time [n] [reps]]"""
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
    elif dist == "gauss":
        g = rs.normal(2.0 ** 31, 2.0 ** 28, size=n)
        k = np.clip(g, 0, 2.0 ** 32 - 1).astype(np.uint32)
    elif dist == "max_keys":
        k = np.where(k % 3 == 0, np.uint32(0xFFFFFFFF), k).astype(np.uint32)
    elif dist == "dups":
        k = (k & np.uint32(0xFFFC0000)) | (k & np.uint32(0xFF))
    elif dist == "hot_bucket":
        k[: n // 300] = (k[: n // 300] & np.uint32(0x3FFFF)) | np.uint32(0x12340000)
    elif dist == "halves":  # the first half of the input holds small keys, the second large ones: slices differ
        k[: n // 2] >>= 1
        k[n // 2:] |= np.uint32(0x80000000)
    elif dist == "tile_period":  # the sampled head of every tile is unlike the rest of it: the sample misjudges every region
        t = np.arange(n) % 8192
        k = np.where(t < 256, k >> 1, k | np.uint32(0x80000000)).astype(np.uint32)
    elif dist == "const":
        k[:] = 0xDEADBEEF
    return k


def pool_counts(gpu):
    a, b = ctypes.c_uint64(), ctypes.c_uint64()
    gpu.check(gpu.lib.vrs_one_call_pool_sorts(gpu.handle, ctypes.byref(a), ctypes.byref(b)))
    return a.value, b.value


def run_cases():
    rs = np.random.RandomState(5)
    cases = [((1 << 22) + 5, "uniform"), (13000001, "uniform"), (30000001, "uniform"), (30000001, "sorted"), (30000001, "reverse"),
             (20000003, "28bit"), (20000000, "24bit"), (25000000, "gauss"), (25000000, "max_keys"), (25000000, "dups"),
             (25000000, "hot_bucket"), (25000000, "halves"), (25000000, "tile_period"), (20000000, "const"), (10 ** 8, "uniform"),
             (10 ** 8 + 4097, "uniform"), (2 * 10 ** 8 + 77, "uniform")]
    bad_total = 0
    with vrs.GPUContext(0) as gpu:
        lib = gpu.lib
        gpu.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 1 << 22)
        gpu.setTuning(capi.VRS_TUNE_MSD_POOL_MIN_KEYS, 1 << 22)
        for n, dist in cases:
            keys = make(n, dist, rs)
            ref = np.sort(keys)
            src = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), keys)
            k0, k1 = vrs.Buffer(gpu, S(4 * n)), vrs.Buffer(gpu, S(4 * n))
            line = f"n={n} {dist}:"
            for mode, misplace, async_ in ((2, 0, 0), (2, 1, 0), (2, 0, 1), (0, 0, 0)):
                gpu.setTuning(capi.VRS_TUNE_MSD_POOL, mode)
                gpu.setTuning(capi.VRS_TUNE_DEBUG_MISPLACE_STREAMS, misplace)
                gpu.setTuning(capi.VRS_TUNE_ASYNC_SORT, async_)
                p0 = pool_counts(gpu)
                ts = []
                for r in range(3):
                    k0.copyFrom(src)
                    gpu.waitIdle()
                    t0 = time.perf_counter()
                    gpu.check(lib.vrs_sort_keys_u32(gpu.handle, k0.handle, k1.handle, n))
                    gpu.waitIdle()
                    ts.append(time.perf_counter() - t0)
                p1 = pool_counts(gpu)
                out = np.empty(n, np.uint32)
                k0.downloadWithStagingBuffer(out)
                ok = bool(np.array_equal(out, ref))
                bad_total += 0 if ok else 1
                line += f" | pool={mode}{'m' if misplace else ''}{'a' if async_ else ''}: exact={ok} took={p1[0] - p0[0]} refused={p1[1] - p0[1]} min={min(ts) * 1e3:.3f}ms"
                if not ok:
                    bad = np.flatnonzero(out != ref)
                    line += f" FIRST BAD {bad[0]} of {bad.size}"
            gpu.setTuning(capi.VRS_TUNE_DEBUG_MISPLACE_STREAMS, 0)
            gpu.setTuning(capi.VRS_TUNE_ASYNC_SORT, 0)
            print(line, flush=True)
            for b in (src, k0, k1):
                b.release()
    print("BAD CASES:", bad_total, flush=True)


def run_time(n, reps):
    keys = [np.random.RandomState(s).randint(0, 2 ** 32, size=n, dtype=np.uint32) for s in (1, 2, 3)]
    with vrs.GPUContext(0) as gpu:
        lib = gpu.lib
        src = [vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), k) for k in keys]
        bat = [vrs.Buffer(gpu, S(4 * n)) for _ in range(reps)]
        k1 = vrs.Buffer(gpu, S(4 * n))
        gpu.setTuning(capi.VRS_TUNE_MSD_POOL_MIN_KEYS, 1 << 22)  # (mode 0: whatever the library takes without the pool form)
        for mode in (2, 0, 2, 0):
            gpu.setTuning(capi.VRS_TUNE_MSD_POOL, mode)
            for timed in (False, True, True):
                for i, b in enumerate(bat):
                    b.copyFrom(src[i % 3])
                gpu.waitIdle()
                t0 = time.perf_counter()
                for b in bat:
                    gpu.check(lib.vrs_sort_keys_u32(gpu.handle, b.handle, k1.handle, n))
                gpu.waitIdle()
                dt = (time.perf_counter() - t0) / reps
                if timed:
                    print(f"n={n} pool={mode}: {dt * 1e3:.4f} ms/sort back to back over {reps} batches  ({n / dt / 1e9:.1f} Gkeys/s)", flush=True)
            # per-kernel breakdown (every launch instrumented: a few per cent slower)
            for i, b in enumerate(bat):
                b.copyFrom(src[i % 3])
            gpu.profileReset()
            gpu.profileEnable(True)
            for b in bat:
                gpu.check(lib.vrs_sort_keys_u32(gpu.handle, b.handle, k1.handle, n))
            gpu.waitIdle()
            gpu.profileEnable(False)
            line = "   "
            for kid, name in capi.KERNEL_NAMES.items():
                cnt, ms = gpu.profileQuery(kid)
                if cnt:
                    line += f" {name} {ms / cnt * 1e3:.1f}us x{cnt} |"
            print(line, pool_counts(gpu), flush=True)
            ok = all(b.verifyKeys(n)[0] == 0 for b in bat)
            out = np.empty(n, np.uint32)
            bat[0].downloadWithStagingBuffer(out)
            print("    sorted:", ok, "exact:", bool(np.array_equal(out, np.sort(keys[0]))), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "time":
        run_time(int(float(sys.argv[2])) if len(sys.argv) > 2 else 10 ** 8, int(sys.argv[3]) if len(sys.argv) > 3 else 10)
    else:
        run_cases()
