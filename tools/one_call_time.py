"""Timing of the one-call sort (vrs_sort_keys_u32 / vrs_sort_pairs_u32): one-read path vs the four contract passes.
usage: one_call_time.py [N] [reps] [dist] [pairs]      dist: uniform | 28bit | mult256 | sorted | const"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import os  # noqa: E402
import ctypes  # noqa: E402
from vkradixsort_amd import capi  # noqa: E402
if os.environ.get("VRS_LIB"):  # lab builds of the library (e.g. another stream count)
    capi.LIB_PATH = Path(os.environ["VRS_LIB"]).resolve()
import vkradixsort_amd as vrs  # noqa: E402


def make(n, dist):
    rs = np.random.RandomState(1)
    k = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32)
    if dist == "28bit":
        k >>= 4
    elif dist == "mult256":
        k &= np.uint32(0xFFFFFF00)
    elif dist == "sorted":
        k.sort()
    elif dist == "const":
        k[:] = 0xDEADBEEF
    elif dist == "lowbyte":
        k &= np.uint32(0xFF)
    return k


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 8
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dist = sys.argv[3] if len(sys.argv) > 3 else "uniform"
    pairs = len(sys.argv) > 4 and sys.argv[4] == "pairs"
    u64 = len(sys.argv) > 4 and sys.argv[4] == "u64"
    keys = make(n, dist)
    if u64:
        keys = (keys.astype(np.uint64) << np.uint64(32)) | make(n, "uniform")[::-1].astype(np.uint64)
    ref = np.sort(keys, kind="stable")
    vals = np.arange(n, dtype=np.uint32)
    with vrs.GPUContext(0) as gpu:
        lib = gpu.lib
        S = vrs.Buffer.BufferSettings(keys.itemsize * n)
        src = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S, keys)
        k0, k1 = vrs.Buffer(gpu, S), vrs.Buffer(gpu, S)
        if pairs:
            vsrc = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S, vals)
            v0, v1 = vrs.Buffer(gpu, S), vrs.Buffer(gpu, S)
        gpu.setTuning(capi.VRS_TUNE_DEBUG_MISPLACE_STREAMS, int(os.environ.get("VRS_MISPLACE", "0")))
        if os.environ.get("VRS_GROUPS"):
            gpu.setTuning(capi.VRS_TUNE_DIGIT_TABLE_GROUPS, int(os.environ["VRS_GROUPS"]))
        for min_keys in ((1,) if os.environ.get("VRS_ONLY_ONE_READ") else (0, 1)):
            gpu.setTuning(capi.VRS_TUNE_ONE_CALL_MIN_KEYS, min_keys)
            times = []
            for r in range(reps + 2):
                k0.copyFrom(src)
                if pairs:
                    v0.copyFrom(vsrc)
                gpu.waitIdle()
                if r == 2 and not os.environ.get("VRS_NO_PROFILE"):
                    gpu.profileReset()
                    gpu.profileEnable(True)
                t0 = time.perf_counter()
                if pairs:
                    rc = lib.vrs_sort_pairs_u32(gpu.handle, k0.handle, k1.handle, v0.handle, v1.handle, n)
                elif u64:
                    rc = lib.vrs_sort_keys_u64(gpu.handle, k0.handle, k1.handle, n)
                else:
                    rc = lib.vrs_sort_keys_u32(gpu.handle, k0.handle, k1.handle, n)
                if rc and not os.environ.get("VRS_DT_DEBUG"):
                    gpu.check(rc)
                gpu.waitIdle()
                if r >= 2:
                    times.append(time.perf_counter() - t0)
            gpu.profileEnable(False)
            out = np.empty(n, dtype=keys.dtype)
            k0.downloadWithStagingBuffer(out)
            ok = bool(np.array_equal(out, ref))
            if pairs:
                vo = np.empty(n, dtype=np.uint32)
                v0.downloadWithStagingBuffer(vo)
                ok = ok and bool(np.array_equal(vo, np.argsort(keys, kind="stable").astype(np.uint32)))
            t = min(times)
            tag = os.environ.get("VRS_TAG", "")
            line = f"{tag} N={n} {dist} {'pairs' if pairs else 'u64 keys' if u64 else 'keys'} one_read={'on' if min_keys else 'off'} exact={ok} min={t*1e3:.3f}ms med={np.median(times)*1e3:.3f}ms {n/t/1e9:.2f} G/s"
            for kid, name in capi.KERNEL_NAMES.items():
                cnt, ms = gpu.profileQuery(kid)
                if cnt:
                    line += f" | {name}: {ms/cnt*1e3:.1f}us x{cnt//reps}"
            cnt, _ = gpu.profileQuery(capi.VRS_KERNEL_LOOKBACK_SCATTER)
            if cnt >= 4:
                per = []
                for i in range(cnt - 4, cnt):
                    ms = ctypes.c_double()
                    gpu.check(lib.vrs_profile_query_launch(gpu.handle, capi.VRS_KERNEL_LOOKBACK_SCATTER, i, ctypes.byref(ms)))
                    per.append(f"{ms.value*1e3:.0f}")
                line += " | last sort's passes: " + "/".join(per)
            print(line, flush=True)


if __name__ == "__main__":
    main()
