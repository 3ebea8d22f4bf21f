"""Randomised end-to-end check on the GPU: random N, B, key width, distribution, pairs, ranking method, against numpy.
Every third case goes through the one-call sort with a random threshold (one counting read + look-back scatter
passes above it), on a sub-range of a larger allocation at a random 4-byte alignment.
   python tools/fuzz_gpu.py [seconds] [seed]"""
import ctypes
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vkradixsort_amd as vrs  # noqa: E402
from vkradixsort_amd import capi  # noqa: E402

S = vrs.Buffer.BufferSettings


def make_keys(rs, n, bits64):
    kind = rs.randint(0, 11)
    if bits64:
        k = (rs.randint(0, 2 ** 32, n, dtype=np.uint64) << np.uint64(32)) | rs.randint(0, 2 ** 32, n, dtype=np.uint64)
        full = np.uint64(0xFFFFFFFFFFFFFFFF)
    else:
        k = rs.randint(0, 2 ** 32, n, dtype=np.uint32)
        full = np.uint32(0xFFFFFFFF)
    if kind == 1:
        k = k & k.dtype.type(0xFF00FF if not bits64 else 0xFF000000FF0000FF)
    elif kind == 2:
        k = np.sort(k)
    elif kind == 3:
        k = np.sort(k)[::-1].copy()
    elif kind == 4:
        k[:] = k[0]
    elif kind == 5:
        k = k >> k.dtype.type(rs.randint(1, 30))
    elif kind == 6 and n > 10:
        k[rs.randint(0, n, max(1, n // 10))] = full
    elif kind == 8:  # one byte constant (some pass of the one-call sort must fall back)
        k = k & ~(k.dtype.type(0xFF) << k.dtype.type(8 * rs.randint(0, k.dtype.itemsize)))
    elif kind == 9:  # a heavy group of 8 digit values in one byte: unbalanced look-back streams
        byte = k.dtype.type(8 * rs.randint(0, k.dtype.itemsize))
        heavy = rs.randint(0, 100, n) < rs.randint(20, 90)
        k = np.where(heavy, (k & ~(k.dtype.type(0xF8) << byte)) | (k.dtype.type(8 * rs.randint(0, 32)) << byte), k).astype(k.dtype)
    elif kind == 10:  # few distinct keys
        k = rs.choice(k[:max(1, min(n, rs.randint(1, 50)))], n).astype(k.dtype)
    return k, kind


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
    rs = np.random.RandomState(seed)
    t_end = time.time() + budget
    cases = 0
    recent = []  # the last few cases, printed with a mismatch: what ran before matters (status words, counters, adaptive counts)
    with vrs.GPUContext(0) as ctx:
        lib = ctx.lib
        knobs = {}
        set_tuning = ctx.setTuning

        def recording_set_tuning(key, value):
            knobs[key] = value
            set_tuning(key, value)
        ctx.setTuning = recording_set_tuning
        while time.time() < t_end:
            if rs.randint(0, 12) == 0:
                # round 3: the second half of the hybrid form for keys that arrive grouped by top byte (what a rank of a large
                # multi-GPU sort receives): random run of top bytes, some of them empty or heavy, any inner order
                first = int(rs.randint(0, 256))
                T = int(rs.randint(1, 257 - first))
                n = int(rs.choice([rs.randint(1, 5000), rs.randint(1, 300000), rs.randint(70000, 6000000)]))
                low, kind = make_keys(rs, n, False)
                tops = rs.randint(first, first + T, n)
                if rs.randint(0, 3) == 0 and T > 2:  # a few top bytes only
                    tops = rs.choice(rs.randint(first, first + T, max(1, T // 5)), n)
                if rs.randint(0, 4) == 0:  # one heavy top byte
                    tops = np.where(rs.randint(0, 100, n) < 60, first + T - 1, tops)
                keys = (low & np.uint32(0x00FFFFFF)) | (tops.astype(np.uint32) << np.uint32(24))
                grouped = keys[np.argsort(keys >> np.uint32(24), kind="stable")]
                gb = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), grouped)
                ob = vrs.Buffer(ctx, S(4 * n))
                ctx.setTuning(capi.VRS_TUNE_MSD_RESERVE, int(rs.choice([0, 2])))
                ctx.setTuning(capi.VRS_TUNE_RANK_MODE, 0)  # the form's local sort needs the LDS-atomic ranking (an earlier case may have switched it off)
                ctx.check(lib.vrs_msd_finish_grouped_u32(ctx.handle, gb.handle, ob.handle, n, first, T))
                took = ctypes.c_int(-1)
                ctx.check(lib.vrs_msd_finish_status(ctx.handle, ctypes.byref(took)))
                res = np.empty(n, np.uint32)
                (ob if took.value else gb).downloadWithStagingBuffer(res)
                ok = np.array_equal(res, np.sort(keys)) if took.value else np.array_equal(res, grouped)
                gb.release()
                ob.release()
                cases += 1
                if not ok:
                    print(f"MISMATCH grouped finish n={n} first={first} T={T} kind={kind} took={took.value} seed={seed} case={cases}")
                    sys.exit(1)
                continue
            bits64 = bool(rs.randint(0, 3) == 0)
            pairs = bool(rs.randint(0, 3) == 0)
            n = int(rs.choice([rs.randint(1, 300), rs.randint(1, 20000), rs.randint(1, 3000000)]))
            B = int(rs.choice([1, 2, 3, 4, 7, 8, 16, 31, 32, 33, 64, 96, 128, 1024, 4096]))
            if n // (B * 256) > 60000:
                B = 32
            mode = int(rs.choice([1, 2]))
            ctx.setTuning(capi.VRS_TUNE_RANK_MODE, mode)
            ctx.setTuning(capi.VRS_TUNE_FUSED_PREFIX, int(rs.randint(0, 2)))
            ctx.setTuning(capi.VRS_TUNE_XCD_REMAP, int(rs.randint(0, 2)))
            one_call = rs.randint(0, 3) == 0
            if one_call:
                n = int(rs.choice([rs.randint(1, 20000), rs.randint(1, 3000000), rs.randint(1000000, 9000000), rs.randint(4200000, 12000000)]))
                if not bits64 and not pairs and rs.randint(0, 12) == 0:
                    n = int(rs.randint(30000000, 45000000))  # buckets beyond one wave: the 256-thread local sort
            sparse = one_call and (bits64 or pairs) and rs.randint(0, 10) == 0
            if sparse:
                n = int(rs.randint(28000000, 52000000))
            keys, kind = make_keys(rs, n, bits64)
            if sparse:
                # three of four top-14-bit buckets empty: the others hold 6800 to 12700 elements -- the 1024-thread local sort of
                # pairs and 64-bit keys (uniform keys only: other kinds have their own bucket shapes)
                keys = keys & ~(keys.dtype.type(3) << keys.dtype.type(8 * keys.itemsize - 14))
            vals = rs.randint(0, 2 ** 32, n, dtype=np.uint32) if pairs else None
            if one_call:
                ctx.setTuning(capi.VRS_TUNE_ONE_CALL_MIN_KEYS, int(rs.choice([0, 1, n, max(1, n // 2), 1 << 20, capi.ONE_CALL_MIN_KEYS_DEFAULT])))
                # round 2 knobs: group count of the counting read, fused plan, single-launch threshold, and (rarely) a
                # withheld look-back tile with a small spin budget -- none of them may change a result
                ctx.setTuning(capi.VRS_TUNE_DIGIT_TABLE_GROUPS, int(rs.choice([0, 8, 16, 32])))
                ctx.setTuning(capi.VRS_TUNE_FUSED_PLAN, int(rs.randint(0, 2)))
                ctx.setTuning(capi.VRS_TUNE_SINGLE_MAX_KEYS, int(rs.choice([0, 4096, 4096, 20000])))
                ctx.setTuning(capi.VRS_TUNE_HYBRID, int(rs.randint(0, 4) != 0))
                ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, int(rs.choice([1 << 22, 1 << 22, capi.HYBRID_MIN_KEYS_DEFAULT])))
                ctx.setTuning(capi.VRS_TUNE_HYBRID_FAST_COUNT, int(rs.randint(0, 3)))
                ctx.setTuning(capi.VRS_TUNE_MSD_RESERVE, int(rs.choice([0, 1, 2, 2])))  # MSD passes by reservation at any size
                # round 4: the pool form (no counting read) at any size the hybrid form may take, off, adaptive or always; odd tiles of its
                # first pass off their slices now and then
                ctx.setTuning(capi.VRS_TUNE_MSD_POOL, int(rs.choice([0, 1, 2, 2])))
                ctx.setTuning(capi.VRS_TUNE_MSD_POOL_MIN_KEYS, int(rs.choice([1 << 22, 1 << 22, 32000000])))
                ctx.setTuning(capi.VRS_TUNE_DEBUG_MISPLACE_STREAMS, int(rs.randint(0, 6) == 0))
                # round 3: enqueue-only sorts (the download below settles them)
                ctx.setTuning(capi.VRS_TUNE_ASYNC_SORT, int(rs.randint(0, 3) == 0))
                hold = rs.randint(0, 8) == 0
                recent.append(("one-call", n, "u64" if bits64 else "u32", "pairs" if pairs else "keys", kind,
                               dict(knobs)))
                recent[:] = recent[-4:]
                ctx.setTuning(capi.VRS_TUNE_DEBUG_HOLD_TILE, int(rs.randint(0, 6)) if hold else -1)
                ctx.setTuning(capi.VRS_TUNE_LOOKBACK_SPIN_BUDGET, int(rs.choice([0, 3, 40])) if hold else 4096)
                kb = keys.itemsize
                off = int(rs.randint(0, 16 // kb))
                g0, g1 = keys.dtype.type(0x11111111), keys.dtype.type(0x22222222)
                big = vrs.Buffer(ctx, S(kb * (n + 8)))
                host = np.concatenate([np.full(off, g0, keys.dtype), keys, np.full(8 - off, g1, keys.dtype)])
                ctx.check(lib.vrs_buffer_upload(ctx.handle, big.handle, host.ctypes.data_as(ctypes.c_void_p), host.nbytes))
                k0 = vrs.Buffer(ctx, S(kb * n), device_ptr=big.getDeviceAddress() + kb * off)
                k1 = vrs.Buffer(ctx, S(kb * n))
                if pairs:
                    v0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), vals)
                    v1 = vrs.Buffer(ctx, S(4 * n))
                    sort_pairs = lib.vrs_sort_pairs_u64 if bits64 else lib.vrs_sort_pairs_u32
                    ctx.check(sort_pairs(ctx.handle, k0.handle, k1.handle, v0.handle, v1.handle, n))
                elif bits64:
                    ctx.check(lib.vrs_sort_keys_u64(ctx.handle, k0.handle, k1.handle, n))
                else:
                    ctx.check(lib.vrs_sort_keys_u32(ctx.handle, k0.handle, k1.handle, n))
                ctx.check(lib.vrs_buffer_download(ctx.handle, big.handle, host.ctypes.data_as(ctypes.c_void_p), host.nbytes))
                order = np.argsort(keys, kind="stable")
                ok = np.array_equal(host[off:off + n], keys[order]) and (host[:off] == g0).all() and \
                    (host[off + n:] == g1).all()
                if pairs:
                    ov = np.empty(n, np.uint32)
                    v0.downloadWithStagingBuffer(ov)
                    ok = ok and np.array_equal(ov, vals[order])
                    v0.release()
                    v1.release()
                for b in (k0, k1, big):
                    b.release()
                ctx.setTuning(capi.VRS_TUNE_ONE_CALL_MIN_KEYS, capi.ONE_CALL_MIN_KEYS_DEFAULT)
                ctx.setTuning(capi.VRS_TUNE_MSD_RESERVE, 1)
                ctx.setTuning(capi.VRS_TUNE_MSD_POOL, 1)
                ctx.setTuning(capi.VRS_TUNE_MSD_POOL_MIN_KEYS, 32000000)
                ctx.setTuning(capi.VRS_TUNE_DEBUG_MISPLACE_STREAMS, 0)
                ctx.setTuning(capi.VRS_TUNE_ASYNC_SORT, 1)
                ctx.setTuning(capi.VRS_TUNE_DEBUG_HOLD_TILE, -1)
                ctx.setTuning(capi.VRS_TUNE_LOOKBACK_SPIN_BUDGET, 4096)
                cases += 1
                if not ok:
                    print(f"MISMATCH one-call n={n} bits64={bits64} pairs={pairs} kind={kind} mode={mode} off={off} seed={seed} case={cases} hold={hold}")
                    for r in recent:
                        print("   recent:", r)
                    sys.exit(1)
                continue
            kb = 8 if bits64 else 4
            W = lib.vrs_workgroup_count(n, B)
            kbuf = [vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(kb * n), keys), vrs.Buffer(ctx, S(kb * n))]
            vbuf = [vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), vals), vrs.Buffer(ctx, S(4 * n))] if pairs else None
            h = vrs.Buffer(ctx, S(W * 1024))
            hist_fn = lib.vrs_multi_radixsort_histograms_u64 if bits64 else lib.vrs_multi_radixsort_histograms
            for i in range(kb):
                pc = vrs.PushConstants(n, 8 * i, W, B)
                ctx.check(hist_fn(ctx.handle, kbuf[i % 2].handle, h.handle, ctypes.byref(pc)))
                if pairs:
                    fn = lib.vrs_multi_radixsort_pairs_u64 if bits64 else lib.vrs_multi_radixsort_pairs
                    ctx.check(fn(ctx.handle, kbuf[i % 2].handle, kbuf[(i + 1) % 2].handle, vbuf[i % 2].handle,
                                 vbuf[(i + 1) % 2].handle, h.handle, ctypes.byref(pc)))
                else:
                    fn = lib.vrs_multi_radixsort_u64 if bits64 else lib.vrs_multi_radixsort
                    ctx.check(fn(ctx.handle, kbuf[i % 2].handle, kbuf[(i + 1) % 2].handle, h.handle, ctypes.byref(pc)))
            out = np.empty(n, keys.dtype)
            kbuf[0].downloadWithStagingBuffer(out)
            order = np.argsort(keys, kind="stable")
            ok = np.array_equal(out, keys[order])
            if pairs:
                ov = np.empty(n, np.uint32)
                vbuf[0].downloadWithStagingBuffer(ov)
                ok = ok and np.array_equal(ov, vals[order])
            for b in kbuf + (vbuf or []) + [h]:
                b.release()
            cases += 1
            if not ok:
                print(f"MISMATCH n={n} B={B} bits64={bits64} pairs={pairs} kind={kind} mode={mode} seed={seed} case={cases}")
                sys.exit(1)
    print(f"fuzz ok: {cases} random cases in {budget:.0f} s (seed {seed})")


if __name__ == "__main__":
    main()
