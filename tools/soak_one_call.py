"""Soak of the one-call sort's look-back path: many large sorts, verified ON the device (ascending + the same
order-independent fingerprint as the input), so timing-dependent faults of the inter-workgroup hand-off would show.
   python tools/soak_one_call.py [seconds] [seed] [keys|pairs|u64|misplaced]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vkradixsort_amd as vrs  # noqa: E402
from vkradixsort_amd import capi  # noqa: E402


def fingerprint(t):
    u = t.to(torch.int64) & 0xFFFFFFFF
    return int(u.sum().item()), int(((u & 0xFFFF) * (u >> 16)).sum().item()), int((u * 2654435761 & 0xFFFFFFFF).sum().item())


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    mode = sys.argv[3] if len(sys.argv) > 3 else "keys"
    cases, keys_sorted, bad = soak(budget, seed, mode)
    if bad:
        print(bad)
        sys.exit(1)
    print(f"soak ok ({mode}): {cases} sorts, {keys_sorted / 1e9:.1f} G keys in {budget:.0f} s (seed {seed})")


def soak(budget, seed=1, mode="keys", max_keys=10 ** 8):
    """Sorts for `budget` seconds; returns (sorts, keys sorted, None or a description of the first mismatch)."""
    torch.manual_seed(seed)
    rs = np.random.RandomState(seed)
    dev = torch.device("cuda", 0)
    S = vrs.Buffer.BufferSettings
    flip = torch.tensor(-2 ** 31, dtype=torch.int32, device=dev)
    cases = keys_sorted = 0
    t_end = time.time() + budget
    with vrs.GPUContext(0, stream=torch.cuda.current_stream().cuda_stream) as gpu:
        gpu.setTuning(capi.VRS_TUNE_DEBUG_MISPLACE_STREAMS, 1 if mode == "misplaced" else 0)
        gpu.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 1 << 22)  # the hybrid form wherever its buckets fit, not only from 4e7 keys on
        import os
        if os.environ.get("VRS_SOAK_REUSE_LAYOUT") == "0":
            gpu.setTuning(capi.VRS_TUNE_MSD_POOL_REUSE_LAYOUT, 0)
        if os.environ.get("VRS_SOAK_POOL") == "0":
            gpu.setTuning(capi.VRS_TUNE_MSD_POOL, 0)
        while time.time() < t_end:
            n = int(rs.choice([rs.randint(1 << 20, 1 << 23), rs.randint(1 << 23, 6 * 10 ** 7), 10 ** 8]))
            n = min(n, max_keys)
            if mode == "misplaced":
                n = int(rs.randint(1 << 20, 1 << 22))  # every other tile re-counts its stream's prefix: slow
            if mode in ("pairs", "u64"):
                n = min(n, 5 * 10 ** 7)
            if mode == "u64":
                hi = torch.randint(-2 ** 31, 2 ** 31, (n,), dtype=torch.int64, device=dev)
                lo = torch.randint(0, 2 ** 32, (n,), dtype=torch.int64, device=dev)
                k = (hi << 32) | lo
                if rs.randint(0, 3) == 0:
                    k = k >> int(rs.randint(1, 40))
                tmp = torch.empty_like(k)
                s0 = int(k.sum().item()), int((k >> 17).sum().item())
                torch.cuda.synchronize()
                k0 = vrs.Buffer(gpu, S(8 * n), device_ptr=k.data_ptr())
                k1 = vrs.Buffer(gpu, S(8 * n), device_ptr=tmp.data_ptr())
                gpu.check(gpu.lib.vrs_sort_keys_u64(gpu.handle, k0.handle, k1.handle, n))
                gpu.waitIdle()
                u = k ^ (-2 ** 63)  # unsigned order on int64 storage
                ok = bool((u[1:] >= u[:-1]).all().item()) and (int(k.sum().item()), int((k >> 17).sum().item())) == s0
                k0.release()
                k1.release()
                cases += 1
                keys_sorted += n
                if not ok:
                    return cases, keys_sorted, f"SOAK MISMATCH u64 n={n} seed={seed} case={cases}"
                del k, tmp, u, hi, lo
                continue
            kind = rs.randint(0, 6)
            k = torch.randint(-2 ** 31, 2 ** 31, (n,), dtype=torch.int64, device=dev).to(torch.int32)
            if kind == 1:
                k = k >> int(rs.randint(1, 20))  # arithmetic shift: sign-extended clusters at both ends
            elif kind == 2:
                k = torch.sort(k)[0]
            elif kind == 3:
                k = k & int(rs.choice([0x00FFFFFF, 0x0FFFFFFF, 0x7FFFFF00, 0x7FF0FFFF]))
            elif kind == 4:
                k = (k & 0x7FFF07FF) | 0x2000  # one heavy stream in pass 2
            tmp = torch.empty_like(k)
            before = fingerprint(k)
            if mode == "pairs":
                src = k.clone()
                v = torch.arange(n, dtype=torch.int32, device=dev)
                vt = torch.empty_like(v)
                torch.cuda.synchronize()
                bufs = [vrs.Buffer(gpu, S(4 * n), device_ptr=t.data_ptr()) for t in (k, tmp, v, vt)]
                gpu.check(gpu.lib.vrs_sort_pairs_u32(gpu.handle, *[b.handle for b in bufs], n))
                gpu.waitIdle()
                u = k ^ flip
                vl = v.to(torch.int64)
                same = u[1:] == u[:-1]
                ok = bool((u[1:] >= u[:-1]).all().item()) and bool((src[vl] == k).all().item()) and \
                    bool((vl[1:][same] > vl[:-1][same]).all().item())
                for b in bufs:
                    b.release()
                cases += 1
                keys_sorted += n
                if not ok:
                    return cases, keys_sorted, f"SOAK MISMATCH pairs n={n} kind={kind} seed={seed} case={cases}"
                del k, tmp, v, vt, src, u, vl, same
                continue
            torch.cuda.synchronize()
            k0 = vrs.Buffer(gpu, S(4 * n), device_ptr=k.data_ptr())
            k1 = vrs.Buffer(gpu, S(4 * n), device_ptr=tmp.data_ptr())
            reps = int(rs.randint(1, 4))
            for r in range(reps):  # re-sorting sorted output is a legal (and differently timed) input
                gpu.check(gpu.lib.vrs_sort_keys_u32(gpu.handle, k0.handle, k1.handle, n))
            gpu.waitIdle()
            u = k ^ flip
            ok = bool((u[1:] >= u[:-1]).all().item()) and fingerprint(k) == before
            k0.release()
            k1.release()
            cases += 1
            keys_sorted += n * reps
            if not ok:
                return cases, keys_sorted, f"SOAK MISMATCH n={n} kind={kind} seed={seed} case={cases}"
            del k, tmp, u
    return cases, keys_sorted, None


if __name__ == "__main__":
    main()
