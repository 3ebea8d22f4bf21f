#!/usr/bin/env python3
"""bench.py -- Gkeys/s sorting uint32 keys with the MI355X-native multi_radixsort path.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A step = one complete sort of one batch of synthetic keys that is already resident in HBM when the timed region starts.
N = 1 times the library's one-call sort (vrs_sort_keys_u32; at 10^8 keys its hybrid form: one counting read, two MSD
look-back scatter passes, the LDS-local bucket sort -- 28 B/key; below 1.3e7 keys the LSD form: one counting read + four
look-back scatter passes, 36 B/key) and reports the reference's stage-by-stage contract path (4 x [histograms, prefix,
scatter], 48 B/key) beside it; --path contract swaps the two.  N = 1 sorts BASELINE.json configs[2]: 10^8 uniform random
uint32 (std::mt19937 raw outputs, seeds 1/2/3 cycled over the K pre-staged batches); --n 1e7 is configs[1], --pairs
configs[3] (key + payload pairs).  N > 1 sorts N x 10^8 keys sharded by key range (configs[4] at N = 8) through
vrs_dist_sort_keys_u32: the hybrid sort with the RCCL all-to-all between its two MSD passes.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy achieves
BYTES_PER_KEY_SORT = 48  # 4 passes x (histogram read 4 + scatter read 4 + scatter write 4)   SURVEY.md section 8d
BYTES_PER_KEY_SORT_ONE_READ = 36  # one counting read 4 + 4 passes x (scatter read 4 + scatter write 4): SURVEY.md section 8d's rule
BYTES_PER_KEY_SORT_HYBRID = 28  # one counting read 4 + 2 MSD scatter passes x 8 + the LDS-local bucket sort (read 4 + write 4)
BYTES_PER_KEY_SORT_POOL = 24  # the hybrid form without the counting read (vrs_msd_pool.hip): 2 MSD passes x 8 + the local sort 8
#                               (+ 0.125: the sample reads 1/32 of the keys -- not counted, like the table traffic of the other forms)
BYTES_PER_KEY_SCATTER = 8  # a scatter pass, per launch: read 4 + write 4
# algorithmic bytes per key and launch of every byte-moving kernel (the library's profile names, vkradixsort_amd/capi.py KERNEL_NAMES)
KERNEL_BYTES_PER_KEY = {"histogram": 4, "scatter": 8, "digit_tables": 4, "lookback_scatter": 8, "local_sort": 8, "pool_pass_a": 8, "pool_pass_b": 8}
KERNEL_WHAT = {
    "histogram": "histogram_kernel: the contract's RADIX_SORT_HISTOGRAMS stage, reads every key once per pass",
    "scatter": "scatter_kernel: the contract's stable scatter, reads and writes every key once per pass",
    "digit_tables": "digit_tables_kernel: the one counting read of the counted one-call forms",
    "lookback_scatter": "onesweep_scatter_kernel / msd_pass_b_kernel: a scatter pass of the one-call sort's counted forms, reads and writes every key once",
    "local_sort": "the LDS-local sort of every bucket (pool form: pool_local_sort_kernel; counted form: msd_local_sort_keys_kernel), reads and writes every key once; at 10^8 keys its memory skeleton alone (the same kernel without its two LDS passes) takes 147-153 us, the passes add 25-33 (profiles/labs/r06_local_sort_skeleton.txt)",
    "pool_pass_a": "pool_pass_a_kernel: the pool form's first MSD pass (reserves in sampled regions), reads and writes every key once",
    "pool_pass_b": "pool_pass_b_kernel: the pool form's second MSD pass (scatters into the buckets' slack regions), reads and writes every key once",
}


def pick_dominant(kernels: dict) -> str | None:
    """The dominant kernel of a timed path, BY KERNEL NAME: the byte-moving kernel with the largest share of the instrumented
    time -- launches x average launch duration (kernels: name -> {"launches", "avg_us"} of one instrumented run).  Ties go
    to the kernel that moves more bytes per launch, then to the name.  None if nothing byte-moving was timed."""
    best, best_key = None, None
    for name, rec in kernels.items():
        if name not in KERNEL_BYTES_PER_KEY or not rec.get("avg_us") or not rec.get("launches"):
            continue
        key = (rec["launches"] * rec["avg_us"], KERNEL_BYTES_PER_KEY[name], name)
        if best_key is None or key > best_key:
            best, best_key = name, key
    return best


def mt19937_keys(seed: int, n: int) -> np.ndarray:
    # numpy's RandomState is MT19937 with the same seeding as std::mt19937(seed); 32-bit draws are the raw outputs
    return np.random.RandomState(seed).randint(0, 2 ** 32, size=n, dtype=np.uint32)


def pool_path_text(lib, n: int, top_bits_setting: int = 0, pairs: bool = False) -> str:
    """config.path of a line whose timed sorts took the pool form, built from what the LIBRARY reports for this size and cut
    (vrs_pool_form_shape_ex: host only) -- round 5's line still said "8 bits ... 6 bits" after the default became 7 + 7
    (tests/test_tools_cpu.py fails when the text and the library disagree)."""
    import ctypes
    a, b, cap, scratch = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint64()
    if lib.vrs_pool_form_shape_ex(n, int(pairs), int(top_bits_setting), ctypes.byref(a), ctypes.byref(b), ctypes.byref(cap), ctypes.byref(scratch)) != 0 or a.value == 0:
        raise SystemExit(f"vrs_pool_form_shape_ex has no pool shape for n = {n}")
    low = 32 - a.value - b.value
    local = "one wave per bucket" if cap.value == 1789 else f"one {1024 if cap.value == 13312 else 512 if cap.value in (14333, 6656) else 256}-thread workgroup per bucket"
    what = "vrs_sort_pairs_u32, the STABLE pool form (a tile's place in a region is its rank there: decoupled look-back)" if pairs else "vrs_sort_keys_u32, pool form"
    return (f"{what} -- the hybrid form without a counting read: a sample of 1/32 of the keys sizes a region per (input slice, first digit); "
            f"the first MSD pass ({a.value} bits) {'places' if pairs else 'reserves'} its output there, the second ({b.value} bits: {1 << (a.value + b.value)} buckets) "
            f"scatters into per-bucket regions of a context-owned slack buffer ({scratch.value / 1e6:.0f} MB of context scratch), the local sort ({local}, up to {cap.value} "
            f"{'pairs' if pairs else 'keys'}) reads every bucket in one piece, sorts it by its low {low} bits inside LDS and writes it to its final place "
            f"({48 if pairs else 24} B/{'pair' if pairs else 'key'})")


def load_traffic_profile(kernel: str = "scatter", algorithmic_bytes: float = 0.0):
    """HBM bytes per launch of the dominant kernel(s) from the committed rocprofv3 --pmc passes (profiles/), if present:
    (bytes per launch of the first dominant kernel or None, detail).  A figure below half the algorithmic bytes cannot be a
    pass over the keys (round 3 committed the 15 KiB of a launch that left at once): refused, with a note."""
    p = ROOT / "profiles" / f"{kernel}_traffic.json"
    if not p.exists():
        return None, {"note": f"profiles/{kernel}_traffic.json is absent"}
    try:
        rec = json.loads(p.read_text())
    except Exception as e:  # noqa: BLE001
        return None, {"note": f"profiles/{kernel}_traffic.json is unreadable: {e}"}
    passes = rec.get("passes") or [rec]
    detail = {"round": rec.get("round"), "commit": rec.get("commit"), "box": rec.get("box"), "passes": []}
    for q in passes:
        b = q.get("hbm_bytes_per_launch")
        ratio = (b / algorithmic_bytes) if (b and algorithmic_bytes) else None
        detail["passes"].append({"kernel": q.get("instantiation", q.get("kernel")), "hbm_bytes_per_launch": b, "dispatches": q.get("dispatches"),
                                 "ratio_to_algorithmic": round(ratio, 3) if ratio else None})
    first = detail["passes"][0]["hbm_bytes_per_launch"] if detail["passes"] else None
    if algorithmic_bytes and any((q["hbm_bytes_per_launch"] or 0) < 0.5 * algorithmic_bytes for q in detail["passes"]):
        detail["note"] = "refused: a committed figure is below half the algorithmic bytes per launch -- not a pass over the keys"
        return None, detail
    return first, detail


def cpu_baseline(host_keys):
    """The reference's verification path (single-threaded std::sort, MultiRadixSort.cpp:141-146) timed on this
    host through the oracle library: every distinct batch (seeds 1, 2, 3) is sorted once -- the results are the
    bit-exact references of the GPU outputs, the times are the baseline.  Only this leg of bench.py touches oracle/."""
    from tests import _oracle
    orc = _oracle.load()
    refs, times = [], []
    for keys in host_keys:
        ref, ms = orc.std_sort(keys)
        refs.append(ref)
        times.append(ms)
    cores, model = orc.cpu_info()
    n = host_keys[0].size
    best = min(times)
    return refs, {"value": round(n / (best * 1e-3) / 1e9, 5), "unit": "Gkeys/s", "cores": 1, "kind": "port",
                  "sample": f"std::sort of the full {n}-key batches of seeds 1..{len(times)}, one repetition each: "
                            + ", ".join(f"{t:.0f}" for t in times) + " ms (value = the fastest)",
                  "host": f"1 thread of {cores} hardware threads ({model})"}


def bench_single(args):
    import vkradixsort_amd as vrs
    from vkradixsort_amd import capi

    n, B, K, W = args.n, args.blocks, args.steps, args.warmup
    S = vrs.Buffer.BufferSettings
    seeds = [1, 2, 3]
    host_keys = [mt19937_keys(s, n) for s in seeds[:max(1, min(3, K))]]
    gpu = vrs.GPUContext(int(os.environ.get("LOCAL_RANK", "0")))
    gpu.init()
    if args.rank_mode:
        gpu.setTuning(capi.VRS_TUNE_RANK_MODE, args.rank_mode)
    if args.variant:
        gpu.setTuning(capi.VRS_TUNE_SCATTER_VARIANT, args.variant)
    tuned = {}
    for kv in (args.tune or []):  # lab switch: --tune KEY=VALUE (numeric ids of include/vkradixsort_amd.h); named in the line's config
        k_, v_ = kv.split("=")
        gpu.setTuning(int(k_), int(v_))
        tuned[int(k_)] = int(v_)
    dev_name, cus, mem = gpu.deviceInfo()
    nbuf = max(K, W, 1)
    need = (nbuf + len(host_keys) + 1) * 4 * n
    if need > 0.8 * mem:
        raise SystemExit(f"{nbuf} pre-staged batches need {need / 2 ** 30:.0f} GiB; lower --steps")
    pristine = [vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), k) for k in host_keys]
    batches = [vrs.Buffer(gpu, S(4 * n)) for _ in range(nbuf)]
    buf1 = vrs.Buffer(gpu, S(4 * n))
    Wg = gpu.lib.vrs_workgroup_count(n, B)
    hist = vrs.Buffer(gpu, S(Wg * 256 * 4))
    p = vrs.MultiRadixSortPass(gpu)
    p.create()
    gis = n // B + (1 if n % B else 0)
    p.setGlobalInvocationSize(p.RADIX_SORT_HISTOGRAMS, gis, 1, 1)
    p.setGlobalInvocationSize(p.RADIX_SORT, gis, 1, 1)
    for pc in (p.m_pushConstantsHistogram, p.m_pushConstants):
        pc.g_num_elements, pc.g_num_workgroups, pc.g_num_blocks_per_workgroup = n, Wg, B
    p.setStorageBuffer(p.RADIX_SORT_HISTOGRAMS, 1, hist)
    p.setStorageBuffer(p.RADIX_SORT, 2, hist)

    def rearm():
        for i, b in enumerate(batches):
            b.copyFrom(pristine[i % len(pristine)])
        gpu.waitIdle()

    def sort_batch(b0):
        # MultiRadixSort::execute's hot loop (MultiRadixSort.cpp:37-61): ping-pong binding + four passes
        a = gpu.getActiveIndex()
        o = (a + 1) % 2
        H, R = p.RADIX_SORT_HISTOGRAMS, p.RADIX_SORT
        p.setStorageBuffer(a, H, 0, b0)
        p.setStorageBuffer(a, R, 0, b0)
        p.setStorageBuffer(o, R, 1, b0)
        p.setStorageBuffer(o, H, 0, buf1)
        p.setStorageBuffer(a, R, 1, buf1)
        p.setStorageBuffer(o, R, 0, buf1)
        tok = None
        for i in range(4):
            p.m_pushConstantsHistogram.g_shift = 8 * i
            p.m_pushConstants.g_shift = 8 * i
            tok = p.execute(tok)
            gpu.incrementActiveIndex()

    def sort_one_call(b0):
        # the library's own loop over the passes: one counting read, then either four LSD look-back scatter passes (K5) or
        # -- large inputs whose top-14-bit buckets fit a workgroup -- two MSD look-back passes + the LDS-local sort (K5b)
        gpu.check(gpu.lib.vrs_sort_keys_u32(gpu.handle, b0.handle, buf1.handle, n))

    def hybrid_sorts():
        h = ctypes.c_uint64()
        gpu.check(gpu.lib.vrs_one_call_hybrid_sorts(gpu.handle, ctypes.byref(h)))
        return h.value

    def pool_sorts():  # (sorts that took the pool form, sorts whose pool form a verdict refused)
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        gpu.check(gpu.lib.vrs_one_call_pool_sorts(gpu.handle, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def hybrid_recounts():  # sorts that needed a second counting read (fast count, then a refused hybrid form)
        h = ctypes.c_uint64()
        gpu.check(gpu.lib.vrs_one_call_hybrid_recounts(gpu.handle, ctypes.byref(h)))
        return h.value

    one_call = args.path == "one_call"
    # (function, the byte-moving kernels of the path: their launches carry events in the sampled steps of the timed region)
    name_to_id = {v: k for k, v in capi.KERNEL_NAMES.items()}
    paths = {"one_call": (sort_one_call, ["digit_tables", "lookback_scatter", "local_sort", "pool_pass_a", "pool_pass_b"]),
             "contract": (sort_batch, ["histogram", "scatter"])}
    primary, timed_names = paths[args.path]
    timed_mask = sum(1 << name_to_id[x] for x in timed_names)

    def kernel_table():
        t = {}
        for kid, name in capi.KERNEL_NAMES.items():
            cnt, ms = gpu.profileQuery(kid)
            if cnt:
                t[name] = {"launches": cnt, "avg_us": round(ms / cnt * 1e3, 2)}
        return t

    def run_steps(fn, count, mask, every=1):
        """count steps of fn over the pre-staged batches; mask = kernels whose launches carry events (0 = none) in every
        `every`-th step (an event pair costs the launch it brackets a few microseconds: the timed region samples)"""
        gpu.profileReset()
        gpu.profileEnableMask(mask)
        gpu.waitIdle()
        t0 = time.perf_counter()
        for i in range(count):
            if every > 1:
                gpu.profileEnableMask(mask if i % every == 0 else 0)
            fn(batches[i])
        gpu.waitIdle()
        dt = time.perf_counter() - t0
        gpu.profileEnable(False)
        return dt, kernel_table()

    rearm()
    run_steps(primary, W, 0)
    rearm()

    # ---- timed region: exactly K steps, inputs resident.  The byte-moving kernels' launches of every 4th step carry HIP events
    # on their own dispatch packets, on the stream they are launched on; nothing else is instrumented.  The DOMINANT kernel is
    # the one of them with the largest share of that time, by kernel name (pick_dominant).
    hybrid_before, recounts_before, pool_before = hybrid_sorts(), hybrid_recounts(), pool_sorts()
    elapsed, kernels = run_steps(primary, K, timed_mask, every=args.event_every)
    dominant_name = pick_dominant(kernels)
    hybrid_steps = hybrid_sorts() - hybrid_before  # K if every timed one-call sort took a hybrid form (pool or counted), 0 if none did
    recount_steps = hybrid_recounts() - recounts_before
    pool_after = pool_sorts()
    pool_steps, pool_refused = pool_after[0] - pool_before[0], pool_after[1] - pool_before[1]
    # the timed region's own outputs, every one of them, before anything overwrites them (one device read each)
    fingerprints = [p_.verifyKeys(n)[1:] for p_ in pristine]
    timed_bad = [i for i in range(K)
                 if (lambda r: r[0] != 0 or r[1:] != fingerprints[i % len(pristine)])(batches[i].verifyKeys(n))]
    if timed_bad:
        raise SystemExit(f"VERIFICATION FAILED: timed-region outputs of steps {timed_bad} are not sorted permutations of their inputs")

    # ---- outside the timed region: (a) the same K steps with no events at all (instrumentation overhead check),
    # (b) once more with every kernel timed, for the per-kernel breakdown
    rearm()
    unprofiled, _ = run_steps(primary, K, 0)
    rearm()
    _, breakdown = run_steps(primary, K, (1 << capi.VRS_KERNEL_COUNT) - 1)

    # ---- verification, second part: the reruns above sorted the same inputs again (rearm() restores them), so the
    # batches now hold the outputs of the last rerun: every one is checked on the device again (ascending + the
    # multiset fingerprint of its input) and the first batch of every seed bit for bit against std::sort.  The timed
    # region's own K outputs were checked on the device right after the timed region.
    def verify_batches(tag, refs):
        bad = []
        for i in range(K):
            d, sm, mx = batches[i].verifyKeys(n)
            if d != 0 or (sm, mx) != fingerprints[i % len(pristine)]:
                bad.append(i)
        res = {f"{tag}_every_batch_ascending_and_permutation_of_its_input": not bad, f"{tag}_batches_checked_on_device": K}
        if refs is not None:
            exact = True
            for j in range(min(len(pristine), K)):
                batches[j].downloadWithStagingBuffer(out0)
                exact = exact and bool(np.array_equal(refs[j], out0))
            res[f"{tag}_bit_exact_vs_std_sort_seeds"] = exact
            res[f"{tag}_batches_compared_with_std_sort"] = min(len(pristine), K)
        return res, bad

    out0 = np.empty(n, dtype=np.uint32)
    base = None
    refs = None
    if not args.no_cpu_baseline:
        refs, base = cpu_baseline(host_keys)
    check, bad = verify_batches(args.path, refs)

    # ---- the other path over the same batches, reported beside the headline (never as `value`)
    other_name = "contract" if one_call else "one_call"
    other_fn, _other_names = paths[other_name]
    rearm()
    run_steps(other_fn, 1, 0)
    rearm()
    hybrid_before_other = hybrid_sorts()
    other_elapsed, _ = run_steps(other_fn, K, 0)
    other_hybrid = (not one_call) and hybrid_sorts() - hybrid_before_other == K
    rearm()
    _, other_breakdown = run_steps(other_fn, K, (1 << capi.VRS_KERNEL_COUNT) - 1)
    other_check, other_bad = verify_batches(other_name, refs)
    check.update(other_check)
    if not all(v for v in check.values() if isinstance(v, bool)):
        raise SystemExit(f"VERIFICATION FAILED: {check} (batches {bad} / {other_bad})")

    # ---- per-step spread (SURVEY 8d asks for min and median): K more steps, each bracketed by a queue-idle wait --
    # outside the timed region, which runs its K steps back to back
    rearm()
    singles = []
    for i in range(K):
        gpu.waitIdle()
        t0 = time.perf_counter()
        primary(batches[i])
        gpu.waitIdle()
        singles.append((time.perf_counter() - t0) * 1e3)
    # ---- the blocking form of the calls (VRS_TUNE_ASYNC_SORT = 0: every call waits for its plan's head), same batches back to back
    blocking = None
    if one_call:
        rearm()
        gpu.setTuning(capi.VRS_TUNE_ASYNC_SORT, 0)
        gpu.waitIdle()
        t0 = time.perf_counter()
        for i in range(K):
            primary(batches[i])
        gpu.waitIdle()
        blocking = (time.perf_counter() - t0) / K * 1e3
        gpu.setTuning(capi.VRS_TUNE_ASYNC_SORT, 1)
    # ---- the same batches with every sort sampling for itself (VRS_TUNE_MSD_POOL_REUSE_LAYOUT = 0): the timed region's sorts after the first
    # start in the regions the context kept from the sort before -- a steady workload's gain, reported beside what a sort costs without it
    every_samples = None
    changing = None
    if one_call:
        rearm()
        gpu.setTuning(capi.VRS_TUNE_MSD_POOL_REUSE_LAYOUT, 0)
        gpu.waitIdle()
        t0 = time.perf_counter()
        for i in range(K):
            primary(batches[i])
        gpu.waitIdle()
        every_samples = (time.perf_counter() - t0) / K * 1e3
        gpu.setTuning(capi.VRS_TUNE_MSD_POOL_REUSE_LAYOUT, 1)
        # ---- a workload that CHANGES from sort to sort at equal n: uniform keys and the reference's own 28-bit keys (MultiRadixSort.cpp:110-118:
        # uniform_int_distribution over [0, 0x0FFFFFFF]) alternating -- every kept layout is stale for the next sort; after two the context stops
        # trying for 16 sorts (the back-off ADVICE r5 asked for; before it every such sort ran its two passes twice)
        narrow = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), host_keys[0] >> np.uint32(4))
        lay0 = (ctypes.c_uint64(), ctypes.c_uint64())
        gpu.check(gpu.lib.vrs_one_call_pool_layouts(gpu.handle, ctypes.byref(lay0[0]), ctypes.byref(lay0[1])))
        for i in range(K):
            batches[i].copyFrom(narrow if i % 2 else pristine[i % len(pristine)])
        gpu.waitIdle()
        t0 = time.perf_counter()
        for i in range(K):
            primary(batches[i])
        gpu.waitIdle()
        changing_ms = (time.perf_counter() - t0) / K * 1e3
        lay1 = (ctypes.c_uint64(), ctypes.c_uint64())
        gpu.check(gpu.lib.vrs_one_call_pool_layouts(gpu.handle, ctypes.byref(lay1[0]), ctypes.byref(lay1[1])))
        ok_changing = all(batches[i].verifyKeys(n)[0] == 0 for i in range(K))
        changing = {"ms_per_step": round(changing_ms, 4), "kept_layouts_tried": lay1[0].value - lay0[0].value, "of_them_stale": lay1[1].value - lay0[1].value,
                    "every_batch_ascending": ok_changing,
                    "note": "K further steps, uniform keys and 28-bit keys (the reference's own distribution) alternating: a kept layout never fits the next sort; "
                            "after two stale ones the context samples for the next 16 sorts (outside the timed region)"}
        narrow.release()
    # ---- what a plain device-to-device copy of one batch achieves here (read + write bytes), beside the 8 TB/s figure
    copy_times = []
    for i in range(6):
        gpu.waitIdle()
        t0 = time.perf_counter()
        buf1.copyFrom(pristine[0])
        gpu.waitIdle()
        copy_times.append(time.perf_counter() - t0)
    copy_gbps = 2 * 4 * n / min(copy_times[1:]) / 1e9

    if one_call and (hybrid_steps not in (0, K) or pool_steps not in (0, K) or pool_refused):
        raise SystemExit(f"the timed steps mixed the forms of the one-call sort ({hybrid_steps} of {K} hybrid, {pool_steps} pool, {pool_refused} refused)")
    hybrid = one_call and hybrid_steps == K
    pool = hybrid and pool_steps == K
    other_pool = (not one_call) and other_hybrid and pool_sorts()[0] - pool_after[0] >= K
    bytes_per_key_sort = {"one_call": BYTES_PER_KEY_SORT_POOL if (pool or other_pool) else BYTES_PER_KEY_SORT_HYBRID if (hybrid or other_hybrid)
                          else BYTES_PER_KEY_SORT_ONE_READ,
                          "contract": BYTES_PER_KEY_SORT}
    dom_us = kernels.get(dominant_name, {}).get("avg_us")
    dom_bytes = KERNEL_BYTES_PER_KEY.get(dominant_name, BYTES_PER_KEY_SCATTER) * n
    traffic, traffic_detail = load_traffic_profile(dominant_name, dom_bytes) if n == 10 ** 8 else (None, {"note": "committed for N = 10^8 only"})
    achieved = (dom_bytes / (dom_us * 1e-6) / 1e9) if dom_us else None
    value = n * K / elapsed / 1e9
    sort_bytes = bytes_per_key_sort[args.path] * n
    other_dom_name = pick_dominant(other_breakdown)
    other_dom_us = other_breakdown.get(other_dom_name, {}).get("avg_us")
    shares = {x: round(r["launches"] * r["avg_us"], 1) for x, r in kernels.items() if x in KERNEL_BYTES_PER_KEY}
    result = {
        "metric": "Gkeys/s sorting 10^8 uint32 at 1/2/4/8 MI355X; % of HBM roofline",
        "value": round(value, 3), "unit": "Gkeys/s", "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "ms_per_step_individually_timed": {"min": round(min(singles), 4), "median": round(float(np.median(singles)), 4),
                                           "max": round(max(singles), 4),
                                           "note": "K further steps, each bracketed by a queue-idle wait (outside the timed region)"},
        "calls": {"timed_region": "enqueue-only (the default on a context with its own stream: vrs_sort_keys_u32 returns at once, the next call "
                                  "or vrs_queue_wait_idle settles what the plan still asks for)" if one_call else "stage calls (asynchronous)",
                  "blocking_form_ms_per_step": round(blocking, 4) if blocking else None,
                  "blocking_form_note": "VRS_TUNE_ASYNC_SORT = 0: every call waits for its plan's head; same batches back to back, outside the timed region"},
        "every_sort_samples_ms_per_step": round(every_samples, 4) if every_samples else None,
        "every_sort_samples_note": "VRS_TUNE_MSD_POOL_REUSE_LAYOUT = 0: no sort starts in the regions an earlier one laid out (the timed region's sorts after the "
                                   "first do: a steady workload's gain); same batches back to back, outside the timed region",
        "changing_inputs": changing,
        "config": {"workload": f"BASELINE.json configs[{ {10 ** 7: 1, 10 ** 8: 2}.get(n, 2) }]: {n} uniform random uint32 keys (std::mt19937 seeds 1,2,3), "
                               f"multi_radixsort, 1xMI355X, keys resident in HBM",
                   "path": pool_path_text(gpu.lib, n, tuned.get(capi.VRS_TUNE_MSD_POOL_TOP_BITS, 0)) if pool else
                           (("vrs_sort_keys_u32, hybrid form: one counting read of the keys, an MSD partition by the top 14 bits in "
                             "two stable scatter passes with decoupled look-back (8 + 6 bits), then every bucket sorted by its "
                             "low 18 bits inside one workgroup's LDS (28 B/key); " + str(recount_steps) + " of the timed sorts "
                             "needed a second counting read") if hybrid else
                            ("vrs_sort_keys_u32: the library runs the four 8-bit passes itself -- one counting read of the "
                             "keys, then four stable scatter passes with decoupled look-back (36 B/key)")) if one_call else
                           ("MultiRadixSortPass stages, as MultiRadixSort::execute drives them: 4 x [histograms, prefix, "
                            "scatter] with the caller-visible [W][256] table (48 B/key)"),
                   "num_elements": n, "num_blocks_per_workgroup": B, "num_workgroups": Wg,
                   "passes": "2 MSD scatter passes + 1 LDS-local sort pass" if hybrid else 4,
                   "rank_mode": {1: "ballot", 2: "lds_atomic"}[gpu.lib.vrs_rank_mode(gpu.handle)], "device": dev_name,
                   "compute_units": cus},
        "roofline": {"bound": "hbm", "kernel": dominant_name, "kernel_is": KERNEL_WHAT.get(dominant_name),
                     "chosen_by": "largest launches x average launch time among the byte-moving kernels whose launches carried events in the timed region",
                     "instrumented_us_by_kernel": shares,
                     "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                     "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_us": dom_us,
                     "traffic": traffic, "traffic_detail": traffic_detail,
                     "traffic_source": f"profiles/{dominant_name}_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                       "this command at N = 10^8 in an earlier run (counters cannot be read from inside this run); its "
                                       "commit and box are in traffic_detail",
                     "measured_d2d_copy_GBps": round(copy_gbps, 1),
                     "measured_d2d_copy_note": "hipMemcpyDtoD of one batch in this run, read + write bytes: what this box's "
                                               "HBM delivers for a mixed read/write stream, beside the 8 TB/s spec peak"},
        "sort_roofline": {"algorithmic_bytes": sort_bytes, "bytes_per_key": bytes_per_key_sort[args.path],
                          "achieved_GBps": round(sort_bytes * K / elapsed / 1e9, 1),
                          "frac_of_peak": round(sort_bytes * K / elapsed / 1e9 / HBM_PEAK_GBS, 4)},
        "kernels_timed_region": kernels,
        "kernels_timed_region_note": f"HIP events on the byte-moving kernels' launches of every {args.event_every}th step of the timed region",
        "kernels_all_instrumented_rerun": breakdown,
        "roofline_by_kernel": {name: {"algorithmic_bytes_per_launch": bpk * n, "avg_launch_us": breakdown[name]["avg_us"],
                                      "frac": round(bpk * n / (breakdown[name]["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
                               for name, bpk in KERNEL_BYTES_PER_KEY.items()
                               if name in breakdown and breakdown[name]["avg_us"]},
        "roofline_by_kernel_note": "every byte-moving kernel of the timed path, from the all-kernels-instrumented rerun (launches carry events: "
                                   "a few per cent slower than in the timed region); all three run at or near the rate a device copy reaches on the box (measured_d2d_copy_GBps); the local sort carries 25-33 us of LDS work on top",
        "ms_per_step_uninstrumented_rerun": round(unprofiled / K * 1e3, 4),
        f"{other_name}_path": {
            "value": round(n * K / other_elapsed / 1e9, 3), "unit": "Gkeys/s", "ms_per_step": round(other_elapsed / K * 1e3, 4),
            "bytes_per_key": bytes_per_key_sort[other_name],
            "frac_of_peak": round(bytes_per_key_sort[other_name] * n * K / other_elapsed / 1e9 / HBM_PEAK_GBS, 4),
            "dominant_kernel": {"name": other_dom_name, "avg_us": other_dom_us,
                                "frac": round(KERNEL_BYTES_PER_KEY[other_dom_name] * n / (other_dom_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                                if other_dom_us else None},
            "kernels": other_breakdown, "note": "same batches, uninstrumented timing; not the headline"},
        "verified": dict(check, timed_region_every_batch_ascending_and_permutation_of_its_input=True,
                         timed_region_batches_checked_on_device=K),
    }
    if base:
        result["cpu_baseline"] = base
    for b in batches + pristine + [buf1, hist]:
        b.release()
    p.release()
    gpu.shutdown()
    return result


def bench_pairs(args):
    """--pairs: BASELINE.json configs[3] -- n uint32 keys, each with a uint32 payload (its input position), sorted by
    vrs_sort_pairs_u32 (stable: == std::stable_sort by key; the reference has no key-value path).  Same
    contract as the keys line: K pre-staged batches, the dominant kernel's launches carry HIP events in the timed region."""
    import vkradixsort_amd as vrs
    from vkradixsort_amd import capi

    n, K, W = args.n, args.steps, args.warmup
    S = vrs.Buffer.BufferSettings
    seeds = [1, 2, 3]
    host_keys = [mt19937_keys(s, n) for s in seeds[:max(1, min(3, K, getattr(args, "exact_seeds", 3)))]]
    iota = np.arange(n, dtype=np.uint32)
    gpu = vrs.GPUContext(int(os.environ.get("LOCAL_RANK", "0")))
    gpu.init()
    for kv in (getattr(args, "tune", None) or []):
        k_, v_ = kv.split("=")
        gpu.setTuning(int(k_), int(v_))
    dev_name, cus, mem = gpu.deviceInfo()
    nbuf = max(K, W, 1)
    if (2 * nbuf + len(host_keys) + 3) * 4 * n > 0.8 * mem:
        raise SystemExit("too many pre-staged batches for this device; lower --steps")
    pristine = [vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), k) for k in host_keys]
    pristine_vals = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), iota)
    keys = [vrs.Buffer(gpu, S(4 * n)) for _ in range(nbuf)]
    vals = [vrs.Buffer(gpu, S(4 * n)) for _ in range(nbuf)]
    ktmp, vtmp = vrs.Buffer(gpu, S(4 * n)), vrs.Buffer(gpu, S(4 * n))

    def rearm():
        for i in range(nbuf):
            keys[i].copyFrom(pristine[i % len(pristine)])
            vals[i].copyFrom(pristine_vals)
        gpu.waitIdle()

    def sort(i):
        gpu.check(gpu.lib.vrs_sort_pairs_u32(gpu.handle, keys[i].handle, ktmp.handle, vals[i].handle, vtmp.handle, n))

    def hybrid_sorts():
        h = ctypes.c_uint64()
        gpu.check(gpu.lib.vrs_one_call_hybrid_sorts(gpu.handle, ctypes.byref(h)))
        return h.value

    def pool_sorts():
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        gpu.check(gpu.lib.vrs_one_call_pool_sorts(gpu.handle, ctypes.byref(a), ctypes.byref(b)))
        return a.value

    def kernel_table():
        t = {}
        for kid, name in capi.KERNEL_NAMES.items():
            cnt, ms = gpu.profileQuery(kid)
            if cnt:
                t[name] = {"launches": cnt, "avg_us": round(ms / cnt * 1e3, 2)}
        return t

    def run_steps(count, mask, every=1):
        gpu.profileReset()
        gpu.profileEnableMask(mask)
        gpu.waitIdle()
        t0 = time.perf_counter()
        for i in range(count):
            if every > 1:
                gpu.profileEnableMask(mask if i % every == 0 else 0)
            sort(i)
        gpu.waitIdle()
        dt = time.perf_counter() - t0
        gpu.profileEnable(False)
        return dt, kernel_table()

    rearm()
    run_steps(W, 0)
    rearm()
    h0, p0 = hybrid_sorts(), pool_sorts()
    # every byte-moving kernel of either form carries events in the timed region; the dominant one is named from them (pick_dominant)
    moving = sum(1 << k for k in (capi.VRS_KERNEL_LOOKBACK_SCATTER, capi.VRS_KERNEL_LOCAL_SORT, capi.VRS_KERNEL_POOL_PASS_A, capi.VRS_KERNEL_POOL_PASS_B,
                                   capi.VRS_KERNEL_DIGIT_TABLES))
    elapsed, kernels = run_steps(K, moving, every=args.event_every)
    hybrid_steps, pool_steps = hybrid_sorts() - h0, pool_sorts() - p0
    # every output of the timed region: keys ascending and a permutation of the input's, payloads a permutation of 0 .. n-1
    key_prints = [p_.verifyKeys(n)[1:] for p_ in pristine]
    val_print = pristine_vals.verifyKeys(n)[1:]
    bad = [i for i in range(K) if (lambda r: r[0] != 0 or r[1:] != key_prints[i % len(pristine)])(keys[i].verifyKeys(n))
           or vals[i].verifyKeys(n)[1:] != val_print]
    if bad:
        raise SystemExit(f"VERIFICATION FAILED: timed-region outputs of steps {bad}")
    # bit for bit against std::stable_sort (keys AND payloads) for the first batch of every seed; its time is the CPU baseline
    base = None
    exact = None
    if not args.no_cpu_baseline:
        from tests import _oracle
        orc = _oracle.load()
        ok, ov = np.empty(n, np.uint32), np.empty(n, np.uint32)
        times, exact = [], True
        for j in range(min(len(pristine), K)):
            rk, rv, ms = orc.stable_sort_pairs(host_keys[j], iota)
            times.append(ms)
            keys[j].downloadWithStagingBuffer(ok)
            vals[j].downloadWithStagingBuffer(ov)
            exact = exact and bool(np.array_equal(rk, ok)) and bool(np.array_equal(rv, ov))
        if not exact:
            raise SystemExit("VERIFICATION FAILED: pairs differ from std::stable_sort")
        cores, model = orc.cpu_info()
        base = {"value": round(n / (min(times) * 1e-3) / 1e9, 5), "unit": "Gkeys/s", "cores": 1, "kind": "port",
                "sample": f"std::stable_sort by key of the full {n}-pair batches of seeds 1..{len(times)} (key << 32 | payload), one "
                          "repetition each: " + ", ".join(f"{t:.0f}" for t in times) + " ms (value = the fastest)",
                "host": f"1 thread of {cores} hardware threads ({model})"}
    rearm()
    unprofiled, _ = run_steps(K, 0)
    rearm()
    _, breakdown = run_steps(K, (1 << capi.VRS_KERNEL_COUNT) - 1)
    if hybrid_steps not in (0, K):
        raise SystemExit(f"the timed steps mixed the two forms of the one-call sort ({hybrid_steps} of {K} hybrid)")
    if pool_steps not in (0, K):
        raise SystemExit(f"the timed steps mixed the pool and the counted form ({pool_steps} of {K} pool)")
    hybrid, pool = hybrid_steps == K, pool_steps == K
    # counting read 4 + (2 MSD passes + local sort | 4 LSD passes) x (read 8 + write 8); the pool form has no counting read
    bpp = 48 if pool else 52 if hybrid else 68
    pair_bytes = {"digit_tables": 4, "lookback_scatter": 16, "local_sort": 16, "pool_pass_a": 16, "pool_pass_b": 16}
    dom_name = pick_dominant({k_: v for k_, v in kernels.items() if k_ in pair_bytes})
    dom_us = kernels.get(dom_name, {}).get("avg_us")
    dom_bytes = pair_bytes.get(dom_name, 16) * n
    achieved = dom_bytes / (dom_us * 1e-6) / 1e9 if dom_us else None
    traffic_name = "lookback_scatter_pairs" if dom_name == "lookback_scatter" else f"{dom_name}_pairs"
    result = {
        "metric": "Gkeys/s sorting 10^8 uint32 at 1/2/4/8 MI355X; % of HBM roofline",
        "value": round(n * K / elapsed / 1e9, 3), "unit": "Gkeys/s", "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[3]: {n} uint32 key + uint32 payload pairs (keys: std::mt19937 seeds 1,2,3; payload = "
                               "input position), 1xMI355X, resident in HBM; every key counts once (a pair per key)",
                   "path": pool_path_text(gpu.lib, n, 0, pairs=True) if pool else
                           ("vrs_sort_pairs_u32, hybrid form: one counting read of the keys, two stable MSD scatter passes with decoupled "
                            "look-back (keys + payloads), LDS-local bucket sort -- 52 B/pair") if hybrid else
                           "vrs_sort_pairs_u32: one counting read + four stable look-back scatter passes -- 68 B/pair",
                   "num_elements": n, "device": dev_name, "compute_units": cus},
        "roofline": {"bound": "hbm", "kernel": dom_name, "kernel_is": (KERNEL_WHAT.get(dom_name) or "") + " -- here with payloads: every key and payload read and written once per launch",
                     "chosen_by": "largest launches x average launch time among the byte-moving kernels whose launches carried events in the timed region",
                     "instrumented_us_by_kernel": {x: round(r["launches"] * r["avg_us"], 1) for x, r in kernels.items() if x in pair_bytes},
                     "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None, "algorithmic_bytes_per_launch": dom_bytes,
                     "avg_launch_us": dom_us, "traffic": load_traffic_profile(traffic_name, float(dom_bytes))[0] if n == 10 ** 8 else None,
                     "traffic_source": f"profiles/{traffic_name}_traffic.json (rocprofv3 --pmc passes of an earlier run of this command)"},
        "sort_roofline": {"algorithmic_bytes": bpp * n, "bytes_per_key": bpp, "achieved_GBps": round(bpp * n * K / elapsed / 1e9, 1),
                          "frac_of_peak": round(bpp * n * K / elapsed / 1e9 / HBM_PEAK_GBS, 4)},
        "kernels_timed_region": kernels, "kernels_all_instrumented_rerun": breakdown,
        "ms_per_step_uninstrumented_rerun": round(unprofiled / K * 1e3, 4),
        "verified": {"timed_region_every_batch_keys_ascending_and_permutations": True, "timed_region_batches_checked_on_device": K,
                     "bit_exact_vs_std_stable_sort_seeds": exact},
    }
    if base:
        result["cpu_baseline"] = base
    for b in keys + vals + pristine + [pristine_vals, ktmp, vtmp]:
        b.release()
    gpu.shutdown()
    return result


class _StdoutToStderr:
    """RCCL prints a version banner on STDOUT when its first communicator comes up; the contract is ONE JSON line
    there.  Route fd 1 to fd 2 while the process group initialises."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def bench_multi_python(args, fallback_reason=None):
    """N > 1 through the Python orchestration (vkradixsort_amd/distributed.py over torch.distributed): --dist-path python, and the
    fallback of the C path should its set-up fail on any rank (fallback_reason: the process group is up already)."""
    import torch
    import torch.distributed as dist

    from vkradixsort_amd.distributed import HipLocalSortBackend, RangeShardedSort

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if fallback_reason is None:
        with _StdoutToStderr():
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            warm = torch.zeros(8, dtype=torch.int64, device=dev)
            dist.all_reduce(warm)  # brings the communicator up (and its banner out) now
            torch.cuda.synchronize()
    # run the sort on a non-blocking stream of its own: work on the legacy null stream would implicitly serialise
    # with every blocking stream, and the exchange of round r+1 (RCCL's stream) is meant to overlap the sort of round r
    work_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(work_stream)
    n, B, K, W = args.n, args.blocks, args.steps, args.warmup
    shard = mt19937_keys(1000 + rank, n)  # shard g uses seed 1000+g
    pristine = torch.from_numpy(shard.view(np.int32)).to(dev)
    nbuf = max(K, W, 1)
    batches = [torch.empty_like(pristine) for _ in range(nbuf)]
    cap = int(n * 1.25) + 4096
    backend = HipLocalSortBackend(local, capacity=cap, blocks_per_workgroup=B)
    sorter = RangeShardedSort(backend, recv_capacity=cap, make_empty=lambda m: torch.empty(m, dtype=torch.int32, device=dev),
                              rounds=args.rounds if world > 1 or args.rounds_forced else 1)

    def rearm():
        for b in batches:
            b.copy_(pristine)
        torch.cuda.synchronize()

    rearm()
    res = None
    # warm-up steps double as a measurement of the exchange pipelining depth: how many rounds pay off depends on the
    # xGMI all-to-all rate relative to the local sort, which only shows on the real node.  Every rank takes the same
    # decision (MAX over ranks of each candidate's time).
    candidates = [r for r in (args.rounds, 2, 1) if r <= sorter.rounds] if (world > 1 and not args.rounds_forced) else []
    candidates = list(dict.fromkeys(candidates))[:W]
    tried = {}
    if candidates:  # set-up, not a step: first use allocates scratch tables and wraps the buffers
        sorter.step(batches[0], n, n_total_hint=n * world)
        torch.cuda.synchronize()
        batches[0].copy_(pristine)
    for i in range(W):
        if i < len(candidates):
            sorter.rounds = candidates[i]
            dist.barrier()
            torch.cuda.synchronize()
            tw = time.perf_counter()
        res = sorter.step(batches[i], n, n_total_hint=n * world)
        if i < len(candidates):
            torch.cuda.synchronize()
            tt = torch.tensor([time.perf_counter() - tw], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tried[candidates[i]] = float(tt.item())
    if tried:
        sorter.rounds = min(tried, key=tried.get)
    torch.cuda.synchronize()
    rearm()
    from vkradixsort_amd import capi
    backend.ctx.profileReset()
    backend.ctx.profileEnableMask(1 << capi.VRS_KERNEL_LOOKBACK_SCATTER)  # events ride on the dominant kernel's own launches
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        res = sorter.step(batches[i], n, n_total_hint=n * world)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    backend.ctx.profileEnable(False)
    lb_launches, lb_ms = backend.ctx.profileQuery(capi.VRS_KERNEL_LOOKBACK_SCATTER)
    elapsed = fabric.reduce(elapsed, "max")

    # verification outside the timed region: every range ascending, ranges ordered across ranks, nothing lost
    out = res.keys[:res.count]
    flipped = out ^ torch.tensor(-2 ** 31, dtype=torch.int32, device=dev)  # unsigned order on int32 storage
    ok_sorted = bool((flipped[1:] >= flipped[:-1]).all().item()) if res.count > 1 else True
    lo = int(flipped[0].item()) if res.count else 2 ** 31 - 1
    hi = int(flipped[-1].item()) if res.count else -2 ** 31
    edges = torch.tensor([lo, hi, res.count, int(out.to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item())],
                         dtype=torch.int64, device=dev)
    gathered = [torch.empty_like(edges) for _ in range(world)]
    dist.all_gather(gathered, edges)
    src_sum = torch.tensor([int(shard.astype(np.uint64).sum())], dtype=torch.int64, device=dev)
    dist.all_reduce(src_sum)
    oks = torch.tensor([1 if ok_sorted else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(oks, op=dist.ReduceOp.MIN)
    g = [x.cpu().tolist() for x in gathered]
    nonempty = [x for x in g if x[2] > 0]
    ordered = all(nonempty[i][1] <= nonempty[i + 1][0] for i in range(len(nonempty) - 1))
    total = sum(x[2] for x in g)
    check = {"ranges_sorted": bool(oks.item()), "ranges_ordered_across_ranks": ordered, "count_ok": total == n * world,
             "checksum_ok": sum(x[3] for x in g) == int(src_sum.item())}
    result = None
    if rank == 0:
        if not all(check.values()):
            raise SystemExit(f"VERIFICATION FAILED: {check}")
        value = n * world * K / elapsed / 1e9
        # dominant kernel: the look-back scatter of the local sorts (one launch per pass per received sub-range); a launch
        # over m keys moves 8 m algorithmic bytes; rank 0's launches of the timed region
        recv_keys = int(g[0][2])
        # every launch of the look-back scatter reads and writes one received sub-range once (the LSD form launches it four
        # times per sub-range, the hybrid form twice)
        lb_bytes = 8 * (recv_keys / max(sorter.rounds, 1)) * lb_launches
        lb_achieved = lb_bytes / (lb_ms * 1e-3) / 1e9 if lb_ms > 0 else None
        # the form the per-range sorts took, read off the launch count: two look-back passes per range = hybrid (28 B/key)
        passes_per_range = lb_launches / max(K * max(sorter.rounds, 1), 1)
        sort_bpk = 28 if passes_per_range <= 2.5 else 36
        step_bpk = 12 + sort_bpk
        base = None
        if not args.no_cpu_baseline:
            from tests import _oracle
            orc = _oracle.load()
            sample = shard[:min(n, 2 * 10 ** 7)]
            _, ms = orc.std_sort(sample)
            cores, model = orc.cpu_info()
            base = {"value": round(sample.size / (ms * 1e-3) / 1e9, 5), "unit": "Gkeys/s", "cores": 1, "kind": "port",
                    "sample": f"std::sort of the first {sample.size} keys of rank 0's shard, 1 repetition, {ms:.0f} ms",
                    "host": f"1 thread of {cores} hardware threads ({model})"}
        result = {
            "metric": "Gkeys/s sorting 10^8 uint32 at 1/2/4/8 MI355X; % of HBM roofline",
            "value": round(value, 3), "unit": "Gkeys/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{world} x {n} uniform random uint32 keys (std::mt19937 seed 1000+rank), sharded by key "
                                   f"range: top-byte partition pass, RCCL all-to-all over xGMI, one ranged one-call sort per received sub-range",
                       "fallback_from_the_c_path_because": fallback_reason,
                       "num_elements_per_gpu": n, "num_blocks_per_workgroup": B, "parallelism": f"range-sharded x{world}",
                       "exchange_rounds": sorter.rounds, "rounds_tried_in_warmup_ms": {str(k): round(v * 1e3, 3) for k, v in tried.items()},
                       "hbm_bytes_per_key": step_bpk,
                       "hbm_bytes_per_key_breakdown": {"partition_pass_histogram_read": 4, "partition_pass_scatter": 8,
                                                       "local_sorts": sort_bpk,
                                                       "local_sort_form": ("hybrid: counting read 4 + two look-back scatters 16 + "
                                                                           "LDS-local bucket sort 8" if sort_bpk == 28 else
                                                                           "LSD: counting read 4 + four look-back scatters 32"),
                                                       "lookback_passes_per_range_sort": round(passes_per_range, 2)}},
            "roofline": {"bound": "hbm", "kernel": "lookback_scatter of the local sorts (rank 0's launches in the timed region)",
                         "achieved": round(lb_achieved, 1) if lb_achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(lb_achieved / HBM_PEAK_GBS, 4) if lb_achieved else None,
                         "launches": lb_launches, "avg_launch_us": round(lb_ms / lb_launches * 1e3, 2) if lb_launches else None,
                         "algorithmic_bytes_per_launch": round(lb_bytes / lb_launches) if lb_launches else None,
                         "traffic": None},
            "step_roofline": {"bytes_per_key": step_bpk, "achieved_GBps_per_gpu": round(step_bpk * n * K / elapsed / 1e9, 1),
                              "frac_of_peak": round(step_bpk * n * K / elapsed / 1e9 / HBM_PEAK_GBS, 4),
                              "note": f"per GPU, whole step incl. the xGMI exchange: 12 B/key partition pass + {sort_bpk} B/key "
                                      "ranged sorts of the received sub-ranges"},
            "shard_sizes": [x[2] for x in g],
            "verified": check,
        }
        if base:
            result["cpu_baseline"] = base
    backend.close()
    dist.destroy_process_group()
    return result


def _rccl_communicator(torch, dist, rank, world, dev):
    """A raw ncclComm_t of the RCCL copy PyTorch already loaded (vrs_dist_create binds that same copy at run time): rank 0
    draws the unique id, torch.distributed carries it to the others."""
    rccl_path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    rccl = ctypes.CDLL(rccl_path if os.path.exists(rccl_path) else "librccl.so", mode=ctypes.RTLD_GLOBAL)

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    uid = UniqueId()
    if rank == 0 and rccl.ncclGetUniqueId(ctypes.byref(uid)) != 0:
        raise RuntimeError("ncclGetUniqueId failed")
    t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).to(dev)
    dist.broadcast(t, 0)
    ctypes.memmove(ctypes.byref(uid), bytes(t.cpu().numpy().tobytes()), 128)
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    if rccl.ncclCommInitRank(ctypes.byref(comm), world, uid, rank) != 0:
        raise RuntimeError("ncclCommInitRank failed")
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    return rccl, comm


class _RcclFabric:
    """What bench_multi needs of the ranks' fabric, over torch.distributed + a raw RCCL communicator (one process per GPU, the way
    the driver launches the bench)."""
    transport = "RCCL (send/recv over xGMI; one process per GPU)"
    wire = "RCCL"

    def __init__(self):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", self.rank))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        torch.cuda.set_device(self.local)
        self.dev = torch.device("cuda", self.local)
        with _StdoutToStderr():
            dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)
            warm = torch.zeros(8, dtype=torch.int64, device=self.dev)
            dist.all_reduce(warm)  # brings the communicator up (and its banner out) now
            torch.cuda.synchronize()
            self.rccl, self.comm = _rccl_communicator(torch, dist, self.rank, self.world, self.dev)
        self.devices = self.world
        self.can_fall_back = True

    def create(self, lib, gpu, cap, rounds, out):
        return lib.vrs_dist_create(gpu.handle, self.comm, self.rank, self.world, cap, rounds, ctypes.byref(out))

    def barrier(self):
        self.dist.barrier()
        self.torch.cuda.synchronize()

    def reduce(self, value, op):
        t = self.torch.tensor([value], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.MIN)
        return float(t.item())

    def all_gather(self, ints):
        t = self.torch.tensor(list(ints), dtype=self.torch.int64, device=self.dev)
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [x.cpu().tolist() for x in out]

    def close(self):
        self.rccl.ncclCommDestroy(self.comm)
        self.dist.destroy_process_group()


class _LoopbackHub:
    """--loopback-ranks W: the ranks are THREADS of this process, the wire the library's in-process transport
    (vrs_dist_loopback_*: device copies ordered by events).  With one GPU all ranks share it -- no scaling number, but every rank-to-rank
    code path of the step and of this bench runs; with W GPUs in one process this is a multi-GPU sort without RCCL."""

    def __init__(self, lib, world, devices):
        import threading
        self.lib, self.world, self.devices = lib, world, devices
        self.handle = ctypes.c_void_p()
        if lib.vrs_dist_loopback_create(world, ctypes.byref(self.handle)) != 0:
            raise RuntimeError("vrs_dist_loopback_create failed")
        self.rendezvous = threading.Barrier(world)
        self.slots = [None] * world

    def destroy(self):
        self.lib.vrs_dist_loopback_destroy(self.handle)


class _LoopbackFabric:
    can_fall_back = False
    wire = "loopback"

    def __init__(self, hub, rank):
        self.hub, self.rank, self.world = hub, rank, hub.world
        self.local = rank % hub.devices
        self.devices = hub.devices
        self.transport = (f"loopback: {hub.world} ranks as host threads of one process on {hub.devices} device(s), device-to-device "
                          "copies ordered by events (no RCCL, no xGMI: not a scaling number)")
        self._wire = None

    def create(self, lib, gpu, cap, rounds, out):
        from vkradixsort_amd import capi
        if self._wire is None:
            self._wire = capi.DistTransport()
            if lib.vrs_dist_loopback_transport(self.hub.handle, self.rank, ctypes.byref(self._wire)) != 0:
                return 1
        return lib.vrs_dist_create_with_transport(gpu.handle, ctypes.byref(self._wire), self.rank, self.world, cap, rounds, ctypes.byref(out))

    def barrier(self):
        self.hub.rendezvous.wait()

    def _exchange(self, value):
        self.hub.slots[self.rank] = value
        self.hub.rendezvous.wait()
        got = list(self.hub.slots)
        self.hub.rendezvous.wait()  # nobody overwrites its slot before everybody has read
        return got

    def reduce(self, value, op):
        got = self._exchange(float(value))
        return max(got) if op == "max" else min(got)

    def all_gather(self, ints):
        return [list(x) for x in self._exchange(list(ints))]

    def close(self):
        pass


def bench_multi_loopback(args):
    """bench.py --loopback-ranks W [--n ...]: bench_multi's step, verification and JSON assembly with W ranks as threads."""
    import threading

    from vkradixsort_amd import capi
    lib = capi.load_library()
    cnt = ctypes.c_int()
    if lib.vrs_device_count(ctypes.byref(cnt)) != 0 or cnt.value == 0:
        raise SystemExit("no device")
    world = args.loopback_ranks
    hub = _LoopbackHub(lib, world, min(cnt.value, world))
    results, errors = [None] * world, []

    def run(r):
        try:
            results[r] = bench_multi(args, _LoopbackFabric(hub, r))
        except BaseException as e:  # noqa: BLE001 -- the other ranks must not wait for this one
            errors.append((r, repr(e)))
            hub.rendezvous.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    hub.destroy()
    if errors:
        raise SystemExit(f"loopback ranks failed: {errors}")
    return results[0]


def bench_multi(args, fabric=None):
    """N > 1 (or VRS_BENCH_FORCE_MULTI=1 at N = 1): the multi-GPU step behind the C ABI (vrs_dist_*, csrc/vrs_dist.hip) over a
    raw RCCL communicator -- the single-GPU hybrid sort with the all-to-all between its two MSD passes.  (fabric: the loopback
    flavour of --loopback-ranks; default RCCL.)"""
    import vkradixsort_amd as vrs
    from vkradixsort_amd import capi

    fabric = fabric or _RcclFabric()
    rank, world, local = fabric.rank, fabric.world, fabric.local
    lib = capi.load_library()
    n, K, W = args.n, args.steps, args.warmup
    S = vrs.Buffer.BufferSettings
    shard = mt19937_keys(1000 + rank, n)  # shard g uses seed 1000+g
    gpu = vrs.GPUContext(local)
    gpu.init()
    pristine = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), shard)
    nbuf = max(K, W, 1)
    batches = [vrs.Buffer(gpu, S(4 * n)) for _ in range(nbuf)]
    cap = int(n * 1.25) + 4096

    def make_dist(rounds):
        d = ctypes.c_void_p()
        rc = fabric.create(lib, gpu, cap, rounds, d)
        if rc != 0:
            raise RuntimeError(f"vrs_dist_create: {lib.vrs_dist_last_error(None).decode()}")
        return d

    def step(d, b):
        out_buf, out_n = ctypes.c_void_p(), ctypes.c_uint32()
        rc = lib.vrs_dist_sort_keys_u32(d, b.handle, n, ctypes.byref(out_buf), ctypes.byref(out_n))
        if rc != 0:
            raise RuntimeError(f"vrs_dist_sort_keys_u32 (rank {rank}): {lib.vrs_dist_last_error(d).decode()}")
        return out_buf, out_n.value

    def rearm():
        for b in batches:
            b.copyFrom(pristine)
        gpu.waitIdle()

    def barrier():
        gpu.waitIdle()
        fabric.barrier()

    rearm()
    # warm-up steps double as a measurement of the exchange pipelining depth: how many rounds pay off depends on the
    # xGMI all-to-all rate relative to the local work, which only shows on the real node.  Every rank takes the same
    # decision (MAX over ranks of each candidate's time).
    max_rounds = max(1, 32 // min(world, 32))
    want = min(args.rounds, max_rounds)
    candidates = [want] if (world == 1 or args.rounds_forced) else list(dict.fromkeys(r for r in (want, 2, 1) if r <= max_rounds))
    # set-up, not a step: first use allocates scratch and loads the code objects.  Should it fail on ANY rank (no RCCL to bind, a
    # transport error), every rank learns so and the run goes through the Python orchestration instead -- a number from the other
    # path is worth more than none; config.path says which path ran.
    handles, tried, failure = {}, {}, None

    def agreed(stage):
        """True if `stage` went well on EVERY rank (a rank that failed skipped its collectives: the others must not enter the
        next stage's and wait for it)."""
        if int(fabric.reduce(0 if failure else 1, "min")) == 1:
            return True
        print(f"[bench] rank {rank}: the C path's {stage} failed ({failure or 'on another rank'})" +
              ("; falling back to --dist-path python" if fabric.can_fall_back else ""), file=sys.stderr)
        return False

    try:  # stage 1, no communication: the endpoints (binds RCCL, allocates the landing areas)
        handles = {r: make_dist(r) for r in candidates}
    except Exception as e:  # noqa: BLE001 -- reported and agreed on below
        failure = repr(e)
    ok = agreed("set-up")
    if ok:
        try:  # stage 2: a first step per candidate (every exit of a step is decided collectively inside the library)
            for r, d in handles.items():
                step(d, batches[0])
                gpu.waitIdle()
                batches[0].copyFrom(pristine)
        except Exception as e:  # noqa: BLE001
            failure = repr(e)
        ok = agreed("first step")
    if not ok:
        for h in handles.values():
            lib.vrs_dist_destroy(h)
        for b in batches + [pristine]:
            b.release()
        gpu.shutdown()
        if not fabric.can_fall_back:
            raise RuntimeError(f"rank {rank}: {failure or 'another rank failed'}")
        return bench_multi_python(args, fallback_reason=failure or "another rank failed")
    if len(candidates) > 1:
        for r, d in handles.items():
            barrier()
            tw = time.perf_counter()
            step(d, batches[0])
            gpu.waitIdle()
            tried[r] = fabric.reduce(time.perf_counter() - tw, "max")
            batches[0].copyFrom(pristine)
        rounds = min(tried, key=tried.get)
    else:
        rounds = candidates[0]
    d = handles[rounds]
    for i in range(W):
        step(d, batches[i % nbuf])
    gpu.waitIdle()
    rearm()
    st0 = [ctypes.c_uint64() for _ in range(3)]
    lib.vrs_dist_stats(d, *[ctypes.byref(x) for x in st0])
    gr0 = ctypes.c_uint64()
    lib.vrs_dist_grouped_rounds(d, ctypes.byref(gr0))
    gpu.profileReset()
    # events ride on the scatter passes' own launches: the look-back / reserving passes and, where a round is finished by the pool
    # form's second half (vrs_msd_finish_grouped_counts_u32), its second pass
    gpu.profileEnableMask((1 << capi.VRS_KERNEL_LOOKBACK_SCATTER) | (1 << capi.VRS_KERNEL_POOL_PASS_B))
    barrier()
    t0 = time.perf_counter()
    out_buf, out_n = None, 0
    for i in range(K):
        out_buf, out_n = step(d, batches[i])
    barrier()
    elapsed = time.perf_counter() - t0
    gpu.profileEnable(False)
    lb_launches, lb_ms = gpu.profileQuery(capi.VRS_KERNEL_LOOKBACK_SCATTER)
    pb_launches, pb_ms = gpu.profileQuery(capi.VRS_KERNEL_POOL_PASS_B)
    pool_finish = pb_launches > 0
    lb_launches, lb_ms = lb_launches + pb_launches, lb_ms + pb_ms
    st1 = [ctypes.c_uint64() for _ in range(3)]
    lib.vrs_dist_stats(d, *[ctypes.byref(x) for x in st1])
    hybrid_rounds, fallback_rounds, byte_steps = (b.value - a.value for a, b in zip(st0, st1))
    gr1 = ctypes.c_uint64()
    lib.vrs_dist_grouped_rounds(d, ctypes.byref(gr1))
    grouped_rounds = gr1.value - gr0.value
    elapsed = fabric.reduce(elapsed, "max")

    # verification outside the timed region: every range ascending (device check), ranges ordered across ranks, nothing lost
    desc, ksum, kmix = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    gpu.check(lib.vrs_verify_keys_u32(gpu.handle, out_buf, out_n, ctypes.byref(desc), ctypes.byref(ksum), ctypes.byref(kmix)))
    ends = np.zeros(2, np.uint32)
    if out_n:
        base_ptr = lib.vrs_buffer_device_ptr(out_buf)
        for j, off in enumerate((0, out_n - 1)):
            v = ctypes.c_void_p()
            gpu.check(lib.vrs_buffer_wrap(gpu.handle, ctypes.c_void_p(base_ptr + 4 * off), 4, ctypes.byref(v)))
            gpu.check(lib.vrs_buffer_download(gpu.handle, v, ends[j:].ctypes.data_as(ctypes.c_void_p), 4))
            lib.vrs_buffer_release(v)
    g = fabric.all_gather([int(ends[0]), int(ends[1]), out_n, ksum.value & (2 ** 62 - 1), 1 if desc.value == 0 else 0,
                           int(shard.astype(np.uint64).sum()) & (2 ** 62 - 1)])
    nonempty = [x for x in g if x[2] > 0]
    check = {"ranges_sorted": all(x[4] == 1 for x in g),
             "ranges_ordered_across_ranks": all(nonempty[i][1] <= nonempty[i + 1][0] for i in range(len(nonempty) - 1)),
             "count_ok": sum(x[2] for x in g) == n * world,
             "checksum_ok": sum(x[3] for x in g) % 2 ** 62 == sum(x[5] for x in g) % 2 ** 62}
    result = None
    if rank == 0:
        if not all(check.values()):
            raise SystemExit(f"VERIFICATION FAILED: {check}")
        value = n * world * K / elapsed / 1e9
        recv_keys = int(g[0][2])
        hybrid = byte_steps == 0
        # rank 0's look-back scatter launches of the timed region: hybrid shape = the first MSD pass over the shard + one second
        # pass per received sub-range (each reads and writes its keys once); byte shape = the launches of the local sorts
        grouped = (not hybrid) and grouped_rounds > 0 and fallback_rounds == 0
        lb_bytes = 8.0 * (n + recv_keys) * K if hybrid else 8.0 * (recv_keys / max(rounds, 1)) * lb_launches
        lb_achieved = lb_bytes / (lb_ms * 1e-3) / 1e9 if lb_ms > 0 else None
        # byte shape: 12 (contract partition pass) + what vrs_sort_keys_u32_ranged moves per received sub-range: 28 in its hybrid
        # form (from 1.3e7 keys on), 36 in its LSD form
        # byte shape with the grouped finish: 12 + (counting read 4 + second MSD pass 8 + local sort 8); with the counts the step
        # has anyway (vrs_msd_finish_grouped_counts_u32: the pool form's second half): 12 + (8 + 8)
        sort_bpk = 28 if hybrid else (28 if pool_finish else 32) if grouped else (40 if recv_keys / max(rounds, 1) >= 1.3e7 else 48)
        # the exchange reads what it sends and writes what it lands: 8 bytes per key -- but for the keys a rank keeps for itself when the
        # byte shape's rounds are finished by the pool form's second half (they stay where the partition pass wrote them: 1 / world)
        own_in_place = grouped and pool_finish and os.environ.get("VRS_DIST_COPY_OWN") != "1"
        xchg_bpk = round(8.0 * (world - 1) / world, 2) if own_in_place else 8
        base = None
        if not args.no_cpu_baseline:
            from tests import _oracle
            orc = _oracle.load()
            sample = shard[:min(n, 2 * 10 ** 7)]
            _, ms = orc.std_sort(sample)
            cores, model = orc.cpu_info()
            base = {"value": round(sample.size / (ms * 1e-3) / 1e9, 5), "unit": "Gkeys/s", "cores": 1, "kind": "port",
                    "sample": f"std::sort of the first {sample.size} keys of rank 0's shard, 1 repetition, {ms:.0f} ms",
                    "host": f"1 thread of {cores} hardware threads ({model})"}
        result = {
            "metric": "Gkeys/s sorting 10^8 uint32 at 1/2/4/8 MI355X; % of HBM roofline",
            "value": round(value, 3), "unit": "Gkeys/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{world} x {n} uniform random uint32 keys (std::mt19937 seed 1000+rank), sharded by key range "
                                   f"(BASELINE.json configs[4] at 8 GPUs), keys resident in HBM",
                       "path": ("vrs_dist_sort_keys_u32, hybrid shape: counting read + first MSD pass of the shard, one all-gather + one "
                                f"all-reduce of counts, {fabric.wire} send/recv of one message per (sender, top byte) in "
                                f"{rounds} round(s), second MSD pass + LDS-local sort per received sub-range") if hybrid else
                               ("vrs_dist_sort_keys_u32, byte shape (the step's default; VRS_DIST_SHAPE=hybrid asks for the other): contract "
                                f"partition pass by the top byte, {fabric.wire} send/recv of one message per (sender, top byte) in {rounds} round(s), "
                                + ("per received sub-range the pool form's second half -- a sample, the second MSD pass by the next 6..8 bits into "
                                   "the buckets' slack regions, the LDS-local sort: nothing is read to be counted (vrs_msd_finish_grouped_counts_u32)"
                                   if pool_finish else
                                   "per received sub-range one counting read + second MSD pass by the next 8 bits + LDS-local sort "
                                   "(vrs_msd_finish_grouped_u32)")) if grouped else
                               ("vrs_dist_sort_keys_u32, byte shape (the global top-14-bit buckets would not fit the local sort): contract "
                                f"partition pass by the top byte, {fabric.wire} send/recv per (sender, round) in {rounds} round(s), "
                                "vrs_sort_keys_u32_ranged per received sub-range (its own 16384 buckets)"),
                       "num_elements_per_gpu": n, "parallelism": f"range-sharded x{world}", "transport": fabric.transport,
                       "devices": fabric.devices, "exchange_rounds": rounds,
                       "rounds_tried_in_warmup_ms": {str(k): round(v * 1e3, 3) for k, v in tried.items()},
                       "received_sub_ranges": {"finished_in_hybrid_shape": int(hybrid_rounds), "finished_grouped_in_byte_shape": int(grouped_rounds),
                                               "sorted_from_scratch_after_a_refused_plan": int(fallback_rounds)},
                       "hbm_bytes_per_key": sort_bpk + xchg_bpk,
                       "own_keys": ("left where the partition pass wrote them: the finish reads every top byte as two pieces "
                                    "(vrs_msd_finish_grouped_split_u32)") if own_in_place else "copied beside the received ones",
                       "hbm_bytes_per_key_breakdown": ({"counting_read": 4, "first_msd_pass": 8, "exchange_read_and_landing_write": 8,
                                                        "second_msd_pass": 8, "local_sort": 8} if hybrid else
                                                       {"partition_pass": 12, "exchange_read_and_landing_write": xchg_bpk,
                                                        "local_sorts": sort_bpk - 12})},
            "roofline": {"bound": "hbm", "kernel": "lookback_scatter (rank 0's launches in the timed region: the first MSD pass over the "
                                                   "shard and the second pass over every received sub-range)",
                         "achieved": round(lb_achieved, 1) if lb_achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(lb_achieved / HBM_PEAK_GBS, 4) if lb_achieved else None,
                         "launches": lb_launches, "avg_launch_us": round(lb_ms / lb_launches * 1e3, 2) if lb_launches else None,
                         "algorithmic_bytes_per_launch": round(lb_bytes / lb_launches) if lb_launches else None,
                         "traffic": None},
            "step_roofline": {"bytes_per_key": sort_bpk + xchg_bpk, "achieved_GBps_per_gpu": round((sort_bpk + xchg_bpk) * n * K / elapsed / 1e9, 1),
                              "frac_of_peak": round((sort_bpk + xchg_bpk) * n * K / elapsed / 1e9 / HBM_PEAK_GBS, 4),
                              "note": "per GPU, whole step incl. the exchange's read of the sent keys and write of the received ones"},
            "shard_sizes": [x[2] for x in g],
            "verified": check,
        }
        if base:
            result["cpu_baseline"] = base
    for h in handles.values():
        lib.vrs_dist_destroy(h)
    for b in batches + [pristine]:
        b.release()
    gpu.shutdown()
    fabric.close()
    return result


def other_configs(args):
    """The default line's `configs` block: BASELINE.json configs[1] (10^7 keys) and configs[3] (10^8 key + payload pairs) run the
    same way as the headline -- pre-staged batches, back-to-back steps, every output verified on the device, the first batch of every
    seed bit for bit against std::sort / std::stable_sort -- in short form (their full lines: --n 1e7 / --pairs)."""
    import copy
    out = {}
    a1 = copy.copy(args)
    a1.n, a1.pairs = 10 ** 7, False
    r = bench_single(a1)
    out["configs[1]"] = {"workload": r["config"]["workload"], "path": r["config"]["path"].split(":")[0].split(" --")[0], "value": r["value"],
                         "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": r["steps"],
                         "ms_per_step_individually_timed": r["ms_per_step_individually_timed"],
                         "sort_roofline": r["sort_roofline"], "contract_path_ms_per_step": r.get("contract_path", {}).get("ms_per_step"),
                         "bit_exact_vs_std_sort": r["verified"].get("one_call_bit_exact_vs_std_sort_seeds"),
                         "every_timed_output_verified_on_device": r["verified"].get("timed_region_every_batch_ascending_and_permutation_of_its_input"),
                         "cpu_baseline": r.get("cpu_baseline", {}).get("value")}
    a3 = copy.copy(args)
    a3.n, a3.pairs, a3.steps, a3.exact_seeds = 10 ** 8, True, min(args.steps, 10), 1  # (one seed: its std::stable_sort takes 9 s)
    r = bench_pairs(a3)
    out["configs[3]"] = {"workload": r["config"]["workload"], "path": r["config"]["path"].split(":")[0], "value": r["value"], "unit": r["unit"],
                         "ms_per_step": r["ms_per_step"], "steps": r["steps"], "sort_roofline": r["sort_roofline"],
                         "roofline_frac_dominant_kernel": r["roofline"]["frac"],
                         "bit_exact_vs_std_stable_sort": r["verified"].get("bit_exact_vs_std_stable_sort_seeds"),
                         "every_timed_output_verified_on_device": r["verified"].get("timed_region_every_batch_keys_ascending_and_permutations"),
                         "cpu_baseline": r.get("cpu_baseline", {}).get("value")}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=lambda s: int(float(s)), default=10 ** 8, help="keys per GPU")
    ap.add_argument("--blocks", type=int, default=32, help="NUM_BLOCKS_PER_WORKGROUP")
    ap.add_argument("--rank-mode", type=int, default=0, help="0 auto, 1 ballot, 2 LDS atomic")
    ap.add_argument("--variant", type=int, default=0, help="scatter variant code (tuning)")
    ap.add_argument("--rounds", type=int, default=4, help="multi-GPU: sub-ranges per rank (exchange/sort pipelining)")
    ap.add_argument("--rounds-forced", action="store_true", help="use --rounds even at world size 1 (testing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="N = 1, default workload: leave out the `configs` block (10^7 keys, 10^8 pairs)")
    ap.add_argument("--loopback-ranks", type=int, default=0,
                    help="the multi-GPU step with this many ranks as THREADS of one process over the library's in-process transport "
                         "(one GPU: all ranks share it -- exercises every rank-to-rank path, measures no scaling)")
    ap.add_argument("--event-every", type=int, default=4,
                    help="N = 1: the dominant kernel's launches carry HIP events in every this-many-th step of the timed region")
    ap.add_argument("--tune", action="append", help="lab switch: KEY=VALUE, a vrs_tuning id and its value (repeatable); the default line never uses it")
    ap.add_argument("--pairs", action="store_true", help="N = 1: BASELINE.json configs[3], key + payload pairs through vrs_sort_pairs_u32")
    ap.add_argument("--dist-path", choices=["c", "python"], default="c",
                    help="N > 1: the step behind the C ABI (vrs_dist_*, hybrid shape; default) or the Python orchestration "
                         "(vkradixsort_amd/distributed.py over torch.distributed)")
    ap.add_argument("--path", choices=["one_call", "contract"], default="one_call",
                    help="N = 1: which path is timed as `value` (the other is reported beside it): the one-call sort "
                         "(one counting read + look-back scatters) or the reference's stage-by-stage contract path")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # convenience: re-launch under torch.distributed.run the way the driver does
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29511"), __file__] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    from vkradixsort_amd import capi
    capi.load_library()  # fails loudly if the HIP extension was not built; there is no fallback path

    if args.loopback_ranks > 0:
        result = bench_multi_loopback(args)
        if result is not None:
            print(json.dumps(result), flush=True)
        return
    if args.gpus > 1 or os.environ.get("VRS_BENCH_FORCE_MULTI") == "1":  # the latter: exchange path at world size 1
        # RCCL prints its banner on the C stdout whenever a communicator first does something: fd 1 belongs to the ONE JSON
        # line, everything else goes to fd 2 for the whole run
        guard = _StdoutToStderr()
        guard.__enter__()
        result = bench_multi(args) if args.dist_path == "c" else bench_multi_python(args)
        sys.stdout.flush()
        if result is not None:
            os.write(guard._saved, (json.dumps(result) + "\n").encode())
        return
    result = bench_pairs(args) if args.pairs else bench_single(args)
    if result is not None and args.tune:
        result["config"]["lab_tunings"] = list(args.tune)  # (not the default line)
    if result is not None and not args.pairs and args.n == 10 ** 8 and args.path == "one_call" and not args.no_configs:
        result["configs"] = other_configs(args)
    if result is not None:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
