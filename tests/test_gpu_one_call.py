"""GPU parity tests of the one-call sort's large-N form (K5 in vrs_one_call.hip): ONE counting read of the keys
(digit_tables_kernel), then four scatter passes that find their offsets by decoupled look-back along 8 streams (one per XCD).
The reference has no such entry point (its loop is MultiRadixSort.cpp:50-61); the acceptance criterion is the
reference's own: the output must equal std::sort (MultiRadixSort.cpp:141-161), bit for bit, and pairs must
equal std::stable_sort by key."""
import ctypes

import numpy as np
import pytest

import vkradixsort_amd as vrs
from vkradixsort_amd import capi

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def pool_form_from_its_round_4_threshold(gpu_context):
    """The subjects of this file are the LSD form and the counted hybrid form at test sizes of a few million keys.  Since round 5 the
    pool form takes bare uint32 keys from 2^22 on (tests/test_gpu_pool.py): here it keeps round 4's threshold on the shared context,
    so that the sizes below reach the forms they are about and 10^8 keys still take the pool form."""
    gpu_context.setTuning(capi.VRS_TUNE_MSD_POOL_MIN_KEYS, 32000000)
    yield
    gpu_context.setTuning(capi.VRS_TUNE_MSD_POOL_MIN_KEYS, 1 << 22)
S = vrs.Buffer.BufferSettings


def make_keys(n, dist, seed=7):
    rs = np.random.RandomState(seed)
    k = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32)
    if dist == "uniform":
        return k
    if dist == "28bit":  # the reference's own generator, MultiRadixSort.cpp:121-133
        return k >> np.uint32(4)
    if dist == "mult256":  # digit 0 constant: every key of pass 1 falls into one group -> that pass must fall back
        return k & np.uint32(0xFFFFFF00)
    if dist == "lowbyte":  # passes 1-3 are the identity: one pass runs, the result is copied home
        return k & np.uint32(0xFF)
    if dist == "16bit":  # passes 2-3 are the identity: two passes run
        return k & np.uint32(0xFFFF)
    if dist == "sorted":
        return np.sort(k)
    if dist == "reverse":
        return np.sort(k)[::-1].copy()
    if dist == "const":
        return np.full(n, 0xDEADBEEF, dtype=np.uint32)
    if dist == "two_values":
        return np.where(k & 1, np.uint32(0xFFFFFFFF), np.uint32(0)).astype(np.uint32)
    if dist == "max_keys":  # the padding value of a ragged tile is a legal key
        return np.where(k % 3 == 0, np.uint32(0xFFFFFFFF), k).astype(np.uint32)
    if dist == "skewed_stream":  # 40 % of the keys share one digit-1 group: no cut between groups balances pass 2
        heavy = (k % 5) < 2
        return np.where(heavy, (k & np.uint32(0xFFFF07FF)) | np.uint32(0x2800), k).astype(np.uint32)
    if dist == "clustered":  # few distinct top bytes, long runs
        return (np.sort(k >> np.uint32(8)) | (rs.randint(0, 4, size=n).astype(np.uint32) << np.uint32(30))).astype(np.uint32)
    raise ValueError(dist)


def launches(ctx, kid):
    return ctx.profileQuery(kid)[0]


def identity_passes(keys):
    """Passes the one-call sort leaves out: every key has the same digit there, so the pass is the identity (the
    plan marks it; the workgroups of the speculatively enqueued pass leave at once)."""
    nbytes = keys.dtype.itemsize
    skipped = 0
    for p in range(nbytes):
        d = (keys >> keys.dtype.type(8 * p)) & keys.dtype.type(255)
        skipped += int(d.min() == d.max())
    return skipped


def sort_keys(ctx, keys, min_keys=1):
    n = keys.size
    ctx.setTuning(capi.VRS_TUNE_ONE_CALL_MIN_KEYS, min_keys)
    k0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), keys)
    k1 = vrs.Buffer(ctx, S(4 * n))
    ctx.profileReset()
    ctx.profileEnable(True)
    try:
        ctx.check(ctx.lib.vrs_sort_keys_u32(ctx.handle, k0.handle, k1.handle, n))
        out = np.empty(n, np.uint32)
        k0.downloadWithStagingBuffer(out)
        stats = {name: launches(ctx, kid) for kid, name in capi.KERNEL_NAMES.items()}
    finally:
        ctx.profileEnable(False)
        ctx.setTuning(capi.VRS_TUNE_ONE_CALL_MIN_KEYS, capi.ONE_CALL_MIN_KEYS_DEFAULT)
        k0.release()
        k1.release()
    return out, stats


DISTS = ["uniform", "28bit", "mult256", "lowbyte", "16bit", "sorted", "reverse", "const", "two_values", "max_keys",
         "skewed_stream", "clustered"]


@pytest.mark.parametrize("dist", DISTS)
@pytest.mark.parametrize("n", [(1 << 20) + 1, 3000001])
def test_one_read_sort_equals_std_sort(gpu_context, oracle, n, dist):
    keys = make_keys(n, dist, seed=n % 1000)
    out, stats = sort_keys(gpu_context, keys)
    assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1
    # the one-read form really ran: one counting read, and every pass is either a look-back scatter or a contract pass
    assert stats["digit_tables"] == 1
    assert stats["lookback_scatter"] + stats["scatter"] == 4 - identity_passes(keys) and stats["histogram"] == stats["scatter"]
    if dist in ("uniform", "28bit", "sorted", "reverse"):
        assert stats["lookback_scatter"] == 4
    if dist in ("mult256", "skewed_stream", "two_values"):
        assert stats["scatter"] >= 1  # unbalanced streams -> contract pass
    if dist == "lowbyte":
        assert stats["lookback_scatter"] == 1 and stats["scatter"] == 0  # passes 1-3 are the identity: left out
    if dist == "const":
        assert stats["lookback_scatter"] == 0 and stats["scatter"] == 0  # every pass is the identity


@pytest.mark.parametrize("n", [1 << 20, (1 << 20) + 8191, (1 << 22) - 1, 5000000, (1 << 23) + 12345])
def test_one_read_and_contract_passes_agree(gpu_context, n):
    keys = make_keys(n, "uniform", seed=n % 977)
    a, sa = sort_keys(gpu_context, keys, min_keys=1)
    b, sb = sort_keys(gpu_context, keys, min_keys=0)
    assert sa["lookback_scatter"] == 4 and sb["lookback_scatter"] == 0 and sb["scatter"] == 4
    assert np.array_equal(a, b) and np.array_equal(a, np.sort(keys))


def test_one_call_stats_count_what_happened(gpu_context):
    ctx = gpu_context

    def stats():
        a, b, c = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        ctx.check(ctx.lib.vrs_one_call_stats(ctx.handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return np.array([a.value, b.value, c.value], dtype=np.int64)

    s0 = stats()
    sort_keys(ctx, make_keys(1 << 21, "uniform"))
    s1 = stats()
    assert (s1 - s0).tolist() == [4, 0, 0]
    sort_keys(ctx, make_keys(1 << 21, "lowbyte"))  # passes 1-3 are the identity
    s2 = stats()
    assert (s2 - s1).tolist() == [1, 0, 3]
    sort_keys(ctx, make_keys(1 << 21, "mult256"))  # pass 0 is the identity, pass 1's keys all sit in one group
    s3 = stats()
    assert (s3 - s2).tolist() == [2, 1, 1]


def test_mildly_unbalanced_streams_take_the_relaunch_path(gpu_context):
    """Passes 1-3 are enqueued before the plan is known, with grids sized for (nearly) even streams.  Keys whose
    digit 0 is a little skewed make pass 1's longest stream ~12 % longer than even: too long for the speculative
    grid, well within what a look-back pass handles -- that pass (and the ones after it) must leave at once and be
    enqueued again with their exact grids.  Same result, four look-back passes, no contract pass."""
    ctx = gpu_context
    n = 6000007
    rs = np.random.RandomState(77)
    k = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32)
    extra = rs.rand(n) < 0.017  # 1.7 % more keys with digit 0 in [0, 8): stream 0 of pass 1 = groups 0-3 grows by ~13 %
    k = np.where(extra, (k & np.uint32(0xFFFFFF07)), k).astype(np.uint32)
    r0, s0 = ctypes.c_uint64(), [ctypes.c_uint64() for _ in range(3)]
    ctx.check(ctx.lib.vrs_one_call_relaunched_passes(ctx.handle, ctypes.byref(r0)))
    ctx.check(ctx.lib.vrs_one_call_stats(ctx.handle, *[ctypes.byref(x) for x in s0]))
    out, stats = sort_keys(ctx, k)
    r1, s1 = ctypes.c_uint64(), [ctypes.c_uint64() for _ in range(3)]
    ctx.check(ctx.lib.vrs_one_call_relaunched_passes(ctx.handle, ctypes.byref(r1)))
    ctx.check(ctx.lib.vrs_one_call_stats(ctx.handle, *[ctypes.byref(x) for x in s1]))
    assert np.array_equal(out, np.sort(k))
    assert stats["lookback_scatter"] == 4 and stats["scatter"] == 0
    assert [b.value - a.value for a, b in zip(s0, s1)] == [4, 0, 0]
    assert r1.value - r0.value == 3  # passes 1, 2, 3 went out twice


def test_threshold_selects_the_form(gpu_context):
    keys = make_keys(600000, "uniform")
    _, below = sort_keys(gpu_context, keys, min_keys=1 << 20)
    assert below["digit_tables"] == 0 and below["scatter"] == 4
    _, above = sort_keys(gpu_context, keys, min_keys=500000)
    assert above["digit_tables"] == 1 and above["lookback_scatter"] == 4
    with pytest.raises(vrs.VrsError):
        gpu_context.setTuning(capi.VRS_TUNE_ONE_CALL_MIN_KEYS, -1)


@pytest.mark.parametrize("dist", ["uniform", "mult256", "two_values", "sorted", "lowbyte", "16bit"])
def test_one_read_pairs_are_stable(gpu_context, oracle, dist):
    ctx, lib, n = gpu_context, gpu_context.lib, 2500003
    keys = make_keys(n, dist) if dist != "uniform" else (make_keys(n, "uniform") & np.uint32(0x00FFFFFF))
    vals = np.arange(n, dtype=np.uint32)
    k0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), keys)
    v0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), vals)
    k1, v1 = vrs.Buffer(ctx, S(4 * n)), vrs.Buffer(ctx, S(4 * n))
    ctx.check(lib.vrs_sort_pairs_u32(ctx.handle, k0.handle, k1.handle, v0.handle, v1.handle, n))
    ok, ov = np.empty(n, np.uint32), np.empty(n, np.uint32)
    k0.downloadWithStagingBuffer(ok)
    v0.downloadWithStagingBuffer(ov)
    rk, rv, _ = oracle.stable_sort_pairs(keys, vals)
    assert np.array_equal(ok, rk) and np.array_equal(ov, rv)
    for b in (k0, k1, v0, v1):
        b.release()


@pytest.mark.parametrize("shift,min_keys", [(40, capi.ONE_CALL_MIN_KEYS_DEFAULT), (40, 0), (0, capi.ONE_CALL_MIN_KEYS_DEFAULT), (17, capi.ONE_CALL_MIN_KEYS_DEFAULT)])
def test_one_call_pairs_u64(gpu_context, shift, min_keys):
    """uint64 keys + uint32 payloads in one call, stable: two groups of [counting read + four look-back passes] (round 3), or the
    eight contract passes below the one-call threshold (min_keys 0 = never the look-back form); 24-bit values (plenty of ties, six
    identity passes), full 64-bit keys, 47-bit keys."""
    ctx, lib, n = gpu_context, gpu_context.lib, 1300001
    ctx.setTuning(capi.VRS_TUNE_ONE_CALL_MIN_KEYS, min_keys)
    keys = make_keys64(n, "uniform") >> np.uint64(shift)
    vals = np.arange(n, dtype=np.uint32)
    k0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(8 * n), keys)
    v0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), vals)
    k1, v1 = vrs.Buffer(ctx, S(8 * n)), vrs.Buffer(ctx, S(4 * n))
    ctx.check(lib.vrs_sort_pairs_u64(ctx.handle, k0.handle, k1.handle, v0.handle, v1.handle, n))
    ok, ov = np.empty(n, np.uint64), np.empty(n, np.uint32)
    k0.downloadWithStagingBuffer(ok)
    v0.downloadWithStagingBuffer(ov)
    ctx.setTuning(capi.VRS_TUNE_ONE_CALL_MIN_KEYS, capi.ONE_CALL_MIN_KEYS_DEFAULT)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(ok, keys[order]) and np.array_equal(ov, vals[order])
    for b in (k0, k1, v0, v1):
        b.release()


@pytest.mark.parametrize("dist,n", [("uniform", (1 << 22) + 12345), ("44bit", (1 << 22) + 1), ("ties", 5000011), ("sorted", 4500001), ("low32", 4200001),
                                    ("big_buckets", 70000001)])
def test_hybrid_form_for_u64_keys_with_payloads(gpu_context, dist, n):
    """vrs_sort_pairs_u64 in the hybrid form (round 5; the reference's SORT_64_BIT stub, MultiRadixSort.h:10-18, has neither 64-bit
    keys nor payloads): one counting read, two STABLE MSD passes (look-back), the buckets sorted inside LDS with their payloads
    -- instead of two groups of four look-back passes.  Equal to a stable argsort, payloads included; plenty of equal keys
    ("ties": 2^20 distinct values), 32-bit values in 64-bit keys (two LDS passes), buckets beyond the small local sort (7e7 pairs:
    4270 per bucket)."""
    ctx, lib = gpu_context, gpu_context.lib
    rs = np.random.RandomState(n % 1000)
    keys = make_keys64(n if dist != "big_buckets" else n, "uniform" if dist in ("ties", "big_buckets") else dist, seed=n % 89)
    if dist == "ties":
        keys = (keys >> np.uint64(44)) << np.uint64(44)
    vals = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32)
    ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 1 << 23)  # (64-bit keys: half of it)
    ctx.setTuning(capi.VRS_TUNE_HYBRID, 1)
    k0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(8 * n), keys)
    v0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), vals)
    k1, v1 = vrs.Buffer(ctx, S(8 * n)), vrs.Buffer(ctx, S(4 * n))
    h0 = hybrid_sorts(ctx)
    ctx.profileReset()
    ctx.profileEnable(True)
    try:
        ctx.check(lib.vrs_sort_pairs_u64(ctx.handle, k0.handle, k1.handle, v0.handle, v1.handle, n))
        ok, ov = np.empty(n, np.uint64), np.empty(n, np.uint32)
        k0.downloadWithStagingBuffer(ok)
        v0.downloadWithStagingBuffer(ov)
        stats = {name: launches(ctx, kid) for kid, name in capi.KERNEL_NAMES.items()}
    finally:
        ctx.profileEnable(False)
        ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 0)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(ok, keys[order]) and np.array_equal(ov, vals[order])
    took = hybrid_sorts(ctx) - h0
    assert took == 1 and stats["local_sort"] == 1 and stats["lookback_scatter"] == 2 and stats["digit_tables"] == 1, stats
    for b in (k0, k1, v0, v1):
        b.release()


@pytest.mark.parametrize("kind", ["keys", "pairs", "u64"])
def test_engine_mirror_one_call_flag(gpu_context, kind):
    """engine.MultiRadixSort.m_oneCallSort (C++: engine::MultiRadixSort::m_oneCallSort): same execute(), same checks."""
    n = 1500003
    keys = make_keys64(n, "uniform") if kind == "u64" else make_keys(n, "uniform", seed=21)
    vals = np.arange(n, dtype=np.uint32)[::-1].copy() if kind == "pairs" else None
    app = vrs.MultiRadixSort(keys=keys, values=vals, quiet=True)
    app.m_oneCallSort = True
    gpu_context.profileReset()
    gpu_context.profileEnable(True)
    try:
        app.execute(gpu_context)  # raises "TEST FAILED." on any mismatch, like the reference
        assert launches(gpu_context, capi.VRS_KERNEL_LOOKBACK_SCATTER) == (8 if kind == "u64" else 4)
    finally:
        gpu_context.profileEnable(False)
    assert np.array_equal(app.sorted_keys, np.sort(keys))


def test_context_reuse_across_sizes(gpu_context):
    """The digit tables must come back zeroed and the status words re-armed whatever the previous sort was."""
    for n, dist in [(1 << 22, "uniform"), ((1 << 20) + 7, "mult256"), (5000000, "28bit"), (1 << 22, "const"),
                    (3000000, "uniform"), (1 << 21, "sorted")]:
        keys = make_keys(n, dist, seed=n % 101)
        out, stats = sort_keys(gpu_context, keys)
        assert stats["digit_tables"] == 1
        assert np.array_equal(out, np.sort(keys)), (n, dist)


def test_look_back_does_not_depend_on_xcd_placement(gpu_context):
    """The fast hand-off of the look-back words assumes a stream's tiles share one XCD's L2; the test hook moves
    every other tile to the neighbouring XCD: those workgroups must notice (HW_REG_XCC_ID) and take the
    placement-independent route -- slower, same result, no hang."""
    ctx = gpu_context
    n = (1 << 21) + 4097
    keys = make_keys(n, "uniform", seed=5)
    ctx.setTuning(capi.VRS_TUNE_DEBUG_MISPLACE_STREAMS, 1)
    try:
        out, stats = sort_keys(ctx, keys)
    finally:
        ctx.setTuning(capi.VRS_TUNE_DEBUG_MISPLACE_STREAMS, 0)
    assert stats["lookback_scatter"] == 4
    assert np.array_equal(out, np.sort(keys))


def xcc_placement(ctx):
    r, m, v = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_int()
    ctx.check(ctx.lib.vrs_debug_xcc_placement(ctx.handle, ctypes.byref(r), ctypes.byref(m), ctypes.byref(v)))
    return r.value, m.value, v.value


@pytest.mark.parametrize("rotate", [1, 5])
def test_sorts_that_find_the_placement_rotated_have_it_probed_again(rotate):
    """The round-robin of a launch's blocks over the XCCs starts at an XCC of the hardware queue's own, and a stream may move to
    another queue after the context probed (round 5 saw it happen between two tests).  The look-back streams and the counted form's
    reservations then take their placement-independent routes -- exact -- and their first workgroups report it; the next sort call
    probes again: ONE sort on the slow routes, not all of them.  Test hook: the probed order rotated."""
    with vrs.GPUContext(0) as gpu:
        gpu.setTuning(capi.VRS_TUNE_MSD_POOL, 0)  # (the pool form takes its lists by the XCC a workgroup runs on: nothing to report)
        n = 3000017
        keys = make_keys(n, "uniform", seed=rotate)
        out, _ = sort_keys(gpu, keys)  # (makes the one-call scratch, and with it the word the reports go to)
        assert np.array_equal(out, np.sort(keys))
        reprobes0, good_map, valid = xcc_placement(gpu)
        assert valid == 1
        gpu.setTuning(capi.VRS_TUNE_DEBUG_XCC_ROTATE, rotate)
        assert xcc_placement(gpu)[1] != good_map
        out, _ = sort_keys(gpu, keys)  # every tile off its stream's XCC: exact all the same
        assert np.array_equal(out, np.sort(keys))
        assert xcc_placement(gpu)[0] == reprobes0  # nobody has looked yet
        out, _ = sort_keys(gpu, keys)  # this call looks first
        assert np.array_equal(out, np.sort(keys))
        reprobes, now_map, valid = xcc_placement(gpu)
        assert reprobes == reprobes0 + 1 and now_map == good_map and valid == 1
        out, _ = sort_keys(gpu, keys)  # ... and nothing more to report
        assert np.array_equal(out, np.sort(keys)) and xcc_placement(gpu)[0] == reprobes
        # the counted hybrid form (reserving MSD passes) reports the same way
        gpu.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 1 << 22)
        big = make_keys(6000011, "uniform", seed=7)
        gpu.setTuning(capi.VRS_TUNE_DEBUG_XCC_ROTATE, rotate)
        for _ in range(2):
            out, stats = sort_keys(gpu, big)
            assert np.array_equal(out, np.sort(big)) and stats["local_sort"] == 1
        assert xcc_placement(gpu)[0] == reprobes + 1 and xcc_placement(gpu)[1] == good_map


@pytest.mark.parametrize("stray", [9, 4095])
def test_a_placement_that_holds_for_most_blocks_only_switches_the_l2_local_forms_off(stray):
    """The look-back streams, the reservation cursors and the pool form's regions share L2-resident words between the workgroups of
    one XCD: they rest on "block b runs on XCC f(b % 8)", which the context probes at creation (and every workgroup re-checks).  If the
    probe finds the rule broken for even ONE block (test hook: block `stray` pretended elsewhere) none of those forms may run: the
    sorts take the contract stages -- agent-scope data only -- whatever the size, and the hybrid form's halves refuse to start."""
    with vrs.GPUContext(0) as gpu:
        lib = gpu.lib
        gpu.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 1 << 22)
        gpu.setTuning(capi.VRS_TUNE_MSD_POOL_MIN_KEYS, 1 << 22)
        gpu.setTuning(capi.VRS_TUNE_MSD_POOL, 2)
        gpu.setTuning(capi.VRS_TUNE_DEBUG_XCC_STRAY_BLOCK, stray)
        for n in (3000001, (1 << 23) + 5):
            keys = make_keys(n, "uniform", seed=n % 89)
            out, stats = sort_keys(gpu, keys)
            assert np.array_equal(out, np.sort(keys))
            assert stats["digit_tables"] == 0 and stats["lookback_scatter"] == 0 and stats["local_sort"] == 0 and stats["pool_sample"] == 0
            assert stats["histogram"] == 4 and stats["scatter"] == 4
        n = (1 << 22) + 1
        keys = make_keys(n, "uniform", seed=3)
        vals = np.arange(n, dtype=np.uint32)
        ok, ov = sort_pairs_once(gpu, keys, vals)
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(ok, keys[order]) and np.array_equal(ov, vals[order])
        kb = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), keys)
        grouped, counts = vrs.Buffer(gpu, S(4 * n)), vrs.Buffer(gpu, S(4 * capi.MSD_COUNT_WORDS))
        assert lib.vrs_msd_partition_u32(gpu.handle, kb.handle, grouped.handle, counts.handle, n) == capi.VRS_ERROR_INVALID_ARGUMENT
        # probed again without the pretence: everything is back
        gpu.setTuning(capi.VRS_TUNE_DEBUG_XCC_STRAY_BLOCK, -1)
        gpu.check(lib.vrs_msd_partition_u32(gpu.handle, kb.handle, grouped.handle, counts.handle, n))
        keys = make_keys((1 << 23) + 5, "uniform", seed=8)
        out, stats = sort_keys(gpu, keys)
        assert np.array_equal(out, np.sort(keys)) and stats["pool_sample"] == 1 and stats["histogram"] == 0
        for b in (kb, grouped, counts):
            b.release()


@pytest.mark.parametrize("n", [1, 2, 255, 256, 257, 1000, 4095, 4096, 4097, 5000, 10000, 70000])
def test_small_n_goes_to_the_single_workgroup_kernel(gpu_context, oracle, n):
    """f1 of SURVEY section 8: vrs_sort_keys_u32 runs small inputs as ONE single_radixsort launch (the reference's
    guidance, README.md:18-21) and larger ones through the multi-block passes; same result on both sides of the
    threshold (default 4096 keys), bit-equal to std::sort and to the oracle's single_radixsort restatement."""
    keys = make_keys(n, "uniform", seed=n)
    out, stats = sort_keys(gpu_context, keys, min_keys=1 << 20)
    assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1
    assert np.array_equal(out, oracle.single_radixsort(keys))
    if n <= 4096:
        assert stats.get("single", 0) == 1 and stats.get("scatter", 0) == 0
    else:
        assert stats.get("single", 0) == 0 and stats["scatter"] == 4
    # the threshold is a tuning knob: 0 switches the single-launch form off
    gpu_context.setTuning(capi.VRS_TUNE_SINGLE_MAX_KEYS, 0)
    try:
        out2, stats2 = sort_keys(gpu_context, keys, min_keys=1 << 20)
    finally:
        gpu_context.setTuning(capi.VRS_TUNE_SINGLE_MAX_KEYS, 4096)
    assert stats2.get("single", 0) == 0 and np.array_equal(out2, out)


@pytest.mark.parametrize("dist", ["uniform", "const", "sorted", "16bit", "mult256", "max_keys"])
@pytest.mark.parametrize("n", [4097, 5000, 8191, 8192, 8193, 20011, 70001, 131072, 300007])
def test_one_read_form_at_small_sizes(gpu_context, oracle, n, dist):
    """The one-read form is the default from 2^13 keys on (it is the faster form at every size above the single-launch
    threshold): a handful of tiles, most of the eight streams empty or one tile long, ragged last tiles."""
    keys = make_keys(n, dist, seed=n % 211)
    out, stats = sort_keys(gpu_context, keys, min_keys=1)
    assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1
    assert stats["digit_tables"] == 1 and stats["single"] == 0
    assert stats["lookback_scatter"] + stats["scatter"] == 4 - identity_passes(keys)
    pairs_keys = keys & np.uint32(0xFFFF)  # plenty of ties: the payloads must stay in input order
    vals = np.arange(n, dtype=np.uint32)
    ctx, lib = gpu_context, gpu_context.lib
    k0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), pairs_keys)
    v0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), vals)
    k1, v1 = vrs.Buffer(ctx, S(4 * n)), vrs.Buffer(ctx, S(4 * n))
    ctx.setTuning(capi.VRS_TUNE_ONE_CALL_MIN_KEYS, 1)
    try:
        ctx.check(lib.vrs_sort_pairs_u32(ctx.handle, k0.handle, k1.handle, v0.handle, v1.handle, n))
    finally:
        ctx.setTuning(capi.VRS_TUNE_ONE_CALL_MIN_KEYS, capi.ONE_CALL_MIN_KEYS_DEFAULT)
    ok, ov = np.empty(n, np.uint32), np.empty(n, np.uint32)
    k0.downloadWithStagingBuffer(ok)
    v0.downloadWithStagingBuffer(ov)
    order = np.argsort(pairs_keys, kind="stable")
    assert np.array_equal(ok, pairs_keys[order]) and np.array_equal(ov, vals[order])
    for b in (k0, k1, v0, v1):
        b.release()


@pytest.mark.parametrize("budget", [16, None])
@pytest.mark.parametrize("dist", ["uniform", "sorted"])
def test_look_back_never_waits_forever(gpu_context, dist, budget):
    """Bounded spin: a tile whose predecessor never publishes its look-back row (the test hook withholds tile 3 of
    every stream in every pass) must run out of budget, count its stream's earlier keys itself and carry on --
    same result, no hang.  budget None = the default (a few milliseconds per pass here)."""
    ctx = gpu_context
    n = (1 << 22) + 77
    keys = make_keys(n, dist, seed=13)
    ctx.setTuning(capi.VRS_TUNE_DEBUG_HOLD_TILE, 3)
    if budget is not None:
        ctx.setTuning(capi.VRS_TUNE_LOOKBACK_SPIN_BUDGET, budget)
    try:
        out, stats = sort_keys(ctx, keys)
    finally:
        ctx.setTuning(capi.VRS_TUNE_DEBUG_HOLD_TILE, -1)
        ctx.setTuning(capi.VRS_TUNE_LOOKBACK_SPIN_BUDGET, 4096)
    assert stats["lookback_scatter"] == 4
    assert np.array_equal(out, np.sort(keys))


@pytest.mark.parametrize("groups", [8, 16, 32])
def test_digit_table_group_counts_agree(gpu_context, groups):
    """The counting read's group count (8 / 16 / 32 per pass) only changes how finely the streams can follow the
    data, never the result."""
    ctx = gpu_context
    ctx.setTuning(capi.VRS_TUNE_DIGIT_TABLE_GROUPS, groups)
    try:
        for n, dist in [((1 << 21) + 5, "uniform"), (3000001, "skewed_stream"), ((1 << 20) + 1, "mult256"), (2500000, "clustered")]:
            keys = make_keys(n, dist, seed=groups)
            out, stats = sort_keys(ctx, keys)
            assert stats["digit_tables"] == 1
            assert np.array_equal(out, np.sort(keys)), (groups, n, dist)
            # the fused form (the counting read's last workgroup makes the plan) gives the same plan
            ctx.setTuning(capi.VRS_TUNE_FUSED_PLAN, 1)
            try:
                out2, stats2 = sort_keys(ctx, keys)
            finally:
                ctx.setTuning(capi.VRS_TUNE_FUSED_PLAN, 0)
            assert np.array_equal(out2, out) and stats2["lookback_scatter"] == stats["lookback_scatter"], (groups, n, dist)
    finally:
        ctx.setTuning(capi.VRS_TUNE_DIGIT_TABLE_GROUPS, 0)
    with pytest.raises(vrs.VrsError):
        ctx.setTuning(capi.VRS_TUNE_DIGIT_TABLE_GROUPS, 12)


@pytest.mark.parametrize("mode", ["keys", "pairs"])
def test_soak_cut(mode):
    """A 12-second cut of tools/soak_one_call.py: back-to-back large sorts of varying size and distribution on a
    borrowed torch stream, each verified on the device (ascending + fingerprint / payload order).  The inter-workgroup
    hand-off of the look-back is the one timing-dependent part of the path."""
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("soak_one_call", Path(__file__).resolve().parent.parent / "tools" / "soak_one_call.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cases, keys_sorted, bad = mod.soak(12.0, seed=5, mode=mode, max_keys=3 * 10 ** 7)
    assert bad is None, bad
    assert cases >= 5


@pytest.mark.parametrize("offset_keys", [1, 2, 3])
def test_unaligned_keys(gpu_context, offset_keys):
    """The counting read uses 16-byte loads: a wrapped pointer that is only 4-byte aligned (a sub-range of a larger
    allocation, as the multi-GPU exchange produces) must peel its way to alignment and touch nothing outside."""
    ctx, lib = gpu_context, gpu_context.lib
    n = (1 << 20) + 77
    keys = make_keys(n, "uniform", seed=3 + offset_keys)
    big = vrs.Buffer(ctx, S(4 * (n + 8)))
    tmp = vrs.Buffer(ctx, S(4 * n))
    host = np.concatenate([np.full(offset_keys, 0xAAAAAAAA, np.uint32), keys, np.full(8 - offset_keys, 0xBBBBBBBB, np.uint32)])
    ctx.check(lib.vrs_buffer_upload(ctx.handle, big.handle, host.ctypes.data_as(ctypes.c_void_p), host.nbytes))
    view = vrs.Buffer(ctx, S(4 * n), device_ptr=big.getDeviceAddress() + 4 * offset_keys)
    ctx.profileReset()
    ctx.profileEnable(True)
    try:
        ctx.check(lib.vrs_sort_keys_u32(ctx.handle, view.handle, tmp.handle, n))
        ctx.waitIdle()
        assert launches(ctx, capi.VRS_KERNEL_DIGIT_TABLES) == 1 and launches(ctx, capi.VRS_KERNEL_LOOKBACK_SCATTER) == 4
    finally:
        ctx.profileEnable(False)
    ctx.check(lib.vrs_buffer_download(ctx.handle, big.handle, host.ctypes.data_as(ctypes.c_void_p), host.nbytes))
    assert (host[:offset_keys] == 0xAAAAAAAA).all() and (host[n + offset_keys:] == 0xBBBBBBBB).all()
    assert np.array_equal(host[offset_keys:n + offset_keys], np.sort(keys))
    for b in (view, big, tmp):
        b.release()


def make_keys64(n, dist, seed=11):
    rs = np.random.RandomState(seed)
    k = (rs.randint(0, 2 ** 32, n, dtype=np.uint64) << np.uint64(32)) | rs.randint(0, 2 ** 32, n, dtype=np.uint64)
    if dist == "uniform":
        return k
    if dist == "44bit":  # the reference's SORT_64_BIT generator, MultiRadixSort.cpp:128
        return k >> np.uint64(20)
    if dist == "low32":  # upper word zero: passes 5-7 are the identity
        return k & np.uint64(0xFFFFFFFF)
    if dist == "high32":
        return k & np.uint64(0xFFFFFFFF00000000)
    if dist == "sorted":
        return np.sort(k)
    if dist == "max_keys":
        return np.where(k % np.uint64(3) == 0, np.uint64(0xFFFFFFFFFFFFFFFF), k).astype(np.uint64)
    raise ValueError(dist)


@pytest.mark.parametrize("dist", ["uniform", "44bit", "low32", "high32", "sorted", "max_keys"])
@pytest.mark.parametrize("n", [(1 << 20) + 3, 2500001])
def test_one_read_sort_u64(gpu_context, n, dist):
    """64-bit keys (the reference's SORT_64_BIT stub): two groups of four passes, each with its own counting read."""
    ctx, lib = gpu_context, gpu_context.lib
    keys = make_keys64(n, dist, seed=n % 89)
    big = vrs.Buffer(ctx, S(8 * (n + 2)))
    tmp = vrs.Buffer(ctx, S(8 * n))
    host = np.concatenate([np.full(1, 0x1111111111111111, np.uint64), keys, np.full(1, 0x2222222222222222, np.uint64)])
    ctx.check(lib.vrs_buffer_upload(ctx.handle, big.handle, host.ctypes.data_as(ctypes.c_void_p), host.nbytes))
    view = vrs.Buffer(ctx, S(8 * n), device_ptr=big.getDeviceAddress() + 8)  # 8-byte, not 16-byte aligned
    ctx.profileReset()
    ctx.profileEnable(True)
    try:
        ctx.check(lib.vrs_sort_keys_u64(ctx.handle, view.handle, tmp.handle, n))
        ctx.waitIdle()
        stats = {name: launches(ctx, kid) for kid, name in capi.KERNEL_NAMES.items()}
    finally:
        ctx.profileEnable(False)
    ctx.check(lib.vrs_buffer_download(ctx.handle, big.handle, host.ctypes.data_as(ctypes.c_void_p), host.nbytes))
    assert host[0] == 0x1111111111111111 and host[-1] == 0x2222222222222222
    assert np.array_equal(host[1:-1], np.sort(keys))
    assert stats["digit_tables"] == 2 and stats["lookback_scatter"] + stats["scatter"] == 8 - identity_passes(keys)
    if dist in ("uniform", "sorted"):
        assert stats["lookback_scatter"] == 8
    if dist in ("low32", "high32"):
        assert stats["lookback_scatter"] + stats["scatter"] == 4  # four identity passes in the constant word
    for b in (view, big, tmp):
        b.release()


def test_one_read_sort_1e8_equals_std_sort(gpu_context, oracle):
    """BASELINE.json config 3 through the one-call entry point, in its three large-N forms: the pool form (no counting read: a
    sample, two MSD passes, the local sort, 24 B/key: the default at this size), the counted hybrid form (a counting
    read, two MSD passes, the in-place local sort, 28 B/key) and the four LSD look-back passes (36 B/key)."""
    n = 10 ** 8
    keys = np.random.RandomState(1).randint(0, 2 ** 32, size=n, dtype=np.uint32)
    ref = oracle.std_sort(keys)[0]
    gpu_context.setTuning(capi.VRS_TUNE_MSD_POOL, 1)  # (also forgets an earlier test's refusal)
    out, stats = sort_keys(gpu_context, keys)
    assert stats["pool_sample"] == 1 and stats["digit_tables"] == 0 and stats["pool_pass_a"] == 1 and stats["pool_pass_b"] == 1 and stats["local_sort"] == 1
    assert stats["lookback_scatter"] == 0
    assert oracle.test_sort(ref, out) == -1
    gpu_context.setTuning(capi.VRS_TUNE_MSD_POOL, 0)
    try:
        out, stats = sort_keys(gpu_context, keys)
        assert stats["digit_tables"] == 1 and stats["lookback_scatter"] == 2 and stats["local_sort"] == 1 and stats["pool_sample"] == 0
        assert oracle.test_sort(ref, out) == -1
        gpu_context.setTuning(capi.VRS_TUNE_HYBRID, 0)
        out, stats = sort_keys(gpu_context, keys)
    finally:
        gpu_context.setTuning(capi.VRS_TUNE_HYBRID, 1)
        gpu_context.setTuning(capi.VRS_TUNE_MSD_POOL, 1)
    assert stats["lookback_scatter"] == 4 and stats["digit_tables"] == 1 and stats["local_sort"] == 0
    assert oracle.test_sort(ref, out) == -1


def hybrid_sorts(ctx):
    h = ctypes.c_uint64()
    ctx.check(ctx.lib.vrs_one_call_hybrid_sorts(ctx.handle, ctypes.byref(h)))
    return h.value


HYBRID_DISTS = ["uniform", "sorted", "reverse", "28bit", "16bit", "const", "max_keys", "clustered", "two_values", "mult256",
                "low18_const", "one_hot_bucket", "24bit", "rare_high_bit"]


def make_hybrid_keys(n, dist, seed):
    if dist == "low18_const":  # every bucket holds one distinct key: the local sort sees one digit value in both passes
        return make_keys(n, "uniform", seed) & np.uint32(0xFFFC0000)
    if dist == "24bit":
        return make_keys(n, "uniform", seed) >> np.uint32(8)
    if dist == "rare_high_bit":  # 20-bit keys except three that use bit 31: the range probe's sample misses them, the counting
        k = make_keys(n, "uniform", seed) >> np.uint32(12)  # read must notice and the four LSD passes must run
        k[[5, n // 2 + 1, n - 2]] |= np.uint32(0x80000000)
        return k
    if dist == "one_hot_bucket":  # one top-14-bit bucket with far more keys than a workgroup can hold: must fall back
        k = make_keys(n, "uniform", seed)
        k[: n // 20] = (k[: n // 20] & np.uint32(0x3FFFF)) | np.uint32(0x56780000)
        return k
    return make_keys(n, dist, seed)


def probed_shift(keys):
    """The bucket shift the counting read derives from its strided sample of 4096 keys (digit_tables_kernel)."""
    samples = min(keys.size, 4096)
    acc = int(np.bitwise_or.reduce(keys[:: keys.size // samples][:samples]))
    return max(acc.bit_length() - 14, 0)


def hybrid_recounts(ctx):
    import ctypes
    r = ctypes.c_uint64()
    ctx.check(ctx.lib.vrs_one_call_hybrid_recounts(ctx.handle, ctypes.byref(r)))
    return r.value


@pytest.mark.parametrize("fast_count", [0, 2])
@pytest.mark.parametrize("dist", HYBRID_DISTS)
@pytest.mark.parametrize("n", [(1 << 22) + 1, 9000001])
def test_hybrid_form_equals_std_sort(gpu_context, oracle, n, dist, fast_count):
    """The hybrid form at sizes a test can afford (VRS_TUNE_HYBRID_MIN_KEYS lowered to 2^22): buckets of a few hundred
    keys, ragged tiles in every top-byte bucket of the second pass, empty buckets, and distributions whose buckets cannot
    fit a workgroup (the plan must say no and the four LSD passes must run -- from the SAME counting read when it counted
    everything, VRS_TUNE_HYBRID_FAST_COUNT = 0, from a second one when it counted only the bucket histogram, = 2)."""
    ctx = gpu_context
    keys = make_hybrid_keys(n, dist, seed=n % 313)
    ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 1 << 22)
    ctx.setTuning(capi.VRS_TUNE_HYBRID_FAST_COUNT, fast_count)
    # the two MSD passes: with decoupled look-back (beside the full count) or by reservation, the default (beside the fast count)
    ctx.setTuning(capi.VRS_TUNE_MSD_RESERVE, 0 if fast_count == 0 else 1)
    h0, r0 = hybrid_sorts(ctx), hybrid_recounts(ctx)
    try:
        out, stats = sort_keys(ctx, keys)
    finally:
        ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, capi.HYBRID_MIN_KEYS_DEFAULT)
        ctx.setTuning(capi.VRS_TUNE_HYBRID_FAST_COUNT, 1)
        ctx.setTuning(capi.VRS_TUNE_MSD_RESERVE, 1)
    took, recounts = hybrid_sorts(ctx) - h0, hybrid_recounts(ctx) - r0
    assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1
    # the buckets are the top 14 bits of the key RANGE (32-bit keys: bits 18-31, the reference's 28-bit keys: bits 14-27);
    # ranges below 27 bits are left to the LSD passes, which then drop an identity pass
    bits = int(keys.max()).bit_length()
    shift = bits - 14
    fits = 13 <= shift <= 18 and int(np.bincount(keys >> np.uint32(shift), minlength=1 << 14).max()) <= capi.LOCAL_SORT_MAX_KEYS
    assert took == (1 if fits else 0), (dist, bits)
    # a fast count leaves the LSD tables out when the SAMPLED range allows the hybrid form: a refusal then counts again
    recount = fast_count == 2 and not fits and probed_shift(keys) >= 13
    assert recounts == (1 if recount else 0)
    assert stats["digit_tables"] == (2 if recount else 1)
    if fits:
        assert stats["lookback_scatter"] == 2 and stats["local_sort"] == 1 and stats["scatter"] == 0
    else:
        assert stats["local_sort"] == 0 and stats["lookback_scatter"] + stats["scatter"] == 4 - identity_passes(keys)


def test_fast_count_is_armed_by_a_hybrid_sort_and_disarmed_by_a_refusal(gpu_context):
    """VRS_TUNE_HYBRID_FAST_COUNT = 1 (default): the counting read leaves the LSD tables out only while the context's last
    hybrid-capable sort took the hybrid form.  uniform, uniform, hot, hot, uniform, uniform -> one recount (the first hot
    sort), and every output sorted."""
    ctx, n = gpu_context, (1 << 22) + 4097
    uniform = make_keys(n, "uniform", seed=41)
    hot = make_hybrid_keys(n, "one_hot_bucket", seed=42)
    ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 1 << 22)
    ctx.setTuning(capi.VRS_TUNE_HYBRID_FAST_COUNT, 1)  # also disarms
    seen = []
    try:
        for keys in (uniform, uniform, hot, hot, uniform, uniform):
            h0, r0 = hybrid_sorts(ctx), hybrid_recounts(ctx)
            out, stats = sort_keys(ctx, keys)
            assert np.array_equal(out, np.sort(keys))
            seen.append((hybrid_sorts(ctx) - h0, hybrid_recounts(ctx) - r0, stats["digit_tables"]))
    finally:
        ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, capi.HYBRID_MIN_KEYS_DEFAULT)
    assert seen == [(1, 0, 1), (1, 0, 1), (0, 1, 2), (0, 0, 1), (1, 0, 1), (1, 0, 1)]


@pytest.mark.parametrize("reserve", [0, 1])
@pytest.mark.parametrize("hook", ["misplace", "hold", "ballot"])
def test_hybrid_form_under_the_look_back_hooks(gpu_context, hook, reserve):
    """The MSD passes with decoupled look-back (VRS_TUNE_MSD_RESERVE = 0; payloads always) or by reservation (the default for
    bare keys): XCD misplacement and a withheld tile must be survived either way; with ballot ranking forced the hybrid form
    is off (its local sort ranks with returning LDS atomics)."""
    ctx = gpu_context
    ctx.setTuning(capi.VRS_TUNE_MSD_RESERVE, reserve)
    n = (1 << 23) + 4321
    keys = make_keys(n, "uniform", seed=91)
    ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 1 << 22)
    if hook == "misplace":
        ctx.setTuning(capi.VRS_TUNE_DEBUG_MISPLACE_STREAMS, 1)
    elif hook == "hold":
        ctx.setTuning(capi.VRS_TUNE_DEBUG_HOLD_TILE, 2)
        ctx.setTuning(capi.VRS_TUNE_LOOKBACK_SPIN_BUDGET, 8)
    else:
        ctx.setTuning(capi.VRS_TUNE_RANK_MODE, 1)
    h0 = hybrid_sorts(ctx)
    try:
        out, stats = sort_keys(ctx, keys)
    finally:
        ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, capi.HYBRID_MIN_KEYS_DEFAULT)
        ctx.setTuning(capi.VRS_TUNE_DEBUG_MISPLACE_STREAMS, 0)
        ctx.setTuning(capi.VRS_TUNE_DEBUG_HOLD_TILE, -1)
        ctx.setTuning(capi.VRS_TUNE_LOOKBACK_SPIN_BUDGET, 4096)
        ctx.setTuning(capi.VRS_TUNE_RANK_MODE, 0)
        ctx.setTuning(capi.VRS_TUNE_MSD_RESERVE, 1)
    assert np.array_equal(out, np.sort(keys))
    assert hybrid_sorts(ctx) - h0 == (0 if hook == "ballot" else 1)


@pytest.mark.parametrize("hook", ["none", "misplace", "hold"])
@pytest.mark.parametrize("dist", ["uniform", "sorted", "low18_const", "28bit", "ties", "one_hot_bucket", "ragged_streams"])
def test_hybrid_form_msd_passes_by_reservation(gpu_context, dist, hook):
    """VRS_TUNE_MSD_RESERVE = 2: both MSD passes over bare keys take their places with one L2-local atomic add per tile and
    digit instead of the look-back (default from 3e7 keys on; forced here at test sizes).  Any order inside a bucket is fine --
    the result is std::sort either way -- for every bucket shape; with every other tile run OFF its stream's XCD (misplace) those
    tiles take their room from the end of the range with device-wide atomics; a withheld tile has nobody waiting for it.  A
    look-back sort (pairs) and a reserving one alternate on the same context: the status words and the counters stay consistent."""
    ctx = gpu_context
    n = 7000003 if dist != "ragged_streams" else (1 << 22) + 8 * 8192 * 3 + 1234
    if dist == "28bit":
        keys = make_keys(n, "uniform", seed=5) >> np.uint32(4)
    elif dist == "ties":
        keys = (make_keys(n, "uniform", seed=6) & np.uint32(0xFFFF)) * np.uint32(65537)
    elif dist == "ragged_streams":
        keys = make_keys(n, "uniform", seed=7)
    else:
        keys = make_hybrid_keys(n, dist, seed=n % 311)
    ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 1 << 22)
    ctx.setTuning(capi.VRS_TUNE_MSD_RESERVE, 2)
    if hook == "misplace":
        ctx.setTuning(capi.VRS_TUNE_DEBUG_MISPLACE_STREAMS, 1)
    elif hook == "hold":
        ctx.setTuning(capi.VRS_TUNE_DEBUG_HOLD_TILE, 2)
        ctx.setTuning(capi.VRS_TUNE_LOOKBACK_SPIN_BUDGET, 8)
    h0 = hybrid_sorts(ctx)
    try:
        out, _ = sort_keys(ctx, keys)
        assert np.array_equal(out, np.sort(keys))
        m = 1 << 22
        pk, pv = make_keys(m, "uniform", seed=9), np.arange(m, dtype=np.uint32)
        ok, ov = sort_pairs_once(ctx, pk, pv)  # look-back passes (payloads): they write the status words
        order = np.argsort(pk, kind="stable")
        assert np.array_equal(ok, pk[order]) and np.array_equal(ov, pv[order])
        out, _ = sort_keys(ctx, keys[::-1].copy())
        assert np.array_equal(out, np.sort(keys))
    finally:
        ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, capi.HYBRID_MIN_KEYS_DEFAULT)
        ctx.setTuning(capi.VRS_TUNE_MSD_RESERVE, 1)
        ctx.setTuning(capi.VRS_TUNE_DEBUG_MISPLACE_STREAMS, 0)
        ctx.setTuning(capi.VRS_TUNE_DEBUG_HOLD_TILE, -1)
        ctx.setTuning(capi.VRS_TUNE_LOOKBACK_SPIN_BUDGET, 4096)
    shift = int(keys.max()).bit_length() - 14
    fits = 13 <= shift <= 18 and int(np.bincount(keys >> np.uint32(shift), minlength=1 << 14).max()) <= capi.LOCAL_SORT_MAX_KEYS
    assert hybrid_sorts(ctx) - h0 == (2 if fits else 0) + 1, (dist, shift, hybrid_sorts(ctx) - h0)  # + the pairs sort


def test_a_refused_reserving_sort_enqueued_blind_makes_no_claim_about_the_status_words(gpu_context):
    """Found by the fuzz (seed 7171): an enqueue-only sort of 24-bit keys with reserving MSD passes is refused by its plan (range
    too narrow) and runs its four LSD look-back passes from one_read_complete -- they write the status words, so the context
    must not go on believing them clear: the look-back sort that follows (pairs) has to clear them first."""
    ctx = gpu_context
    n = 5000003
    keys = make_keys(n, "uniform", seed=3) & np.uint32(0xFF00FF)
    m = 3000001
    pk, pv = make_keys(m, "uniform", seed=4), np.arange(m, dtype=np.uint32)
    ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 1 << 22)
    ctx.setTuning(capi.VRS_TUNE_MSD_RESERVE, 2)
    try:
        for _ in range(2):
            ctx.setTuning(capi.VRS_TUNE_ASYNC_SORT, 1)
            out, _ = sort_keys(ctx, keys)
            ctx.setTuning(capi.VRS_TUNE_ASYNC_SORT, 0)
            assert np.array_equal(out, np.sort(keys))
            ctx.setTuning(capi.VRS_TUNE_RANK_MODE, 1)  # ballot ranking: no hybrid form, four look-back passes with payloads
            ok, ov = sort_pairs_once(ctx, pk, pv)
            ctx.setTuning(capi.VRS_TUNE_RANK_MODE, 0)
            order = np.argsort(pk, kind="stable")
            assert np.array_equal(ok, pk[order]) and np.array_equal(ov, pv[order])
    finally:
        ctx.setTuning(capi.VRS_TUNE_ASYNC_SORT, 1)  # (the default of a context with its own stream)
        ctx.setTuning(capi.VRS_TUNE_RANK_MODE, 0)
        ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, capi.HYBRID_MIN_KEYS_DEFAULT)
        ctx.setTuning(capi.VRS_TUNE_MSD_RESERVE, 1)


@pytest.mark.parametrize("mode", [0, 2])
def test_hybrid_form_u64_with_and_without_reservation(gpu_context, mode):
    ctx, lib = gpu_context, gpu_context.lib
    n = 6000001
    keys = make_keys64(n, "uniform", 77)
    ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 1 << 22)
    ctx.setTuning(capi.VRS_TUNE_HYBRID, 1)
    ctx.setTuning(capi.VRS_TUNE_MSD_RESERVE, mode)
    h0 = hybrid_sorts(ctx)
    k0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(8 * n), keys)
    k1 = vrs.Buffer(ctx, S(8 * n))
    try:
        for _ in range(2):
            ctx.check(lib.vrs_buffer_upload(ctx.handle, k0.handle, keys.ctypes.data_as(ctypes.c_void_p), keys.nbytes))
            ctx.check(lib.vrs_sort_keys_u64(ctx.handle, k0.handle, k1.handle, n))
            out = np.empty(n, np.uint64)
            k0.downloadWithStagingBuffer(out)
            assert np.array_equal(out, np.sort(keys))
    finally:
        ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, capi.HYBRID_MIN_KEYS_DEFAULT)
        ctx.setTuning(capi.VRS_TUNE_MSD_RESERVE, 1)
        k0.release()
        k1.release()
    assert hybrid_sorts(ctx) - h0 == 2


def sort_pairs_once(ctx, keys, vals):
    lib, n = ctx.lib, keys.size
    k0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), keys)
    v0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), vals)
    k1, v1 = vrs.Buffer(ctx, S(4 * n)), vrs.Buffer(ctx, S(4 * n))
    try:
        ctx.check(lib.vrs_sort_pairs_u32(ctx.handle, k0.handle, k1.handle, v0.handle, v1.handle, n))
        ok, ov = np.empty(n, np.uint32), np.empty(n, np.uint32)
        k0.downloadWithStagingBuffer(ok)
        v0.downloadWithStagingBuffer(ov)
    finally:
        for b in (k0, k1, v0, v1):
            b.release()
    return ok, ov


@pytest.mark.parametrize("dist", ["uniform", "sorted", "low18_const", "28bit", "ties", "one_hot_bucket", "24bit"])
@pytest.mark.parametrize("n", [(1 << 22) + 77, 7000003])
def test_hybrid_form_pairs_are_stable(gpu_context, oracle, n, dist):
    """Key + payload pairs through the hybrid form: both partition passes and the two local passes are stable, so payloads
    of equal keys must come out in input order (the reference's LSD passes guarantee exactly that,
    multi_radixsort.comp:130-141); distributions that cannot take the form fall back to the LSD passes, payloads and all."""
    ctx = gpu_context
    if dist == "28bit":
        keys = make_keys(n, "uniform", seed=5) >> np.uint32(4)
    elif dist == "ties":  # 2^16 distinct keys spread over the whole range: long runs of equal keys inside every bucket
        keys = (make_keys(n, "uniform", seed=6) & np.uint32(0xFFFF)) * np.uint32(65537)
    else:
        keys = make_hybrid_keys(n, dist, seed=n % 311)
    vals = make_keys(n, "uniform", seed=8)
    ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 1 << 22)
    h0 = hybrid_sorts(ctx)
    try:
        ok, ov = sort_pairs_once(ctx, keys, vals)
    finally:
        ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, capi.HYBRID_MIN_KEYS_DEFAULT)
    took = hybrid_sorts(ctx) - h0
    rk, rv, _ = oracle.stable_sort_pairs(keys, vals)
    assert np.array_equal(ok, rk) and np.array_equal(ov, rv)
    shift = int(keys.max()).bit_length() - 14
    fits = 13 <= shift <= 18 and int(np.bincount(keys >> np.uint32(shift), minlength=1 << 14).max()) <= capi.LOCAL_SORT_MAX_PAIRS
    assert took == (1 if fits else 0), (dist, shift)


@pytest.mark.parametrize("mode", ["keys", "pairs", "u64", "pairs_too_large"])
def test_hybrid_form_with_buckets_beyond_the_small_local_sort(gpu_context, mode):
    """Buckets of 7166-14333 keys take the 512-thread local sort, buckets of 6657-13312 pairs or 64-bit keys the 1024-thread
    one (what uniform inputs of 10^8 to 2 * 10^8 elements have).  Three of four top-14-bit buckets are empty here, so 3.6e7
    keys fill the others with about 8800 each; 5.6e7 pairs fill them with 13700: refused, the LSD passes run."""
    ctx, lib = gpu_context, gpu_context.lib
    n = 56000001 if mode == "pairs_too_large" else 36000001
    keys = make_keys(n, "uniform", seed=23) & np.uint32(0xFFF3FFFF)
    largest = int(np.bincount(keys >> np.uint32(18), minlength=1 << 14).max())
    if mode == "keys":
        assert capi.LOCAL_SORT_SMALL_KEYS < largest <= capi.LOCAL_SORT_MAX_KEYS
    elif mode == "pairs_too_large":
        assert largest > capi.LOCAL_SORT_MAX_PAIRS
    else:
        assert capi.LOCAL_SORT_SMALL_PAIRS < largest <= capi.LOCAL_SORT_MAX_PAIRS
    ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 1 << 22)
    ctx.setTuning(capi.VRS_TUNE_HYBRID, 1)  # forgets an earlier refusal of a 64-bit sort
    ctx.setTuning(capi.VRS_TUNE_MSD_POOL, 0)  # the COUNTED form is the subject (the pool form picks its local sort from n alone: it would refuse these buckets)
    ctx.setTuning(capi.VRS_TUNE_ASYNC_SORT, 0)  # ... in its blocking form, where the PLAN picks the local sort's shape (enqueued blind, N alone does)
    h0 = hybrid_sorts(ctx)
    try:
        if mode == "keys":
            out, stats = sort_keys(ctx, keys)
            assert np.array_equal(out, np.sort(keys))
            assert stats["local_sort"] == 1 and stats["lookback_scatter"] == 2
        elif mode == "u64":
            low = make_keys(n, "uniform", seed=24).astype(np.uint64)
            keys64 = (keys.astype(np.uint64) << np.uint64(32)) | low
            k0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(8 * n), keys64)
            k1 = vrs.Buffer(ctx, S(8 * n))
            try:
                ctx.check(lib.vrs_sort_keys_u64(ctx.handle, k0.handle, k1.handle, n))
                out = np.empty(n, np.uint64)
                k0.downloadWithStagingBuffer(out)
            finally:
                k0.release()
                k1.release()
            assert np.array_equal(out, np.sort(keys64))
        else:
            vals = np.arange(n, dtype=np.uint32)
            ok, ov = sort_pairs_once(ctx, keys, vals)
            order = np.argsort(keys, kind="stable")
            assert np.array_equal(ok, keys[order]) and np.array_equal(ov, vals[order])
    finally:
        ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, capi.HYBRID_MIN_KEYS_DEFAULT)
        ctx.setTuning(capi.VRS_TUNE_MSD_POOL, 1)
        ctx.setTuning(capi.VRS_TUNE_ASYNC_SORT, 1)
    assert hybrid_sorts(ctx) - h0 == (0 if mode == "pairs_too_large" else 1)


def make_hybrid_keys64(n, dist, seed):
    if dist == "narrow":  # 20-bit keys: the range is too narrow for the form, the LSD passes drop their identity passes
        return make_keys64(n, "uniform", seed) >> np.uint64(44)
    if dist == "one_hot_bucket":  # a bucket no workgroup can hold
        k = make_keys64(n, "uniform", seed)
        k[: n // 20] = (k[: n // 20] & np.uint64((1 << 50) - 1)) | np.uint64(0x1234 << 50)
        return k
    if dist == "low14_only":  # inside a bucket the keys differ ONLY in their low 14 bits: the top-four-digits shortcut of the
        k = make_keys64(n, "uniform", seed)  # local sort finds them out of order and all six passes run
        return (k & np.uint64(0xFFFC000000003FFF)) | np.uint64(0x0000123456784000)
    if dist == "ties":  # 2^12 distinct keys spread over 64 bits: one value per bucket, six passes over equal keys
        return (make_keys64(n, "uniform", seed) >> np.uint64(52)) * np.uint64(0x0010000100001001)
    return make_keys64(n, dist, seed)


@pytest.mark.parametrize("dist", ["uniform", "44bit", "low32", "high32", "sorted", "max_keys", "narrow", "one_hot_bucket", "ties", "low14_only"])
@pytest.mark.parametrize("n", [(1 << 22) + 99, 6000001])
def test_hybrid_form_u64(gpu_context, n, dist):
    """64-bit keys through the hybrid form: buckets = the top 14 bits of the key range, the local sort takes the remaining
    low bits in ceil(bits / 9) LDS passes (44-bit keys: four, full 64-bit keys: six, upper word zero: two).  Ranges below 27
    bits and buckets that do not fit are refused -- nothing has moved by then -- and the eight LSD passes run."""
    ctx, lib = gpu_context, gpu_context.lib
    keys = make_hybrid_keys64(n, dist, seed=n % 211)
    ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 1 << 22)
    ctx.setTuning(capi.VRS_TUNE_HYBRID, 1)  # also forgets an earlier refusal (after one, only every 16th 64-bit sort tries again)
    h0 = hybrid_sorts(ctx)
    k0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(8 * n), keys)
    k1 = vrs.Buffer(ctx, S(8 * n))
    try:
        ctx.check(lib.vrs_sort_keys_u64(ctx.handle, k0.handle, k1.handle, n))
        out = np.empty(n, np.uint64)
        k0.downloadWithStagingBuffer(out)
    finally:
        ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, capi.HYBRID_MIN_KEYS_DEFAULT)
        k0.release()
        k1.release()
    assert np.array_equal(out, np.sort(keys))
    bits = int(keys.max()).bit_length()
    shift = bits - 14
    fits = shift >= 13 and int(np.bincount((keys >> np.uint64(shift)).astype(np.int64), minlength=1 << 14).max()) <= capi.LOCAL_SORT_MAX_PAIRS
    assert hybrid_sorts(ctx) - h0 == (1 if fits else 0), (dist, bits)


# ---- the one-call sorts as enqueue-only calls (VRS_TUNE_ASYNC_SORT) and the bounded wait for the plan

def _busy_stream(ms):
    """Put about `ms` milliseconds of work on torch's current stream; returns at once."""
    import torch
    torch.cuda._sleep(int(ms * 1e-3 * 2.0e9))  # cycles of the device's ~2 GHz clock


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["hybrid", "hybrid_refused", "lsd", "lsd_abnormal", "pairs_hybrid", "u64_lsd"])
def test_async_sort_returns_at_once_and_settles(case):
    """VRS_TUNE_ASYNC_SORT = 1 on a borrowed stream that is busy for ~60 ms: vrs_sort_* must return in well under a
    millisecond (the reference's execute() never blocks, ComputePass.h:31-56), vrs_sort_pending says whether a second half is
    owed, and after vrs_sort_settle + a stream sync the result is bit-exact -- for a sort the hybrid form takes (enqueued
    completely: nothing owed beyond the bookkeeping), one it refuses (the settle runs the LSD sort), plain LSD sorts with and
    without abnormal passes, pairs and 64-bit keys."""
    import time
    import torch
    dev = torch.device("cuda", 0)
    n = {"hybrid": (1 << 23) + 77, "hybrid_refused": (1 << 23) + 77, "lsd": 3000001, "lsd_abnormal": 3000001,
         "pairs_hybrid": (1 << 23) + 5, "u64_lsd": 2000003}[case]
    rs = np.random.RandomState(len(case))
    wide = case == "u64_lsd"
    if wide:
        keys = rs.randint(0, 2 ** 63, size=n, dtype=np.int64).astype(np.uint64)
    else:
        keys = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32)
    if case == "hybrid_refused":
        keys[: n // 8] = (keys[: n // 8] & np.uint32(0x3FFFF)) | np.uint32(0x12340000 & ~0x3FFFF)  # one bucket far too large
    if case == "lsd_abnormal":
        keys &= np.uint32(0x00FFFF00)  # two identity passes, and low byte constant: streams of pass 1 collapse
    with vrs.GPUContext(0, stream=torch.cuda.current_stream().cuda_stream) as gpu:
        gpu.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 1 << 22)
        kt = torch.from_numpy(keys.view(np.int64 if wide else np.int32)).to(dev)
        tmp = torch.empty_like(kt)
        vals = torch.arange(n, dtype=torch.int32, device=dev) if case == "pairs_hybrid" else None
        vtmp = torch.empty_like(vals) if vals is not None else None
        eb = 8 if wide else 4
        k0 = vrs.Buffer(gpu, S(eb * n), device_ptr=kt.data_ptr())
        k1 = vrs.Buffer(gpu, S(eb * n), device_ptr=tmp.data_ptr())
        v0 = vrs.Buffer(gpu, S(4 * n), device_ptr=vals.data_ptr()) if vals is not None else None
        v1 = vrs.Buffer(gpu, S(4 * n), device_ptr=vtmp.data_ptr()) if vals is not None else None

        def call():
            if wide:
                return gpu.lib.vrs_sort_keys_u64(gpu.handle, k0.handle, k1.handle, n)
            if vals is not None:
                return gpu.lib.vrs_sort_pairs_u32(gpu.handle, k0.handle, k1.handle, v0.handle, v1.handle, n)
            return gpu.lib.vrs_sort_keys_u32(gpu.handle, k0.handle, k1.handle, n)

        # warm-up in the default mode: scratch allocations, module load
        pristine = kt.clone()
        gpu.check(call())
        torch.cuda.synchronize()
        kt.copy_(pristine)
        if vals is not None:
            vals.copy_(torch.arange(n, dtype=torch.int32, device=dev))
        torch.cuda.synchronize()
        gpu.setTuning(capi.VRS_TUNE_ASYNC_SORT, 1)
        h0 = hybrid_sorts(gpu)
        _busy_stream(60)
        t0 = time.perf_counter()
        rc = call()
        dt = time.perf_counter() - t0
        gpu.check(rc)
        assert dt < 2e-3, f"the enqueue-only sort took {dt * 1e3:.2f} ms on the host"
        assert gpu.lib.vrs_sort_pending(gpu.handle) == 1
        assert not torch.cuda.current_stream().query()  # the stream is still busy with the work queued before the sort
        gpu.check(gpu.lib.vrs_sort_settle(gpu.handle))
        assert gpu.lib.vrs_sort_pending(gpu.handle) == 0
        torch.cuda.synchronize()
        took = hybrid_sorts(gpu) - h0
        gpu.setTuning(capi.VRS_TUNE_ASYNC_SORT, 0)
        out = kt.cpu().numpy().view(keys.dtype)
        if vals is not None:
            order = np.argsort(keys, kind="stable")
            assert np.array_equal(out, keys[order]) and np.array_equal(vals.cpu().numpy().view(np.uint32), order.astype(np.uint32))
        else:
            assert np.array_equal(out, np.sort(keys))
        assert took == (1 if case in ("hybrid", "pairs_hybrid") else 0)
        for b in (k0, k1, v0, v1):
            if b is not None:
                b.release()


@pytest.mark.gpu
def test_async_sorts_back_to_back_and_implicit_settle():
    """Several enqueue-only sorts in a row (each call settles its predecessor first) and an entry point that settles by
    itself (the blocking download) instead of an explicit vrs_sort_settle."""
    with vrs.GPUContext(0) as gpu:
        gpu.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 1 << 22)
        gpu.setTuning(capi.VRS_TUNE_ASYNC_SORT, 1)
        bufs = []
        for i, n in enumerate([(1 << 22) + 3, 1500001, (1 << 23) + 9, 70001]):
            keys = make_keys(n, ["uniform", "16bit", "uniform", "mult256"][i], seed=40 + i)
            k0 = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), keys)
            k1 = vrs.Buffer(gpu, S(4 * n))
            gpu.check(gpu.lib.vrs_sort_keys_u32(gpu.handle, k0.handle, k1.handle, n))
            bufs.append((keys, k0, k1))
        for keys, k0, k1 in bufs:
            out = np.empty(keys.size, np.uint32)
            k0.downloadWithStagingBuffer(out)  # settles the last sort, then waits for the stream
            assert np.array_equal(out, np.sort(keys))
            k0.release()
            k1.release()
        assert gpu.lib.vrs_sort_pending(gpu.handle) == 0


@pytest.mark.gpu
def test_the_wait_for_the_plan_is_bounded():
    """Default (blocking) mode behind a stream that stays busy longer than VRS_TUNE_PLAN_WAIT_MS: the call gives up with
    VRS_ERROR_TIMEOUT instead of spinning on; the sort is still queued and a later vrs_sort_settle completes it."""
    import torch
    dev = torch.device("cuda", 0)
    n = 3000001
    keys = make_keys(n, "uniform", seed=3)
    with vrs.GPUContext(0, stream=torch.cuda.current_stream().cuda_stream) as gpu:
        kt = torch.from_numpy(keys.view(np.int32)).to(dev)
        tmp = torch.empty_like(kt)
        k0 = vrs.Buffer(gpu, S(4 * n), device_ptr=kt.data_ptr())
        k1 = vrs.Buffer(gpu, S(4 * n), device_ptr=tmp.data_ptr())
        pristine = kt.clone()
        gpu.check(gpu.lib.vrs_sort_keys_u32(gpu.handle, k0.handle, k1.handle, n))
        torch.cuda.synchronize()
        kt.copy_(pristine)
        torch.cuda.synchronize()
        gpu.setTuning(capi.VRS_TUNE_PLAN_WAIT_MS, 5)
        _busy_stream(300)
        rc = gpu.lib.vrs_sort_keys_u32(gpu.handle, k0.handle, k1.handle, n)
        assert rc == capi.VRS_ERROR_TIMEOUT
        assert gpu.lib.vrs_sort_pending(gpu.handle) == 1
        gpu.setTuning(capi.VRS_TUNE_PLAN_WAIT_MS, 60000)
        gpu.check(gpu.lib.vrs_sort_settle(gpu.handle))
        torch.cuda.synchronize()
        assert np.array_equal(kt.cpu().numpy().view(np.uint32), np.sort(keys))
        k0.release()
        k1.release()


@pytest.mark.gpu
@pytest.mark.parametrize("reserve", [1, 2])
@pytest.mark.parametrize("span_bits", [27, 29])
def test_ranged_sort_buckets_a_sub_range_of_the_key_space(gpu_context, span_bits, reserve):
    """vrs_sort_keys_u32_ranged: keys of a sub-range [floor, floor + 2^span) -- what a rank of the multi-GPU step receives --
    are bucketed by key - floor, so the hybrid form sees the 16384 evenly filled buckets a full key range would give; a key
    below the promised floor only costs the hybrid form (the plan refuses), never the result."""
    ctx, lib = gpu_context, gpu_context.lib
    n = (1 << 23) + 3
    floor_key = 0xA3000000 if span_bits == 27 else 0x60000000
    rs = np.random.RandomState(span_bits)
    keys = (np.uint32(floor_key) + rs.randint(0, 1 << span_bits, size=n, dtype=np.uint32)).astype(np.uint32)
    ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 1 << 22)
    ctx.setTuning(capi.VRS_TUNE_MSD_RESERVE, reserve)  # 2: the MSD passes reserve at this size too (ragged tiles pad with floor - 1)
    try:
        for stray in (False, True):
            k = keys.copy()
            if stray:
                k[n // 3] = np.uint32(floor_key - 5)
            k0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), k)
            k1 = vrs.Buffer(ctx, S(4 * n))
            h0 = hybrid_sorts(ctx)
            ctx.check(lib.vrs_sort_keys_u32_ranged(ctx.handle, k0.handle, k1.handle, n, floor_key + 12345))  # rounded down to 2^24
            out = np.empty(n, np.uint32)
            k0.downloadWithStagingBuffer(out)
            assert np.array_equal(out, np.sort(k))
            assert hybrid_sorts(ctx) - h0 == (0 if stray else 1)
            k0.release()
            k1.release()
    finally:
        ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, capi.HYBRID_MIN_KEYS_DEFAULT)
        ctx.setTuning(capi.VRS_TUNE_MSD_RESERVE, 1)


@pytest.mark.parametrize("n,kind", [(3000, "keys"), (5000, "keys"), (100000, "keys"), ((1 << 22) + 11, "keys"), (14000000, "keys"),
                                    (9000, "pairs"), (300000, "u64"), (26000000, "pairs"), (21000000, "u64")])
def test_the_form_that_runs_is_the_form_the_decision_function_names(oracle, n, kind):
    """vrs_sort_form_for (host only; tests/test_capi_cpu.py walks its whole table) against what a fresh context really launches: the
    dispatcher asks the same function, so the kernels that run must be those of the form it names -- and the result std::sort's."""
    import ctypes
    rs = np.random.RandomState(n % 1009)
    wide, pairs = kind == "u64", kind == "pairs"
    keys = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32)
    if wide:
        keys = (keys.astype(np.uint64) << np.uint64(32)) | rs.randint(0, 2 ** 32, size=n, dtype=np.uint64)
    form = ctypes.c_int()
    with vrs.GPUContext(0) as ctx:  # a fresh context: the defaults vrs_sort_form_for assumes
        assert ctx.lib.vrs_sort_form_for(n, 8 if wide else 4, int(pairs), None, 0, ctypes.byref(form), None) == 0
        name = capi.FORM_NAMES[form.value]
        S = vrs.Buffer.BufferSettings
        k0, k1 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(keys.nbytes), keys), vrs.Buffer(ctx, S(keys.nbytes))
        v0 = v1 = None
        if pairs:
            v0, v1 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), np.arange(n, dtype=np.uint32)), vrs.Buffer(ctx, S(4 * n))
        ctx.profileReset()
        ctx.profileEnable(True)
        if pairs:
            ctx.check(ctx.lib.vrs_sort_pairs_u32(ctx.handle, k0.handle, k1.handle, v0.handle, v1.handle, n))
        elif wide:
            ctx.check(ctx.lib.vrs_sort_keys_u64(ctx.handle, k0.handle, k1.handle, n))
        else:
            ctx.check(ctx.lib.vrs_sort_keys_u32(ctx.handle, k0.handle, k1.handle, n))
        out = np.empty_like(keys)
        k0.downloadWithStagingBuffer(out)
        ran = {nm: launches(ctx, kid) for kid, nm in capi.KERNEL_NAMES.items()}
        ctx.profileEnable(False)
        for b in (k0, k1, v0, v1):
            if b is not None:
                b.release()
    assert np.array_equal(out, np.sort(keys))
    expect = {"single": ran["single"] == 1 and ran["histogram"] == 0 and ran["digit_tables"] == 0,
              "contract": ran["histogram"] == keys.dtype.itemsize and ran["scatter"] == keys.dtype.itemsize and ran["digit_tables"] == 0,
              "lsd": ran["digit_tables"] >= 1 and ran["local_sort"] == 0 and ran["pool_pass_a"] == 0 and ran["histogram"] == 0,
              "counted": ran["digit_tables"] == 1 and ran["local_sort"] >= 1 and ran["pool_pass_a"] == 0,
              "pool": ran["pool_pass_a"] == 1 and ran["pool_pass_b"] == 1 and ran["local_sort"] >= 1 and ran["digit_tables"] == 0}
    assert expect[name], (name, ran)
    assert name == {3000: "single", 5000: "contract", 100000: "lsd", (1 << 22) + 11: "pool", 14000000: "pool", 9000: "lsd", 300000: "lsd",
                    26000000: "pool", 21000000: "counted"}[n]
