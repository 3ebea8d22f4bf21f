"""GPU parity tests of the pool form of vrs_sort_keys_u32 (vrs_msd_pool.hip): the hybrid form WITHOUT a counting read --
a sample sizes a region per (input slice, top byte), the first MSD pass reserves its output there, the second scatters into
per-bucket regions of a slack buffer (sized from a sample of the first pass's output), the local sort reads every bucket in one
piece and writes it to its final place (24 instead of 28 bytes per key).  The reference counts the keys once
per pass (multi_radixsort_histograms.comp:42-50); here they are not counted at all, and the acceptance criterion stays the
reference's own: the output equals std::sort, bit for bit (MultiRadixSort.cpp:141-161).  Whatever the sample misjudges -- a
region, the key range, a bucket's size -- must end in a refusal and a counted sort of the untouched input, never in a wrong
result."""
import ctypes

import numpy as np
import pytest

import vkradixsort_amd as vrs
from vkradixsort_amd import capi

from .test_gpu_one_call import launches, make_keys

pytestmark = pytest.mark.gpu
S = vrs.Buffer.BufferSettings
POOL_MIN = 1 << 22  # the form's own floor (the default threshold is 3.2e7 keys: below it the counted form is the faster one)


def pool_counts(ctx):
    a, b = ctypes.c_uint64(), ctypes.c_uint64()
    ctx.check(ctx.lib.vrs_one_call_pool_sorts(ctx.handle, ctypes.byref(a), ctypes.byref(b)))
    return a.value, b.value


def pool_keys(n, dist, seed):
    rs = np.random.RandomState(seed)
    if dist == "gauss":  # top bytes far from uniform, but the same everywhere in the input: the sample sizes every region right
        return np.clip(rs.normal(2.0 ** 31, 2.0 ** 28, size=n), 0, 2.0 ** 32 - 1).astype(np.uint32)
    if dist == "halves":  # the input's first half holds small keys, the second large ones: the slices' regions differ
        k = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32)
        k[: n // 2] >>= np.uint32(1)
        k[n // 2:] |= np.uint32(0x80000000)
        return k
    if dist == "tile_period":  # the sampled head of every 8192-key tile is unlike the rest of it: every region is misjudged
        k = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32)
        t = np.arange(n) % 8192
        return np.where(t < 256, k >> np.uint32(1), k | np.uint32(0x80000000)).astype(np.uint32)
    if dist == "dups":  # 256 distinct low parts per bucket: many ties inside every bucket
        k = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32)
        return (k & np.uint32(0xFFFC0000)) | (k & np.uint32(0xFF))
    if dist == "hot_bucket":  # one bucket with 0.33 % of the keys: above the local sort's capacity, verdict 2 must say no
        k = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32)
        k[: n // 300] = (k[: n // 300] & np.uint32(0x3FFFF)) | np.uint32(0x12340000)
        return k
    if dist == "rare_high_bit":  # 28-bit keys except three that use bit 31: the probe misses them, the second pass must not
        k = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32) >> np.uint32(4)
        k[[5, n // 2 + 1, n - 2]] |= np.uint32(0x80000000)
        return k
    if dist == "24bit":
        return rs.randint(0, 2 ** 32, size=n, dtype=np.uint32) >> np.uint32(8)
    return make_keys(n, dist, seed)


@pytest.fixture
def pool_ctx(gpu_context):
    """the shared context with the pool form forced at test sizes; everything restored afterwards"""
    ctx = gpu_context
    ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, POOL_MIN)
    ctx.setTuning(capi.VRS_TUNE_MSD_POOL_MIN_KEYS, POOL_MIN)
    ctx.setTuning(capi.VRS_TUNE_MSD_POOL, 2)
    yield ctx
    ctx.setTuning(capi.VRS_TUNE_MSD_POOL, 1)
    ctx.setTuning(capi.VRS_TUNE_MSD_POOL_MIN_KEYS, POOL_MIN)  # (the default since round 5)
    ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 0)
    ctx.setTuning(capi.VRS_TUNE_DEBUG_MISPLACE_STREAMS, 0)
    ctx.setTuning(capi.VRS_TUNE_MSD_POOL_SUB_BITS, 0)
    ctx.setTuning(capi.VRS_TUNE_ASYNC_SORT, 1)  # (the default of a context with its own stream)


def sort_and_stats(ctx, keys, key_floor=None):
    n = keys.size
    k0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), keys)
    k1 = vrs.Buffer(ctx, S(4 * n))
    ctx.profileReset()
    ctx.profileEnable(True)
    before = pool_counts(ctx)
    try:
        if key_floor is None:
            ctx.check(ctx.lib.vrs_sort_keys_u32(ctx.handle, k0.handle, k1.handle, n))
        else:
            ctx.check(ctx.lib.vrs_sort_keys_u32_ranged(ctx.handle, k0.handle, k1.handle, n, key_floor))
        out = np.empty(n, np.uint32)
        k0.downloadWithStagingBuffer(out)
        stats = {name: launches(ctx, kid) for kid, name in capi.KERNEL_NAMES.items()}
    finally:
        ctx.profileEnable(False)
        k0.release()
        k1.release()
    after = pool_counts(ctx)
    return out, stats, (after[0] - before[0], after[1] - before[1])


TAKEN = ["uniform", "28bit", "gauss", "halves", "dups"]
REFUSED = ["24bit", "const", "tile_period", "hot_bucket", "rare_high_bit", "two_values", "max_keys"]


@pytest.mark.parametrize("dist", TAKEN)
@pytest.mark.parametrize("n", [POOL_MIN + 5, 9000001, 30000001])
def test_pool_form_equals_std_sort(pool_ctx, oracle, n, dist):
    keys = pool_keys(n, dist, seed=n % 997)
    out, stats, (took, refused) = sort_and_stats(pool_ctx, keys)
    assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1
    retries = ctypes.c_uint64()
    pool_ctx.check(pool_ctx.lib.vrs_one_call_pool_retries(pool_ctx.handle, ctypes.byref(retries)))
    if dist == "gauss":
        # the fullest top byte holds 2.5 % of the keys: its buckets are six times the mean, above the local sort the form picked
        # from n alone -- no refusal: they lie whole in their slack regions, and the settle enqueues a larger local sort
        # (at 4.2e6 keys the fullest bucket -- 1640 keys -- still fits the one-wave local sort)
        assert (took, refused) == (1, 0) and stats["local_sort"] == (2 if n > 5000000 else 1) and (retries.value >= 1 or n < 5000000)
        return
    # the form really ran, and nothing was counted ahead: a sample, its two passes, the local sort
    assert (took, refused) == (1, 0)
    assert stats["pool_sample"] == 1 and stats["digit_tables"] == 0 and stats["lookback_scatter"] == 0 and stats["local_sort"] == 1
    assert stats["pool_pass_a"] == 1 and stats["pool_pass_b"] == 1


@pytest.mark.parametrize("dist", REFUSED)
def test_pool_form_refuses_and_the_counted_form_sorts_the_untouched_input(pool_ctx, oracle, dist):
    n = 12000003
    keys = pool_keys(n, dist, seed=11)
    out, stats, (took, refused) = sort_and_stats(pool_ctx, keys)
    assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1
    assert (took, refused) == (0, 1)
    assert stats["digit_tables"] >= 1  # the counted form (hybrid or LSD) ran after the refusal


@pytest.mark.parametrize("dist", ["sorted", "reverse", "clustered"])
def test_pool_form_on_ordered_input_is_exact_whatever_the_verdict(pool_ctx, oracle, dist):
    """sorted input: every slice holds 32 of the 256 top bytes -- the regions follow (the sample sees every tile); whether a
    verdict refuses is the form's business, the result is not"""
    n = 16000001
    keys = pool_keys(n, dist, seed=5)
    out, _, (took, refused) = sort_and_stats(pool_ctx, keys)
    assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1
    assert took + refused == 1


def test_pool_form_with_workgroups_off_their_slices(pool_ctx, oracle):
    """test hook: odd tiles of the first pass are read from the neighbouring slice (the row of cursors a workgroup adds to
    follows the XCC it runs on, not the slice it reads)"""
    n = 20000003
    keys = pool_keys(n, "uniform", seed=3)
    pool_ctx.setTuning(capi.VRS_TUNE_DEBUG_MISPLACE_STREAMS, 1)
    out, _, (took, refused) = sort_and_stats(pool_ctx, keys)
    assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1
    assert took + refused == 1
    # slices with different key ranges: the neighbour's keys do not fit the regions the sample sized -- refused, still exact
    keys = pool_keys(n, "halves", seed=4)
    out, _, (took, refused) = sort_and_stats(pool_ctx, keys)
    assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1
    assert took + refused == 1


@pytest.mark.parametrize("sub_bits", [6, 7])
@pytest.mark.parametrize("dist", ["uniform", "28bit", "halves", "dups"])
def test_pool_form_with_either_width_of_the_second_pass(pool_ctx, oracle, sub_bits, dist):
    """VRS_TUNE_MSD_POOL_SUB_BITS: 16384 buckets of 18 low bits or 32768 of 17 (the default follows the size: 7 bits where 6 would
    need the 512-thread local sort)"""
    n = 11000017
    pool_ctx.setTuning(capi.VRS_TUNE_MSD_POOL_SUB_BITS, sub_bits)
    keys = pool_keys(n, dist, seed=sub_bits)
    out, stats, (took, refused) = sort_and_stats(pool_ctx, keys)
    assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1
    assert (took, refused) == (1, 0) and stats["pool_pass_b"] == 1 and stats["digit_tables"] == 0


@pytest.mark.parametrize("rotate", [1, 3, 7])
def test_pool_form_does_not_depend_on_where_the_round_robin_starts(oracle, rotate):
    """The dispatcher deals the blocks of a launch out to the XCCs round-robin, starting at an XCC of the hardware queue's own -- and a
    stream may move to another queue after the context probed the placement (seen in round 5: a probe that said 0 1 2 .. 7, sorts
    that ran 7 0 1 .. 6).  Both passes of the pool form take their work lists by the XCC they RUN on: with the probed order
    rotated (test hook) the form is still taken, every tile is claimed exactly once, and the result is exact.  The counted form
    and the look-back passes fall back to their placement-independent routes: exact too."""
    with vrs.GPUContext(0) as gpu:
        gpu.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, POOL_MIN)
        gpu.setTuning(capi.VRS_TUNE_MSD_POOL_MIN_KEYS, POOL_MIN)
        gpu.setTuning(capi.VRS_TUNE_MSD_POOL, 2)
        gpu.setTuning(capi.VRS_TUNE_DEBUG_XCC_ROTATE, rotate)
        for dist in ("uniform", "halves"):
            keys = pool_keys(9000011, dist, seed=rotate)
            out, stats, (took, refused) = sort_and_stats(gpu, keys)
            assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1
            assert (took, refused) == (1, 0), dist
        gpu.setTuning(capi.VRS_TUNE_MSD_POOL, 0)  # the counted form, then (below its threshold) the look-back passes
        for n in (9000011, 3000001):
            keys = pool_keys(n, "uniform", seed=n % 97)
            out, _, _ = sort_and_stats(gpu, keys)
            assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1


def test_pool_form_beyond_the_small_local_sort(gpu_context):
    """1.3e8 uniform keys: 16384 buckets would need the 512-thread local sort, so the second pass takes seven bits (32768 buckets of
    about 4000 keys) -- round 4's in-place second pass stopped at 1.15e8 keys (56 tiles per top byte)"""
    n = 130000003
    keys = np.random.RandomState(4).randint(0, 2 ** 32, size=n, dtype=np.uint32)
    gpu_context.setTuning(capi.VRS_TUNE_MSD_POOL, 1)
    out, stats, (took, refused) = sort_and_stats(gpu_context, keys)
    assert (took, refused) == (1, 0) and stats["digit_tables"] == 0 and stats["pool_pass_b"] == 1
    assert np.array_equal(out, np.sort(keys))


def pool_layouts(ctx):
    a, b = ctypes.c_uint64(), ctypes.c_uint64()
    ctx.check(ctx.lib.vrs_one_call_pool_layouts(ctx.handle, ctypes.byref(a), ctypes.byref(b)))
    return a.value, b.value


def test_a_kept_layout_serves_the_next_sort_of_its_size_and_is_dropped_when_it_does_not_fit(oracle):
    """VRS_TUNE_MSD_POOL_REUSE_LAYOUT (the default): the second sort of a size runs its first pass in the regions the first one's sample
    laid out -- no sample kernel --; other keys of the same distribution fit; keys of ANOTHER distribution (here: another key range, and
    slices that differ) do not: that sort is flagged like any sort whose regions overflow, runs again with a sample of its own, and is
    exact; a size change samples; switched off, every sort samples."""
    n = 6000011
    with vrs.GPUContext(0) as gpu:
        gpu.setTuning(capi.VRS_TUNE_MSD_POOL, 2)
        first = pool_keys(n, "uniform", seed=1)
        out, stats, (took, refused) = sort_and_stats(gpu, first)
        assert oracle.test_sort(oracle.std_sort(first)[0], out) == -1 and (took, refused) == (1, 0) and stats["pool_sample"] == 1
        assert pool_layouts(gpu) == (0, 0)
        for seed in (2, 3):  # other uniform keys: the kept regions fit
            keys = pool_keys(n, "uniform", seed=seed)
            out, stats, (took, refused) = sort_and_stats(gpu, keys)
            assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1
            assert (took, refused) == (1, 0) and stats["pool_sample"] == 0 and stats["pool_pass_a"] == 1
        assert pool_layouts(gpu) == (2, 0)
        for dist in ("28bit", "halves"):  # another key range / other regions: stale, sampled again, exact, no refusal of the FORM
            keys = pool_keys(n, dist, seed=4)
            out, stats, (took, refused) = sort_and_stats(gpu, keys)
            assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1, dist
            assert (took, refused) == (1, 0) and stats["pool_sample"] == 1 and stats["pool_pass_a"] == 2, (dist, stats)
            out, stats, (took, refused) = sort_and_stats(gpu, keys)  # ... and ITS layout serves the next sort of these keys
            assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1 and stats["pool_sample"] == 0
        reused, stale = pool_layouts(gpu)
        assert (reused, stale) == (6, 2)
        # ... and keys that miss the kept regions by far: the sorted output of a sort (every slice then holds 32 of the 256 top bytes,
        # eight times what their regions take -- the soak's case: the first pass's cursors count what was ASKED for, and the plan
        # kernel's sample must not follow them behind the overflow scratch)
        rnd = pool_keys(n, "uniform", seed=9)
        out, _, _ = sort_and_stats(gpu, rnd)          # (sampled: the layout of uniform keys in random order)
        out2, stats, (took, refused) = sort_and_stats(gpu, out)  # the same keys, sorted: stale
        assert np.array_equal(out2, out) and oracle.test_sort(oracle.std_sort(rnd)[0], out) == -1 and took + refused == 1
        reused2, stale2 = pool_layouts(gpu)
        assert reused2 == reused + 2 and stale2 >= stale + 1  # (both started in a kept layout; the sorted keys surely did not fit theirs)
        other = pool_keys(n + 8192, "uniform", seed=5)  # another size: sampled
        out, stats, _ = sort_and_stats(gpu, other)
        assert oracle.test_sort(oracle.std_sort(other)[0], out) == -1 and stats["pool_sample"] == 1
        gpu.setTuning(capi.VRS_TUNE_MSD_POOL_REUSE_LAYOUT, 0)
        for _ in range(2):
            out, stats, _ = sort_and_stats(gpu, other)
            assert oracle.test_sort(oracle.std_sort(other)[0], out) == -1 and stats["pool_sample"] == 1
        assert pool_layouts(gpu)[0] == reused2 + 0  # (nothing reused while switched off; the sort of `other` before sampled: another size)


def test_pool_form_enqueue_only(pool_ctx, oracle):
    """VRS_TUNE_ASYNC_SORT: the call returns with the whole form enqueued; a refusal is handled by the settle"""
    pool_ctx.setTuning(capi.VRS_TUNE_ASYNC_SORT, 1)
    for dist in ("uniform", "tile_period"):
        keys = pool_keys(10000019, dist, seed=8)
        out, _, (took, refused) = sort_and_stats(pool_ctx, keys)
        assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1
        assert (took, refused) == ((1, 0) if dist == "uniform" else (0, 1))


def test_releasing_a_buffer_of_a_pending_sort_settles_the_sort_first(pool_ctx, oracle):
    """An enqueue-only sort whose first half was refused still owes a whole counted sort -- over the raw pointers of its two
    buffers.  vrs_buffer_release of one of them (callers written against the blocking form release keys_tmp right after the
    call) must run that second half before the memory goes, not hand it freed memory."""
    pool_ctx.setTuning(capi.VRS_TUNE_ASYNC_SORT, 1)
    for dist in ("tile_period", "24bit", "uniform"):
        n = 7000003
        keys = pool_keys(n, dist, seed=13)
        k0 = vrs.Buffer.fillDeviceWithStagingBuffer(pool_ctx, S(4 * n), keys)
        k1 = vrs.Buffer(pool_ctx, S(4 * n))
        pool_ctx.check(pool_ctx.lib.vrs_sort_keys_u32(pool_ctx.handle, k0.handle, k1.handle, n))
        k1.release()  # the sort may not have settled yet
        filler = vrs.Buffer(pool_ctx, S(4 * n))  # (likely the memory k1 just gave back)
        out = np.empty(n, np.uint32)
        k0.downloadWithStagingBuffer(out)
        assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1, dist
        k0.release()
        filler.release()


def test_pool_form_of_a_ranged_sort(pool_ctx, oracle):
    """vrs_sort_keys_u32_ranged: buckets and digits are taken from key - floor; a key below the promised floor refuses"""
    n = 10000007
    rs = np.random.RandomState(21)
    keys = (np.uint32(0x30000000) + rs.randint(0, 1 << 28, size=n, dtype=np.uint32)).astype(np.uint32)
    out, stats, (took, refused) = sort_and_stats(pool_ctx, keys, key_floor=0x30000000)
    assert np.array_equal(out, np.sort(keys)) and (took, refused) == (1, 0) and stats["digit_tables"] == 0
    keys[12345] = 0x2FFFFFFF  # below the floor
    out, _, (took, refused) = sort_and_stats(pool_ctx, keys, key_floor=0x30000000)
    assert np.array_equal(out, np.sort(keys)) and (took, refused) == (0, 1)


def test_pool_and_counted_forms_alternate_on_one_context(pool_ctx, oracle):
    """the forms share the reservation counters, the bucket histogram's memory and the plan: whatever order they run in"""
    ctx = pool_ctx
    for i, (mode, dist) in enumerate([(2, "uniform"), (0, "uniform"), (2, "tile_period"), (2, "gauss"), (0, "28bit"), (2, "28bit"), (2, "hot_bucket"),
                                      (2, "uniform")]):
        ctx.setTuning(capi.VRS_TUNE_MSD_POOL, mode)
        keys = pool_keys(6000011 + 4099 * i, dist, seed=30 + i)
        out, _, _ = sort_and_stats(ctx, keys)
        assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1, (i, mode, dist)


def test_pool_form_adaptive_mode_backs_off_after_a_refusal(pool_ctx, oracle):
    """VRS_TUNE_MSD_POOL = 1 (the default): after a refusal the next 15 hybrid-capable sorts of bare keys take the counted form"""
    ctx = pool_ctx
    ctx.setTuning(capi.VRS_TUNE_MSD_POOL, 1)
    bad, good = pool_keys(POOL_MIN + 77, "tile_period", seed=2), pool_keys(POOL_MIN + 77, "uniform", seed=2)
    _, _, first = sort_and_stats(ctx, bad)
    assert first == (0, 1)
    for _ in range(15):
        out, _, c = sort_and_stats(ctx, good)
        assert c == (0, 0) and np.array_equal(out, np.sort(good))
    out, _, c = sort_and_stats(ctx, good)
    assert c == (1, 0) and np.array_equal(out, np.sort(good))


def test_pool_form_default_at_1e8_keys(gpu_context, oracle):
    """BASELINE.json configs[2] with the library's defaults: the pool form, 24 B/key"""
    n = 10 ** 8
    keys = np.random.RandomState(2).randint(0, 2 ** 32, size=n, dtype=np.uint32)
    gpu_context.setTuning(capi.VRS_TUNE_MSD_POOL, 1)
    out, stats, (took, refused) = sort_and_stats(gpu_context, keys)
    assert (took, refused) == (1, 0) and stats["digit_tables"] == 0 and stats["pool_sample"] == 1
    assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1


# ---- key + payload pairs: the STABLE pool form (round 5) --------------------------------------------------------------------
def sort_pairs_and_stats(ctx, keys, vals):
    lib, n = ctx.lib, keys.size
    k0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), keys)
    v0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), vals)
    k1, v1 = vrs.Buffer(ctx, S(4 * n)), vrs.Buffer(ctx, S(4 * n))
    ctx.profileReset()
    ctx.profileEnable(True)
    before = pool_counts(ctx)
    try:
        ctx.check(lib.vrs_sort_pairs_u32(ctx.handle, k0.handle, k1.handle, v0.handle, v1.handle, n))
        ok, ov = np.empty(n, np.uint32), np.empty(n, np.uint32)
        k0.downloadWithStagingBuffer(ok)
        v0.downloadWithStagingBuffer(ov)
        stats = {name: launches(ctx, kid) for kid, name in capi.KERNEL_NAMES.items()}
    finally:
        ctx.profileEnable(False)
        for b in (k0, k1, v0, v1):
            b.release()
    after = pool_counts(ctx)
    return ok, ov, stats, (after[0] - before[0], after[1] - before[1])


@pytest.mark.parametrize("dist", ["uniform", "28bit", "halves", "dups", "ties", "gauss"])
@pytest.mark.parametrize("n", [POOL_MIN + 77, 9000001, 30000001])
def test_pool_form_of_pairs_is_stable(pool_ctx, oracle, n, dist):
    """Pairs through the pool form: no counting read, and both passes place a tile by its RANK in the sampled region (decoupled
    look-back) -- payloads of equal keys come out in input order, as the reference's LSD passes leave them
    (multi_radixsort.comp:130-141).  `ties`: 2^16 distinct keys, long runs of equal keys in every bucket."""
    if dist == "ties":
        keys = (make_keys(n, "uniform", seed=6) & np.uint32(0xFFFF)) * np.uint32(65537)
    else:
        keys = pool_keys(n, dist, seed=n % 991)
    vals = make_keys(n, "uniform", seed=8)
    ok, ov, stats, (took, refused) = sort_pairs_and_stats(pool_ctx, keys, vals)
    rk, rv, _ = oracle.stable_sort_pairs(keys, vals)
    assert np.array_equal(ok, rk) and np.array_equal(ov, rv)
    assert (took, refused) == (1, 0), stats
    assert stats["pool_pass_a"] == 1 and stats["pool_pass_b"] == 1 and stats["digit_tables"] == 0 and stats["lookback_scatter"] == 0
    assert stats["local_sort"] == (2 if dist == "gauss" and n > 20000000 else 1)  # (gauss: the fullest buckets need the larger local sort -- a retry, no refusal)


@pytest.mark.parametrize("dist", ["tile_period", "hot_bucket", "rare_high_bit", "24bit"])
def test_pool_form_of_pairs_refuses_and_the_counted_form_sorts_the_untouched_input(pool_ctx, oracle, dist):
    n = 9000001
    keys = pool_keys(n, dist, seed=3)
    vals = make_keys(n, "uniform", seed=9)
    ok, ov, stats, (took, refused) = sort_pairs_and_stats(pool_ctx, keys, vals)
    rk, rv, _ = oracle.stable_sort_pairs(keys, vals)
    assert np.array_equal(ok, rk) and np.array_equal(ov, rv)
    assert took == 0 and (refused == 1 or dist == "24bit")  # (24 bits: the sample kernel arms nothing -- the LSD passes run; the others: a pass flags the sort)


def test_pool_form_of_pairs_back_to_back_and_switched_off(pool_ctx, oracle):
    """three sorts of one size on one context (the second and third start in the first one's kept layout when layouts are kept; the
    look-back words are cleared by every taken sort's local sort), then the form switched off: the counted form sorts the same"""
    n = 12000017
    vals = np.arange(n, dtype=np.uint32)
    for seed in (1, 2, 3):
        keys = make_keys(n, "uniform", seed=seed)
        ok, ov, stats, (took, refused) = sort_pairs_and_stats(pool_ctx, keys, vals)
        assert (took, refused) == (1, 0)
        order = np.argsort(keys, kind="stable").astype(np.uint32)
        assert np.array_equal(ov, order) and np.array_equal(ok, keys[order])
    pool_ctx.setTuning(capi.VRS_TUNE_MSD_POOL_PAIRS, 0)
    try:
        keys = make_keys(n, "uniform", seed=4)
        ok, ov, stats, (took, refused) = sort_pairs_and_stats(pool_ctx, keys, vals)
        order = np.argsort(keys, kind="stable").astype(np.uint32)
        assert (took, refused) == (0, 0) and stats["digit_tables"] == 1 and np.array_equal(ov, order)
    finally:
        pool_ctx.setTuning(capi.VRS_TUNE_MSD_POOL_PAIRS, 1)


def test_pool_form_of_pairs_at_1e8(gpu_context, oracle):
    """configs[3]'s size with the defaults: 10^8 pairs, payload = input position -- bit for bit what std::stable_sort leaves"""
    ctx = gpu_context
    n = 100000000
    keys = make_keys(n, "uniform", seed=1)
    vals = np.arange(n, dtype=np.uint32)
    ok, ov, stats, (took, refused) = sort_pairs_and_stats(ctx, keys, vals)
    assert (took, refused) == (1, 0), stats
    assert np.all(ok[1:] >= ok[:-1]) and np.array_equal(keys[ov], ok)
    same = ok[1:] == ok[:-1]
    assert np.all(ov[1:][same] > ov[:-1][same])  # equal keys: input order
    rk, rv, _ = oracle.stable_sort_pairs(keys, vals)
    assert np.array_equal(ok, rk) and np.array_equal(ov, rv)


@pytest.mark.parametrize("packed", [0, 1])
@pytest.mark.parametrize("dist", ["uniform", "28bit", "dups", "ties", "gauss"])
@pytest.mark.parametrize("n", [POOL_MIN + 77, 30000001])
def test_both_forms_of_the_pairs_local_sort(pool_ctx, oracle, n, dist, packed):
    """The pairs' local sort either carries the payloads through both LDS passes or sorts ONE word per pair (the key's bits below the bucket's
    | the pair's place in the bucket) and fetches each payload once, by place (VRS_TUNE_MSD_POOL_PAIRS_PACKED; by default the buckets' mean
    size decides).  Both are stable sorts of every bucket: bit for bit std::stable_sort's answer.  28bit: a bucket's keys differ in 14 bits,
    not 18; ties / dups: runs of equal keys whose payloads must keep their input order; gauss at 3e7: the retry in the 1024-thread shape."""
    if dist == "ties":
        keys = (make_keys(n, "uniform", seed=6) & np.uint32(0xFFFF)) * np.uint32(65537)
    else:
        keys = pool_keys(n, dist, seed=n % 983)
    vals = make_keys(n, "uniform", seed=11)
    pool_ctx.setTuning(capi.VRS_TUNE_MSD_POOL_PAIRS_PACKED, packed)
    try:
        ok, ov, stats, (took, refused) = sort_pairs_and_stats(pool_ctx, keys, vals)
    finally:
        pool_ctx.setTuning(capi.VRS_TUNE_MSD_POOL_PAIRS_PACKED, -1)
    rk, rv, _ = oracle.stable_sort_pairs(keys, vals)
    assert np.array_equal(ok, rk) and np.array_equal(ov, rv)
    assert (took, refused) == (1, 0), stats


@pytest.mark.parametrize("hook", ["hold_tile", "no_patience"])
def test_pool_form_of_pairs_with_a_tile_that_never_publishes(pool_ctx, oracle, hook):
    """The stable passes wait for their predecessors' look-back rows -- never without a bound: a tile whose predecessor does not publish
    within the spin budget refuses the sort (the caller's buffers have not been written), and the counted form sorts it.  hold_tile: tile
    3 of every slice never publishes in the first pass; no_patience: a budget of zero polls (whoever finds a row unpublished gives up)."""
    n = 9000001
    keys = pool_keys(n, "dups", seed=21)
    vals = make_keys(n, "uniform", seed=22)
    if hook == "hold_tile":
        pool_ctx.setTuning(capi.VRS_TUNE_DEBUG_HOLD_TILE, 3)
        pool_ctx.setTuning(capi.VRS_TUNE_LOOKBACK_SPIN_BUDGET, 16)
    else:
        pool_ctx.setTuning(capi.VRS_TUNE_LOOKBACK_SPIN_BUDGET, 0)
    try:
        ok, ov, stats, (took, refused) = sort_pairs_and_stats(pool_ctx, keys, vals)
    finally:
        pool_ctx.setTuning(capi.VRS_TUNE_DEBUG_HOLD_TILE, -1)
        pool_ctx.setTuning(capi.VRS_TUNE_LOOKBACK_SPIN_BUDGET, 4096)
    rk, rv, _ = oracle.stable_sort_pairs(keys, vals)
    assert np.array_equal(ok, rk) and np.array_equal(ov, rv)
    if hook == "hold_tile":
        assert (took, refused) == (0, 1), stats
    # and the next sort of the context is none the worse for it
    ok, ov, stats, (took, refused) = sort_pairs_and_stats(pool_ctx, keys, vals)
    assert np.array_equal(ok, rk) and np.array_equal(ov, rv)


@pytest.mark.parametrize("top_bits", [6, 8])
@pytest.mark.parametrize("dist", ["uniform", "gauss", "dups"])
def test_pool_form_with_the_other_cuts_of_its_buckets(pool_ctx, oracle, top_bits, dist):
    """VRS_TUNE_MSD_POOL_TOP_BITS: the 16384 buckets cut 8 + 6 bits between the two passes (the default until late in round 5) or 6 + 8
    instead of 7 + 7 -- the same buckets, the same result, for keys and for pairs (the ragged last tile's padding key carries the
    largest digit of whatever width)"""
    n = 9000001
    keys = pool_keys(n, dist, seed=top_bits)
    vals = make_keys(n, "uniform", seed=31)
    pool_ctx.setTuning(capi.VRS_TUNE_MSD_POOL_TOP_BITS, top_bits)
    try:
        out, stats, (took, refused) = sort_and_stats(pool_ctx, keys)
        assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1 and (took, refused) == (1, 0)
        ok, ov, stats, (took, refused) = sort_pairs_and_stats(pool_ctx, keys, vals)
        rk, rv, _ = oracle.stable_sort_pairs(keys, vals)
        assert np.array_equal(ok, rk) and np.array_equal(ov, rv) and (took, refused) == (1, 0)
    finally:
        pool_ctx.setTuning(capi.VRS_TUNE_MSD_POOL_TOP_BITS, 7)
    assert pool_ctx.lib.vrs_set_tuning(pool_ctx.handle, capi.VRS_TUNE_MSD_POOL_TOP_BITS, 5) == capi.VRS_ERROR_INVALID_ARGUMENT


def test_pool_form_of_pairs_beyond_the_512_thread_local_sort(gpu_context):
    """1.3e8 pairs: 16384 buckets of 7900 would need the 1024-thread local sort; the second pass takes seven bits instead (32768 buckets
    of 3970) and the 512-thread workgroup stays"""
    ctx = gpu_context
    n = 130000000
    keys = make_keys(n, "uniform", seed=3)
    vals = np.arange(n, dtype=np.uint32)
    ok, ov, stats, (took, refused) = sort_pairs_and_stats(ctx, keys, vals)
    assert (took, refused) == (1, 0), stats
    assert np.all(ok[1:] >= ok[:-1]) and np.array_equal(keys[ov], ok)
    same = ok[1:] == ok[:-1]
    assert np.all(ov[1:][same] > ov[:-1][same])


def _u64(ctx, fn):
    v = ctypes.c_uint64()
    ctx.check(getattr(ctx.lib, fn)(ctx.handle, ctypes.byref(v)))
    return v.value


def test_no_room_for_the_scratch_is_no_error_of_the_sort(pool_ctx, oracle):
    """a device with no room for the pool form's scratch (about 1.5 n slots) must not fail the sort -- the counted / LSD forms need
    none.  VRS_TUNE_DEBUG_POOL_NO_MEMORY makes the next allocation fail like a full device."""
    ctx = pool_ctx
    freed = ctypes.c_uint64()
    ctx.check(ctx.lib.vrs_context_trim_scratch(ctx.handle, ctypes.byref(freed)))  # (so that the sort below HAS to allocate)
    n = POOL_MIN + 4099
    keys = make_keys(n, "uniform", 31)
    before = _u64(ctx, "vrs_one_call_pool_no_memory")
    ctx.setTuning(capi.VRS_TUNE_DEBUG_POOL_NO_MEMORY, 1)
    out, stats, (took, refused) = sort_and_stats(ctx, keys)
    assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1
    assert (took, refused) == (0, 0) and stats["pool_pass_a"] == 0 and _u64(ctx, "vrs_one_call_pool_no_memory") == before + 1
    # with room again the form is taken (VRS_TUNE_MSD_POOL = 2 here: no adaptive skip), and what it allocated can be given back
    out, stats, (took, refused) = sort_and_stats(ctx, keys)
    assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1 and (took, refused) == (1, 0)
    ctx.check(ctx.lib.vrs_context_trim_scratch(ctx.handle, ctypes.byref(freed)))
    assert freed.value >= 4 * n  # the slack buffer alone holds n keys and their room
    out, stats, (took, refused) = sort_and_stats(ctx, keys)
    assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1 and (took, refused) == (1, 0)
    ctx.check(ctx.lib.vrs_context_trim_scratch(ctx.handle, None))


def test_no_room_for_the_payload_twins(pool_ctx):
    """... and pairs: the keys' scratch exists, the payloads' twins find no room -- the counted form sorts the pairs, stable."""
    ctx = pool_ctx
    n = POOL_MIN + 77
    keys = make_keys(n, "uniform", 5) & np.uint32(0xFFFF0FFF)
    vals = np.arange(n, dtype=np.uint32)
    ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, 8 * POOL_MIN // 5 + 8)  # (pairs: the hybrid threshold is 5/8 of the setting)
    ctx.check(ctx.lib.vrs_context_trim_scratch(ctx.handle, None))
    sort_and_stats(ctx, make_keys(n, "uniform", 6))  # the keys' scratch, of the size the pairs ask for too
    k0, k1 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), keys), vrs.Buffer(ctx, S(4 * n))
    v0, v1 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), vals), vrs.Buffer(ctx, S(4 * n))
    before = _u64(ctx, "vrs_one_call_pool_no_memory")
    try:
        ctx.setTuning(capi.VRS_TUNE_DEBUG_POOL_NO_MEMORY, 1)  # the next allocation: the payloads' twin of the overflow room
        ctx.check(ctx.lib.vrs_sort_pairs_u32(ctx.handle, k0.handle, k1.handle, v0.handle, v1.handle, n))
        ov = np.empty(n, np.uint32)
        v0.downloadWithStagingBuffer(ov)
        assert np.array_equal(ov, np.argsort(keys, kind="stable").astype(np.uint32))
        assert _u64(ctx, "vrs_one_call_pool_no_memory") == before + 1
    finally:
        ctx.setTuning(capi.VRS_TUNE_DEBUG_POOL_NO_MEMORY, 0)
        for b in (k0, k1, v0, v1):
            b.release()


def test_stale_layouts_back_off(oracle):
    """kept layouts had no back-off -- a workload of equal n whose distribution changes from sort to sort (here: uniform keys and
    28-bit keys, alternating) found every kept layout stale: two passes, a host round trip and the whole sort again, each time.
    After two stale layouts in a row the next 16 candidates sample for themselves.  (A context of its own: the shared one keeps no layouts.)"""
    n = POOL_MIN + 123457
    a = make_keys(n, "uniform", 3)
    b = pool_keys(n, "28bit", 4)  # another key range: another layout altogether (and one the form takes)
    with vrs.GPUContext(0) as ctx:
        ctx.setTuning(capi.VRS_TUNE_HYBRID_MIN_KEYS, POOL_MIN)
        ctx.setTuning(capi.VRS_TUNE_MSD_POOL_MIN_KEYS, POOL_MIN)
        ctx.setTuning(capi.VRS_TUNE_MSD_POOL, 2)
        sort_and_stats(ctx, a)  # a layout to keep
        assert pool_layouts(ctx) == (0, 0)
        for i in range(12):
            keys = b if i % 2 == 0 else a
            out, _, _ = sort_and_stats(ctx, keys)
            assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1
        # every reused layout was stale (the distributions alternate) -- but only two sorts paid for finding out, not all twelve
        assert pool_layouts(ctx) == (2, 2)
        for i in range(8):  # the pause ends after 16 candidates: a kept layout is tried again (and with a steady workload it fits)
            sort_and_stats(ctx, a)
        assert pool_layouts(ctx) == (4, 2)  # (ten candidates paused in the loop, six here, then two that reuse and fit)
