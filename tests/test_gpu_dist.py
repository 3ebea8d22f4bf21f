"""The multi-GPU step behind the C ABI (vrs_dist_*) with TWO ranks on the one GPU of the box: each rank is a host thread
with its own context / stream, the wire is the library's in-process loopback transport (device copies ordered by events).
This runs the rank-to-rank bookkeeping of vrs_dist.hip -- who sends what where, in which round, landing at which offset --
with non-zero peer counts, which world size 1 cannot: the concatenation of the ranks' outputs must be std::sort of the
concatenation of their shards, bit for bit.  (BASELINE.json configs[4]; no reference counterpart: VkRadixSort is single-GPU.)"""
import ctypes
import threading

import numpy as np
import pytest

import vkradixsort_amd as vrs
from vkradixsort_amd import capi

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _hybrid_shape_unless_the_test_says_otherwise(monkeypatch):
    """the tests of this file were written when the hybrid shape was the step's default: they keep asking for it (a test that wants
    the byte shape sets VRS_DIST_SHAPE itself; test_the_step_takes_the_byte_shape_by_default unsets it)"""
    monkeypatch.setenv("VRS_DIST_SHAPE", "hybrid")
S = vrs.Buffer.BufferSettings


def keys_of(kind, n, seed):
    rs = np.random.RandomState(seed)
    k = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32)
    if kind == "28bit":
        k >>= np.uint32(4)  # the reference's own test keys (MultiRadixSort.cpp:121-133)
    if kind == "16bit":
        k >>= np.uint32(16)  # everything under top byte 0: no byte-aligned cut can balance two ranks
    if kind == "24bit":
        k >>= np.uint32(8)
    if kind == "three_values":  # massive ties: no cut between key values balances anything
        k = (k % np.uint32(3)) * np.uint32(0x10000001)
    if kind == "clustered":  # three quarters of the keys under ONE top byte
        m = rs.rand(n) < 0.75
        k[m] = (k[m] & np.uint32(0x00FFFFFF)) | np.uint32(0x40000000)
    if kind == "hot_bucket":  # one top-14-bit bucket far beyond the local sort's capacity: that round's plan must refuse
        k[: n // 10] = (k[: n // 10] & np.uint32(0x3FFFF)) | np.uint32(0x9ABC0000 & ~0x3FFFF)
    return k


def run_ranks(shards, rounds, capacity=None, world=None, env_shape=None, steps=2, reserve=None):
    """`steps` steps with len(shards) ranks as threads; returns [(rc, sorted range or error text, stats)] per rank."""
    world = world or len(shards)
    lib = capi.load_library()
    hub = ctypes.c_void_p()
    assert lib.vrs_dist_loopback_create(world, ctypes.byref(hub)) == 0
    cap = capacity or int(max(s.size for s in shards) * 1.3) + 70000
    results = [None] * world
    errors = []

    def rank_main(r):
        try:
            with vrs.GPUContext(0) as gpu:
                if reserve is not None:
                    gpu.setTuning(capi.VRS_TUNE_MSD_RESERVE, reserve)
                tr = capi.DistTransport()
                assert lib.vrs_dist_loopback_transport(hub, r, ctypes.byref(tr)) == 0
                d = ctypes.c_void_p()
                rc = lib.vrs_dist_create_with_transport(gpu.handle, ctypes.byref(tr), r, world, cap, rounds, ctypes.byref(d))
                assert rc == 0, lib.vrs_dist_last_error(None)
                keys = shards[r]
                kb = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * max(keys.size, 1)), keys) if keys.size else vrs.Buffer(gpu, S(16))
                out_buf, out_n = ctypes.c_void_p(), ctypes.c_uint32()
                outs = []
                for _ in range(steps):  # more than once: every buffer and event of the step is reused
                    rc = lib.vrs_dist_sort_keys_u32(d, kb.handle, keys.size, ctypes.byref(out_buf), ctypes.byref(out_n))
                    if rc != 0:
                        outs.append((rc, lib.vrs_dist_last_error(d).decode()))
                        continue
                    gpu.waitIdle()
                    out = np.empty(out_n.value, np.uint32)
                    if out.size:
                        gpu.check(lib.vrs_buffer_download(gpu.handle, out_buf, out.ctypes.data_as(ctypes.c_void_p), out.nbytes))
                    outs.append((0, out))
                st = [ctypes.c_uint64() for _ in range(5)]
                lib.vrs_dist_stats(d, *[ctypes.byref(x) for x in st[:3]])
                lib.vrs_dist_grouped_rounds(d, ctypes.byref(st[3]))
                lib.vrs_dist_splitter_steps(d, ctypes.byref(st[4]))
                results[r] = (outs, tuple(x.value for x in st))  # (hybrid rounds, refused rounds, byte-shape steps, grouped rounds, splitter steps)
                kb.release()
                lib.vrs_dist_destroy(d)
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    hung = [t for t in threads if t.is_alive()]
    assert not hung, "a rank is still inside the step: the ranks did not leave it together"
    lib.vrs_dist_loopback_destroy(hub)
    assert not errors, errors
    return results


def check_sorted_ranges(shards, results, steps=2):
    allkeys = np.sort(np.concatenate(shards))
    for step in range(steps):
        outs = [results[r][0][step] for r in range(len(shards))]
        assert all(rc == 0 for rc, _ in outs), outs
        got = np.concatenate([o for _, o in outs])
        assert np.array_equal(got, allkeys)
        sizes = [o.size for _, o in outs]
        assert max(sizes) <= 1.15 * allkeys.size / len(shards) + 1


@pytest.mark.parametrize("rounds", [1, 4])
@pytest.mark.parametrize("kind", ["uniform", "28bit"])
def test_two_ranks_hybrid_shape(kind, rounds):
    """Every received sub-range is finished by the second MSD pass + local sort (no rank sorts from scratch)."""
    shards = [keys_of(kind, 3000017, 1000), keys_of(kind, 2600001, 1001)]
    res = run_ranks(shards, rounds)
    check_sorted_ranges(shards, res)
    for outs, (hybrid_rounds, fallback_rounds, byte_steps, _grouped, _splitters) in res:
        assert byte_steps == 0 and fallback_rounds == 0 and hybrid_rounds == 2 * rounds


@pytest.mark.parametrize("shape", ["hybrid", "byte"])
def test_two_ranks_with_the_msd_passes_reserving(shape, monkeypatch):
    """What shards of 10^8 keys do by default, forced at test sizes: the partition's first MSD pass and every round's second pass
    take their places by reservation (VRS_TUNE_MSD_RESERVE = 2); `grouped` is then grouped by top byte in no particular inner
    order, which is all the exchange needs."""
    if shape == "byte":
        monkeypatch.setenv("VRS_DIST_SHAPE", "byte")
    shards = [keys_of("uniform", 3000017, 1000), keys_of("28bit" if shape == "hybrid" else "uniform", 2600001, 1001)]
    if shape == "hybrid":
        shards[0] = keys_of("28bit", 3000017, 1000)
    res = run_ranks(shards, 2, reserve=2)
    check_sorted_ranges(shards, res)
    assert all(st[1] == 0 for _, st in res)


@pytest.mark.parametrize("rounds", [1, 4])
def test_two_ranks_one_empty_shard(rounds):
    shards = [keys_of("uniform", 2500003, 7), np.zeros(0, np.uint32)]
    res = run_ranks(shards, rounds, capacity=2000000 + 900000)
    check_sorted_ranges(shards, res)
    assert all(st[2] == 0 for _, st in res)  # still the hybrid shape: the empty shard has nothing to say about the key range


def test_two_ranks_a_refused_round_is_sorted_from_scratch():
    shards = [keys_of("hot_bucket", 3000000, 21), keys_of("uniform", 3000000, 22)]
    res = run_ranks(shards, 2, capacity=5000000)
    check_sorted_ranges(shards, res)
    assert sum(st[1] for _, st in res) >= 2  # the round with the hot bucket, in both steps, on the rank that owns it


@pytest.mark.parametrize("grouped_finish", [True, False])
@pytest.mark.parametrize("rounds", [1, 4])
def test_two_ranks_byte_shape(rounds, grouped_finish, monkeypatch):
    """Byte shape: top-byte partition pass, the keys landing grouped by top byte and every round finished by ONE counting read +
    the second MSD pass + the local sort (vrs_msd_finish_grouped_u32) -- or, switched off, one message per (sender, round) and a
    whole ranged sort per round."""
    monkeypatch.setenv("VRS_DIST_SHAPE", "byte")
    if not grouped_finish:
        monkeypatch.setenv("VRS_DIST_GROUPED_FINISH", "0")
    shards = [keys_of("uniform", 1200007, 3), keys_of("uniform", 999999, 4)]
    res = run_ranks(shards, rounds)
    check_sorted_ranges(shards, res)
    assert all(st[2] == 2 and st[0] == 0 for _, st in res)
    assert all((st[3] == 2 * rounds) == grouped_finish and (st[3] > 0) == grouped_finish for _, st in res), [st for _, st in res]


def test_two_ranks_byte_shape_a_round_the_grouped_finish_refuses_is_sorted_whole(monkeypatch):
    """One (top byte, next 8 bits) bucket of 300 000 keys: that round's plan refuses, its keys have not moved, and the rank
    sorts the sub-range whole; the other rounds take the grouped finish."""
    monkeypatch.setenv("VRS_DIST_SHAPE", "byte")
    shards = [keys_of("uniform", 2000003, 21), keys_of("uniform", 2100001, 22)]
    shards[0][:300000] = (shards[0][:300000] & np.uint32(0xFFFF)) | np.uint32(0x9ABC0000)
    res = run_ranks(shards, 2, capacity=3200000)
    check_sorted_ranges(shards, res)
    assert sum(st[1] for _, st in res) == 2 and sum(st[3] for _, st in res) == 6  # 2 steps x (4 rounds: 1 refused, 3 grouped)


def test_two_ranks_a_total_beyond_the_hybrid_shape_is_remembered(monkeypatch):
    """N_total / 16384 beyond the local sort's capacity (8 x 1e8 keys in earnest; here the test knob lowers the limit): the first
    step finds out from the gathered table and takes the byte shape, the following ones go straight to it -- no counting read
    and first MSD pass for nothing, one all-gather -- and the 65th looks again.  Every step bit-exact, both ranks alike."""
    monkeypatch.setenv("VRS_DIST_HYBRID_MAX_BUCKET", "100")
    shards = [keys_of("uniform", 1100003, 13), keys_of("uniform", 1000001, 14)]
    shards = [s[:300007] for s in shards]  # 66 steps: small shards keep the test short
    res = run_ranks(shards, 2, steps=66)
    check_sorted_ranges(shards, res, steps=66)
    assert all(st[2] == 66 and st[0] == 0 for _, st in res)


def test_two_ranks_small_shards_take_the_byte_shape_together():
    shards = [keys_of("uniform", 40000, 5), keys_of("uniform", 3000000, 6)]  # rank 0 is below the partition's minimum
    res = run_ranks(shards, 2, capacity=3200000)
    check_sorted_ranges(shards, res)
    assert all(st[2] == 2 for _, st in res)


@pytest.mark.parametrize("shape", ["hybrid", "byte"])
def test_three_ranks(shape, monkeypatch):
    """Three ranks: 85 / 86 top bytes each, rounds of 42-43 (no multiple of 8: some XCDs of the second MSD pass walk one group
    more); the byte shape's rounds go through the grouped finish with 7 sub-bucket bits."""
    if shape == "byte":
        monkeypatch.setenv("VRS_DIST_SHAPE", "byte")
    shards = [keys_of("uniform", 1500000 + 777 * r, 60 + r) for r in range(3)]
    res = run_ranks(shards, 2)
    check_sorted_ranges(shards, res)
    if shape == "byte":
        assert all(st[3] == 4 and st[1] == 0 for _, st in res), [st for _, st in res]


@pytest.mark.parametrize("kind", ["16bit", "clustered", "24bit"])
@pytest.mark.parametrize("shape", ["hybrid_first", "byte_first"])
def test_two_ranks_concentrated_top_bytes_are_cut_at_sampled_keys(kind, shape, monkeypatch):
    """Top bytes too concentrated for byte-aligned cuts (small keys, clustered keys): the step pools 2048 sampled keys per rank,
    cuts at their quantiles and groups the shard by range -- bit-exact, every rank within 15 % of the even share -- whether the
    step tried the hybrid shape first (two all-gathers before the sample) or went straight to the byte shape (one)."""
    if shape == "byte_first":
        monkeypatch.setenv("VRS_DIST_SHAPE", "byte")
    shards = [keys_of(kind, 2000000, 31), keys_of(kind, 2000000, 32)]
    res = run_ranks(shards, 2)
    check_sorted_ranges(shards, res)
    assert all(st[4] == 2 for _, st in res), [st for _, st in res]


def test_three_ranks_uneven_shards_cut_at_sampled_keys():
    """Shards of very different sizes (one empty): every sample weighs its shard's size / 2048 keys, an empty shard adds nothing."""
    shards = [keys_of("16bit", 3000000, 33), keys_of("16bit", 400000, 34), np.empty(0, np.uint32)]
    res = run_ranks(shards, 2)
    check_sorted_ranges(shards, res)
    assert all(st[4] == 2 for _, st in res), [st for _, st in res]


@pytest.mark.parametrize("kind", ["three_values", "16bit_sampling_off"])
def test_two_ranks_unbalanced_key_ranges_leave_together(kind, monkeypatch):
    """Massive ties (three key values for two ranks x two rounds: one value holds more than a rank's share), or concentrated top
    bytes with the sampled splitters switched off: VRS_ERROR_UNBALANCED on BOTH ranks, nobody hangs."""
    if kind == "16bit_sampling_off":
        monkeypatch.setenv("VRS_DIST_SAMPLED_SPLITTERS", "0")
        kind = "16bit"
    shards = [keys_of(kind, 2000000, 31), keys_of(kind, 2000000, 32)]
    res = run_ranks(shards, 2)
    for outs, _ in res:
        assert all(rc == capi.VRS_ERROR_UNBALANCED for rc, _ in outs), outs


@pytest.mark.parametrize("rounds", [1, 4])
@pytest.mark.parametrize("shape", ["hybrid", "byte"])
def test_eight_ranks_loopback(shape, rounds, monkeypatch):
    """World size 8 (BASELINE.json configs[4]'s node) on the one GPU of the box: eight contexts, 8 x 2e6 keys, uniform and the
    reference's 28-bit test keys, one shard empty; 32 top bytes per rank, rounds of 8.  The concatenation of the eight outputs
    must be std::sort of all keys, every rank within 15 % of the even share."""
    if shape == "byte":
        monkeypatch.setenv("VRS_DIST_SHAPE", "byte")
    for kind, empty in (("uniform", None), ("28bit", None), ("uniform", 5)):
        shards = [keys_of(kind, 2000000 + 1001 * r, 80 + r) if r != empty else np.empty(0, np.uint32) for r in range(8)]
        res = run_ranks(shards, rounds, capacity=2800000, steps=1)
        check_sorted_ranges(shards, res, steps=1)
        if shape == "hybrid":
            assert all(st[0] == rounds and st[1] == 0 for _, st in res), [st for _, st in res]


def test_eight_ranks_small_keys_cut_at_sampled_keys():
    """World size 8, 24-bit keys: 32 parts cut at the quantiles of 16384 pooled samples."""
    shards = [keys_of("24bit", 1000000 + 77 * r, 90 + r) for r in range(8)]
    res = run_ranks(shards, 4, steps=1)
    check_sorted_ranges(shards, res, steps=1)
    assert all(st[4] == 1 for _, st in res), [st for _, st in res]


@pytest.mark.parametrize("shape", ["hybrid_first", "byte_first"])
def test_two_ranks_a_shard_above_its_capacity_fails_on_both(shape, monkeypatch):
    """Rank 1's shard exceeds the capacity it was created with: it reports INVALID_ARGUMENT, rank 0 PEER -- after both took
    part in the same collectives (byte shape first: the status rides in the one all-gather of the top-byte prefixes)."""
    if shape == "byte_first":
        monkeypatch.setenv("VRS_DIST_SHAPE", "byte")
    shards = [keys_of("uniform", 1000000, 41), keys_of("uniform", 1500000, 42)]
    res = run_ranks(shards, 1, capacity=1200000)
    assert all(rc == capi.VRS_ERROR_PEER for rc, _ in res[0][0]), res[0][0]
    assert all(rc == capi.VRS_ERROR_INVALID_ARGUMENT for rc, _ in res[1][0]), res[1][0]


def test_bench_multi_path_at_world_eight_over_the_loopback_transport():
    """bench.py's multi-GPU path (step, cross-rank verification, JSON assembly) with eight ranks as threads on the one GPU:
    what the driver runs on an 8-GPU node, minus RCCL -- n_gpus = 8, the byte breakdown, cpu_baseline, all checks true."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    p = subprocess.run([sys.executable, str(root / "bench.py"), "--loopback-ranks", "8", "--n", "2e6", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["steps"] == 3 and line["unit"] == "Gkeys/s" and line["value"] > 0
    assert all(line["verified"].values()), line["verified"]
    assert sum(line["shard_sizes"]) == 8 * 2000000 and max(line["shard_sizes"]) <= 1.15 * 2000000
    assert sum(line["config"]["hbm_bytes_per_key_breakdown"].values()) == line["config"]["hbm_bytes_per_key"]
    assert line["config"]["transport"].startswith("loopback") and line["cpu_baseline"]["kind"] == "port"
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["launches"] > 0


def test_cpp_host_drives_the_step_over_the_loopback_transport():
    """distsortexample: the step from a C++ host through the C ABI alone -- one std::thread per rank, the in-process transport --
    verified the reference's way (equal to std::sort of all keys, MultiRadixSort.cpp:141-161)."""
    import subprocess
    from vkradixsort_amd import build
    exe = build.build_dist_example()
    for args in (["2", "1500000", "2"], ["3", "700001", "1", "7"], ["2", "40000", "1"], ["4", "600000", "2", "9", "20"]):
        p = subprocess.run([str(exe), *args], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        assert "[DistSort] Test passed." in p.stdout


def test_msd_halves_several_finishes_enqueued_before_any_is_asked_about():
    """vrs_msd_partition_u32 + TWO vrs_msd_finish_u32 (the lower and the upper half of the top bytes, like two rounds of the
    step), both enqueued before either plan is looked at: a ticket per finish, vrs_msd_finish_status_at answers for that
    finish whatever ran since (asked in reverse order), the concatenation is std::sort of the input.  A third finish over a
    sub-range with one bucket far beyond the local sort's capacity is refused -- its ticket says so -- while the tickets of
    the other two still answer."""
    lib = capi.load_library()
    n = (1 << 23) + 12345
    keys = keys_of("uniform", n, 77)
    with vrs.GPUContext(0) as gpu:
        kb = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), keys)
        grouped, out = vrs.Buffer(gpu, S(4 * n)), vrs.Buffer(gpu, S(4 * n))
        counts = vrs.Buffer(gpu, S(4 * capi.MSD_COUNT_WORDS))
        gpu.check(lib.vrs_msd_partition_u32(gpu.handle, kb.handle, grouped.handle, counts.handle, n))
        host_counts = np.empty(capi.MSD_COUNT_WORDS, np.uint32)
        counts.downloadWithStagingBuffer(host_counts)
        hist, shift = host_counts[:16384], int(host_counts[16384 + 8 * 256])
        assert shift == 18 and int(hist.sum()) == n
        cut = int(hist[: 128 * 64].sum())  # keys under top bytes 0..127
        tickets, parts = [], [(0, cut, 0, 128), (cut, n - cut, 128, 256)]
        round_counts = []
        for off, cnt, lo, hi in parts:
            c = np.zeros(capi.MSD_COUNT_WORDS, np.uint32)
            c[lo * 64: hi * 64] = hist[lo * 64: hi * 64]
            c[16384 + 8 * 256] = shift
            cb = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * capi.MSD_COUNT_WORDS), c)
            round_counts.append(cb)
            gv = vrs.Buffer(gpu, S(4 * cnt), device_ptr=grouped.getDeviceAddress() + 4 * off)
            ov = vrs.Buffer(gpu, S(4 * cnt), device_ptr=out.getDeviceAddress() + 4 * off)
            gpu.check(lib.vrs_msd_finish_u32(gpu.handle, gv.handle, ov.handle, cb.handle, cnt, 0))
            t = ctypes.c_uint32()
            gpu.check(lib.vrs_msd_finish_ticket(gpu.handle, ctypes.byref(t)))
            tickets.append(t.value)
            gv.release()
            ov.release()
        assert tickets[1] == tickets[0] + 1
        # a third finish the plan must refuse: the same lower half, but its histogram claims one bucket of 100 000 keys
        c = np.zeros(capi.MSD_COUNT_WORDS, np.uint32)
        c[: 128 * 64] = hist[: 128 * 64]
        moved = min(100000, cut // 2)
        c[5] += moved
        big = int(np.argmax(c[6: 128 * 64])) + 6
        take = moved
        for b in range(6, 128 * 64):  # keep the total: take the keys from other buckets
            d = min(int(c[b]), take)
            c[b] -= d
            take -= d
            if take == 0:
                break
        assert int(c[: 128 * 64].sum()) == cut and big >= 6
        c[16384 + 8 * 256] = shift
        cb = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * capi.MSD_COUNT_WORDS), c)
        scratch_in, scratch_out = vrs.Buffer(gpu, S(4 * cut)), vrs.Buffer(gpu, S(4 * cut))
        gpu.check(lib.vrs_msd_finish_u32(gpu.handle, scratch_in.handle, scratch_out.handle, cb.handle, cut, 0))
        t = ctypes.c_uint32()
        gpu.check(lib.vrs_msd_finish_ticket(gpu.handle, ctypes.byref(t)))
        took = ctypes.c_int(-1)
        gpu.check(lib.vrs_msd_finish_status_at(gpu.handle, t.value, ctypes.byref(took)))
        assert took.value == 0
        for tk in reversed(tickets):
            gpu.check(lib.vrs_msd_finish_status_at(gpu.handle, tk, ctypes.byref(took)))
            assert took.value == 1
        assert lib.vrs_msd_finish_status_at(gpu.handle, 0, ctypes.byref(took)) == capi.VRS_ERROR_INVALID_ARGUMENT
        assert lib.vrs_msd_finish_status_at(gpu.handle, (t.value - 40) & 0xFFFFFFFF, ctypes.byref(took)) == capi.VRS_ERROR_INVALID_ARGUMENT
        res = np.empty(n, np.uint32)
        out.downloadWithStagingBuffer(res)
        assert np.array_equal(res, np.sort(keys))
        for b in [kb, grouped, out, counts, cb, scratch_in, scratch_out] + round_counts:
            b.release()


@pytest.mark.parametrize("first,top_bytes,n", [(0, 256, 6000001), (0x40, 64, 5000003), (0xA0, 32, 4000001), (0x10, 9, 3000007),
                                                (0xFE, 2, 2500001), (0x7F, 1, 2000003), (0x03, 100, 7000001), (0xE0, 32, 70001)])
def test_msd_finish_grouped_sorts_keys_that_arrive_grouped_by_top_byte(first, top_bytes, n):
    """vrs_msd_finish_grouped_u32: keys of top bytes [first, first + count), grouped by top byte in any inner order (what a rank
    holds after the exchange of a large multi-GPU sort) -> one counting read, the second MSD pass by the next 8 (or 14 -
    ceil(log2(count))) bits, the local sort.  Bit-exact vs std::sort for group counts from 1 to 256 (no multiple of 8 among them:
    some XCDs then walk one group more), and the ticket says the form was taken."""
    lib = capi.load_library()
    rs = np.random.RandomState(first * 1000 + top_bytes)
    keys = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32)
    keys = (keys & np.uint32(0x00FFFFFF)) | ((rs.randint(first, first + top_bytes, size=n).astype(np.uint32)) << np.uint32(24))
    grouped_host = keys[np.argsort(keys >> np.uint32(24), kind="stable")]
    with vrs.GPUContext(0) as gpu:
        g = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), grouped_host)
        out = vrs.Buffer(gpu, S(4 * n))
        for _ in range(2):  # twice: the second call finds the status words already clear
            gpu.check(lib.vrs_buffer_upload(gpu.handle, g.handle, grouped_host.ctypes.data_as(ctypes.c_void_p), grouped_host.nbytes))
            gpu.check(lib.vrs_msd_finish_grouped_u32(gpu.handle, g.handle, out.handle, n, first, top_bytes))
            t = ctypes.c_uint32()
            gpu.check(lib.vrs_msd_finish_ticket(gpu.handle, ctypes.byref(t)))
            took = ctypes.c_int(-1)
            gpu.check(lib.vrs_msd_finish_status_at(gpu.handle, t.value, ctypes.byref(took)))
            assert took.value == 1
            res = np.empty(n, np.uint32)
            out.downloadWithStagingBuffer(res)
            assert np.array_equal(res, np.sort(keys))
        g.release()
        out.release()


@pytest.mark.parametrize("first,top_bytes,n", [(0, 256, 6000001), (0x40, 64, 5000003), (0xA0, 32, 4000001), (0x10, 9, 3000007),
                                                (0xFE, 2, 2500001), (0x7F, 1, 2000003), (0x03, 100, 7000001), (0xE0, 8, 9000001),
                                                (0x20, 32, 70001)])
def test_msd_finish_grouped_counts_needs_no_counting_read(first, top_bytes, n):
    """vrs_msd_finish_grouped_counts_u32: the same second half for a caller that knows every top byte's keys (the multi-GPU step
    does) -- the pool form's plan samples the grouped keys, the second pass scatters into the buckets' slack regions, the local sort
    reads every bucket in one piece: 16 instead of 20 bytes per key, no counting read.  Bit-exact vs std::sort for 1 to 256 top
    bytes; where the form has no shape for the buckets (a single top byte with 2e6 keys needs more than 8 bits) or too few keys, the
    counted finish runs instead -- the ticket says taken either way; wrong counts are an error."""
    lib = capi.load_library()
    rs = np.random.RandomState(first * 1000 + top_bytes + 1)
    keys = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32)
    keys = (keys & np.uint32(0x00FFFFFF)) | ((rs.randint(first, first + top_bytes, size=n).astype(np.uint32)) << np.uint32(24))
    grouped_host = keys[np.argsort(keys >> np.uint32(24), kind="stable")]
    counts = np.bincount((grouped_host >> np.uint32(24)).astype(np.int64) - first, minlength=top_bytes).astype(np.uint32)
    cptr = counts.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
    with vrs.GPUContext(0) as gpu:
        g = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), grouped_host)
        out = vrs.Buffer(gpu, S(4 * n))
        gpu.profileEnable(True)
        for rep in range(2):
            gpu.check(lib.vrs_buffer_upload(gpu.handle, g.handle, grouped_host.ctypes.data_as(ctypes.c_void_p), grouped_host.nbytes))
            gpu.profileReset()
            gpu.check(lib.vrs_msd_finish_grouped_counts_u32(gpu.handle, g.handle, out.handle, n, first, top_bytes, cptr))
            t = ctypes.c_uint32()
            gpu.check(lib.vrs_msd_finish_ticket(gpu.handle, ctypes.byref(t)))
            took = ctypes.c_int(-1)
            gpu.check(lib.vrs_msd_finish_status_at(gpu.handle, t.value, ctypes.byref(took)))
            assert took.value == 1
            res = np.empty(n, np.uint32)
            out.downloadWithStagingBuffer(res)
            assert np.array_equal(res, np.sort(keys))
            pooled = gpu.profileQuery(capi.VRS_KERNEL_POOL_PASS_B)[0] == 1
            counted = gpu.profileQuery(capi.VRS_KERNEL_DIGIT_TABLES)[0] == 1
            assert pooled != counted
            # the pool form's second half wherever 6 .. 8 bits below the top byte make buckets a workgroup can hold
            assert pooled == (n >= (1 << 20) and n / (top_bytes * 256) < 13000)
        gpu.profileEnable(False)
        bad = counts.copy()
        bad[0] += 1
        assert lib.vrs_msd_finish_grouped_counts_u32(gpu.handle, g.handle, out.handle, n, first, top_bytes,
                                                     bad.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))) == capi.VRS_ERROR_INVALID_ARGUMENT
        g.release()
        out.release()


def test_msd_finish_grouped_counts_refuses_what_does_not_fit_and_keys_off_their_top_byte():
    """a bucket beyond its region or the local sort (375 000 keys under one (top byte, next 8 bits)), a key whose top byte is not
    the one its place says: the second pass flags the finish, nothing of `grouped` has moved"""
    lib = capi.load_library()
    n, first, top_bytes = 3000001, 0x20, 16
    rs = np.random.RandomState(5)
    keys = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32)
    keys = (keys & np.uint32(0x00FFFFFF)) | ((rs.randint(first, first + top_bytes, size=n).astype(np.uint32)) << np.uint32(24))
    hot = keys.copy()
    hot[: n // 8] = (hot[: n // 8] & np.uint32(0xFFFF)) | np.uint32(0x25AB0000)
    stray = keys[np.argsort(keys >> np.uint32(24), kind="stable")].copy()
    with vrs.GPUContext(0) as gpu:
        out = vrs.Buffer(gpu, S(4 * n))
        for case, host in (("hot", hot[np.argsort(hot >> np.uint32(24), kind="stable")]), ("stray", stray)):
            counts = np.bincount((host >> np.uint32(24)).astype(np.int64) - first, minlength=top_bytes).astype(np.uint32)
            if case == "stray":
                host[1234] = (host[1234] & np.uint32(0x00FFFFFF)) | np.uint32((first + top_bytes - 1) << 24)  # (counts as given: still add up)
            g = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), host)
            gpu.check(lib.vrs_msd_finish_grouped_counts_u32(gpu.handle, g.handle, out.handle, n, first, top_bytes,
                                                            counts.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))
            took = ctypes.c_int(-1)
            gpu.check(lib.vrs_msd_finish_status(gpu.handle, ctypes.byref(took)))
            assert took.value == 0, case
            back = np.empty(n, np.uint32)
            g.downloadWithStagingBuffer(back)
            assert np.array_equal(back, host), case
            g.release()
        out.release()


def test_msd_finish_grouped_refuses_a_bucket_no_workgroup_can_hold_and_rejects_bad_ranges():
    lib = capi.load_library()
    n, first, top_bytes = 3000001, 0x20, 16
    rs = np.random.RandomState(5)
    keys = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32)
    keys = (keys & np.uint32(0x00FFFFFF)) | ((rs.randint(first, first + top_bytes, size=n).astype(np.uint32)) << np.uint32(24))
    keys[: n // 8] = (keys[: n // 8] & np.uint32(0xFFFF)) | np.uint32(0x25AB0000)  # 375 000 keys in ONE (top byte, next 8 bits) bucket
    grouped_host = keys[np.argsort(keys >> np.uint32(24), kind="stable")]
    with vrs.GPUContext(0) as gpu:
        g = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), grouped_host)
        out = vrs.Buffer(gpu, S(4 * n))
        gpu.check(lib.vrs_msd_finish_grouped_u32(gpu.handle, g.handle, out.handle, n, first, top_bytes))
        took = ctypes.c_int(-1)
        gpu.check(lib.vrs_msd_finish_status(gpu.handle, ctypes.byref(took)))
        assert took.value == 0
        back = np.empty(n, np.uint32)
        g.downloadWithStagingBuffer(back)
        assert np.array_equal(back, grouped_host)  # refused: nothing has moved
        assert lib.vrs_msd_finish_grouped_u32(gpu.handle, g.handle, out.handle, n, 250, 10) == capi.VRS_ERROR_INVALID_ARGUMENT
        assert lib.vrs_msd_finish_grouped_u32(gpu.handle, g.handle, out.handle, n, 0, 0) == capi.VRS_ERROR_INVALID_ARGUMENT
        assert lib.vrs_msd_finish_grouped_u32(gpu.handle, g.handle, g.handle, n, 0, 16) == capi.VRS_ERROR_INVALID_ARGUMENT
        g.release()
        out.release()


@pytest.mark.parametrize("first,top_bytes,n,own_share", [(0, 256, 6000001, 0.125), (0, 256, 6000001, 1.0), (0x40, 64, 5000003, 0.5),
                                                          (0x10, 9, 3000007, 0.3), (0xE0, 8, 9000001, 0.0), (0x20, 32, 70001, 0.4),
                                                          (0x7F, 1, 2000003, 0.6), (0x03, 100, 7000001, 0.9)])
def test_msd_finish_grouped_split_reads_a_ranks_own_keys_where_they_lie(first, top_bytes, n, own_share):
    """vrs_msd_finish_grouped_split_u32: a part of every top byte's keys -- what a rank keeps for itself -- is not in the grouped buffer
    (the LAST slots of the top byte's range there are a hole) but in a second buffer, the top bytes' own parts one after the other from
    an offset on.  The second pass reads a top byte as two pieces; the result is std::sort's, the hole's content is never looked at,
    neither buffer is written.  Where the form cannot run (too few keys, no shape) the own parts are copied into the holes and the
    counted finish runs."""
    lib = capi.load_library()
    rs = np.random.RandomState(first * 1000 + top_bytes + 7)
    keys = rs.randint(0, 2 ** 32, size=n, dtype=np.uint32)
    keys = (keys & np.uint32(0x00FFFFFF)) | ((rs.randint(first, first + top_bytes, size=n).astype(np.uint32)) << np.uint32(24))
    grouped_host = keys[np.argsort(keys >> np.uint32(24), kind="stable")]
    counts = np.bincount((grouped_host >> np.uint32(24)).astype(np.int64) - first, minlength=top_bytes).astype(np.uint32)
    own_counts = np.minimum(counts, (counts * own_share + rs.randint(0, 3, size=top_bytes)).astype(np.uint32)) if own_share < 1.0 else counts.copy()
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
    own_offset = 12345
    own_host = np.full(own_offset + int(own_own := own_counts.sum()) + 99, 0xDEADBEEF, np.uint32)
    holed = grouped_host.copy()
    at = own_offset
    for a in range(top_bytes):
        lo = starts[a] + int(counts[a]) - int(own_counts[a])
        own_host[at:at + int(own_counts[a])] = grouped_host[lo:lo + int(own_counts[a])]
        holed[lo:lo + int(own_counts[a])] = 0x0BADF00D  # the hole: garbage with a wrong top byte
        at += int(own_counts[a])
    P = ctypes.POINTER(ctypes.c_uint32)
    with vrs.GPUContext(0) as gpu:
        g = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), holed)
        o = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * own_host.size), own_host)
        out = vrs.Buffer(gpu, S(4 * n))
        gpu.profileEnable(True)
        gpu.profileReset()
        gpu.check(lib.vrs_msd_finish_grouped_split_u32(gpu.handle, g.handle, o.handle, own_offset, out.handle, n, first, top_bytes,
                                                       counts.ctypes.data_as(P), own_counts.ctypes.data_as(P)))
        took = ctypes.c_int(-1)
        gpu.check(lib.vrs_msd_finish_status(gpu.handle, ctypes.byref(took)))
        assert took.value == 1
        res = np.empty(n, np.uint32)
        out.downloadWithStagingBuffer(res)
        assert np.array_equal(res, np.sort(keys))
        pooled = gpu.profileQuery(capi.VRS_KERNEL_POOL_PASS_B)[0] == 1
        gpu.profileEnable(False)
        assert pooled == (n >= (1 << 20) and n / (top_bytes * 256) < 13000)
        back = np.empty(own_host.size, np.uint32)
        o.downloadWithStagingBuffer(back)
        assert np.array_equal(back, own_host)
        if pooled:  # (the counted finish filled the holes: that is its way in)
            gb = np.empty(n, np.uint32)
            g.downloadWithStagingBuffer(gb)
            assert np.array_equal(gb, holed)
        bad = own_counts.copy()
        bad[0] = counts[0] + 1
        assert lib.vrs_msd_finish_grouped_split_u32(gpu.handle, g.handle, o.handle, own_offset, out.handle, n, first, top_bytes,
                                                    counts.ctypes.data_as(P), bad.ctypes.data_as(P)) == capi.VRS_ERROR_INVALID_ARGUMENT
        for b in (g, o, out):
            b.release()


@pytest.mark.parametrize("world", [1, 3])
def test_the_step_takes_the_byte_shape_by_default(world, monkeypatch):
    """no VRS_DIST_SHAPE: the byte shape -- contract partition pass, the rank's own keys left in place, every round finished by the
    pool form's second half (since round 5 it moves fewer bytes than the hybrid shape at every world size and takes every total)"""
    monkeypatch.delenv("VRS_DIST_SHAPE", raising=False)
    shards = [keys_of("uniform", 3000017 - 200001 * r, 1000 + r) for r in range(world)]
    res = run_ranks(shards, 2)
    check_sorted_ranges(shards, res)
    for outs, (hybrid_rounds, fallback_rounds, byte_steps, grouped, _splitters) in res:
        assert hybrid_rounds == 0 and fallback_rounds == 0 and byte_steps == 2 and grouped == 2 * 2


@pytest.mark.parametrize("shape", ["hybrid", "byte"])
def test_a_wire_that_costs_something_changes_the_timing_not_the_result(monkeypatch, shape):
    """vrs_dist_loopback_set_wire (round 5's verdict: the step had only ever run over a wire that costs nothing -- device copies): every
    group of sends / receives holds the receiver's stream for (bytes on its busiest link) / rate + a latency, every small collective for
    the latency.  Three ranks, two rounds: the ranges are still std::sort of the shards' concatenation, bit for bit -- and a slower wire
    makes a slower step (the model is really in the path)."""
    import time
    monkeypatch.setenv("VRS_DIST_SHAPE", shape)
    world, n = 3, 1500003
    shards = [keys_of("uniform", n + 17 * r, 40 + r) for r in range(world)]
    allkeys = np.sort(np.concatenate(shards))
    lib = capi.load_library()
    hub = ctypes.c_void_p()
    assert lib.vrs_dist_loopback_create(world, ctypes.byref(hub)) == 0
    cap = int(n * 1.3) + 70000
    gate = threading.Barrier(world)
    times = {}
    outs = {}
    errors = []

    def rank_main(r):
        try:
            with vrs.GPUContext(0) as gpu:
                tr = capi.DistTransport()
                assert lib.vrs_dist_loopback_transport(hub, r, ctypes.byref(tr)) == 0
                d = ctypes.c_void_p()
                assert lib.vrs_dist_create_with_transport(gpu.handle, ctypes.byref(tr), r, world, cap, 2, ctypes.byref(d)) == 0
                kb = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * shards[r].size), shards[r])
                out_buf, out_n = ctypes.c_void_p(), ctypes.c_uint32()
                for label, gbps, lat in (("free", 0.0, 0.0), ("slow", 0.5, 50.0), ("free_again", 0.0, 0.0)):
                    gate.wait()
                    if r == 0:
                        assert lib.vrs_dist_loopback_set_wire(hub, gbps, lat) == 0
                    gate.wait()
                    best = None
                    for _ in range(3):
                        gpu.waitIdle()
                        gate.wait()
                        t0 = time.perf_counter()
                        assert lib.vrs_dist_sort_keys_u32(d, kb.handle, shards[r].size, ctypes.byref(out_buf), ctypes.byref(out_n)) == 0
                        gpu.waitIdle()
                        dt = time.perf_counter() - t0
                        best = dt if best is None else min(best, dt)
                    out = np.empty(out_n.value, np.uint32)
                    gpu.check(lib.vrs_buffer_download(gpu.handle, out_buf, out.ctypes.data_as(ctypes.c_void_p), out.nbytes))
                    outs[(label, r)] = out
                    times[(label, r)] = best
                kb.release()
                lib.vrs_dist_destroy(d)
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))
            gate.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not [t for t in threads if t.is_alive()] and not errors, errors
    assert lib.vrs_dist_loopback_set_wire(None, 1.0, 0.0) != 0 and lib.vrs_dist_loopback_set_wire(hub, -1.0, 0.0) != 0
    lib.vrs_dist_loopback_destroy(hub)
    for label in ("free", "slow", "free_again"):
        assert np.array_equal(np.concatenate([outs[(label, r)] for r in range(world)]), allkeys), label
    # a rank receives about 2/3 of 6 MB from two peers: 2 MB on its busiest link = 4 ms at 0.5 GB/s, plus 50 us for each of the collectives
    free = max(times[("free", r)] for r in range(world))
    slow = max(times[("slow", r)] for r in range(world))
    again = max(times[("free_again", r)] for r in range(world))
    assert slow > free + 3e-3 and again < free + 2e-3, (free, slow, again)
