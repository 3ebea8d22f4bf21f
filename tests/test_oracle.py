"""CPU tests: the oracle against the committed golden vectors, the reference's acceptance criterion
(std::sort equality, MultiRadixSort.cpp:141-161) and independent numpy cross-checks."""
from pathlib import Path

import numpy as np
import pytest

GOLDEN = sorted((Path(__file__).parent / "golden").glob("*.npz"))


def test_mt19937_known_answer(oracle):
    # C++ standard [rand.predef]: 10000th invocation of a default-constructed mt19937 is 4123659995
    assert int(oracle.mt19937(5489, 10000)[-1]) == 4123659995
    # seed 12345 starts 3992670690, 3823185381, 1358822685, 561383553
    assert oracle.mt19937(12345, 4).tolist() == [3992670690, 3823185381, 1358822685, 561383553]
    # reference-faithful 28-bit keys == raw >> 4
    assert oracle.mt19937(12345, 2, 4).tolist() == [249541918, 238949086]
    assert np.array_equal(oracle.mt19937(99, 5000), np.random.RandomState(99).randint(0, 2 ** 32, 5000, dtype=np.uint32))


@pytest.mark.parametrize("n,B,W", [(10 ** 6, 32, 123), (10 ** 6, 1, 3907), (10 ** 6, 4096, 1), (10 ** 7, 32, 1221),
                                   (10 ** 7, 512, 77), (10 ** 8, 32, 12208), (10 ** 8, 128, 3052), (10 ** 8, 4096, 96)])
def test_workgroup_count_matches_reference_readme(oracle, n, B, W):
    # (N,B)->W pairs quoted by the reference (README.md:257-261) and derived in SURVEY.md appendix A
    assert oracle.workgroup_count(n, B) == W


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: p.stem)
def test_oracle_reproduces_golden(oracle, path):
    g = np.load(path)
    n, B, seed, tbz, W = (int(x) for x in g["meta"])
    keys = g["keys"]
    assert np.array_equal(keys, oracle.mt19937(seed, n, tbz))
    assert oracle.workgroup_count(n, B) == W
    cur = keys
    for i in range(4):
        hist = oracle.histograms(cur, 8 * i, W, B)
        assert np.array_equal(hist, g[f"hist{i}"])
        assert int(hist.sum()) == n
        assert np.array_equal(oracle.offsets(hist, W), g[f"offsets{i}"])
        cur = oracle.scatter(cur, hist, 8 * i, W, B)
        assert np.array_equal(cur, g[f"pass{i}"])
    assert np.array_equal(cur, g["sorted"])
    assert np.array_equal(oracle.multi_radixsort(keys, B), g["sorted"])
    assert np.array_equal(oracle.single_radixsort(keys), g["sorted"])
    assert oracle.test_sort(g["sorted"], cur) == -1


@pytest.mark.parametrize("subgroup_size", [32, 64])
@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: p.stem)
def test_two_independent_restatements_agree_on_every_stage_table(oracle, path, subgroup_size):
    """The stage-level fixtures (histogram table, per-workgroup offset table, every pass's output) are pinned by TWO restatements of the
    reference's shaders that share no code: oracle/vrs_oracle.c (the net effect: counters and running offsets, in C) and
    tests/golden/glsl_emulation.py (the shader text invocation by invocation: LDS atomicAdd, the bin_flags bit masks and bitCount,
    subgroupAdd / subgroupExclusiveAdd / subgroupElect / subgroupBroadcast with SUBGROUP_SIZE 32 and 64).  The reference itself ships no
    vectors and cannot be built here; two restatements agreeing table for table is what can be had."""
    from tests.golden import glsl_emulation as glsl
    g = np.load(path)
    n, B, seed, tbz, W = (int(x) for x in g["meta"])
    assert glsl.workgroup_count(n, B) == W == oracle.workgroup_count(n, B)
    cur = g["keys"]
    for i, (hist, offsets, out) in enumerate(glsl.multi_radixsort(g["keys"], B, subgroup_size)):
        hist, offsets = hist.ravel(), offsets.ravel()  # (the fixtures and the oracle keep the [W][256] tables flat)
        assert np.array_equal(hist, g[f"hist{i}"]) and np.array_equal(hist, oracle.histograms(cur, 8 * i, W, B))
        assert np.array_equal(offsets, g[f"offsets{i}"]) and np.array_equal(offsets, oracle.offsets(hist, W))
        assert np.array_equal(out, g[f"pass{i}"]) and np.array_equal(out, oracle.scatter(cur, hist, 8 * i, W, B))
        cur = out
    assert np.array_equal(cur, g["sorted"])


@pytest.mark.parametrize("n,B", [(1, 1), (255, 1), (256, 1), (257, 2), (513, 1), (3000, 5)])
def test_the_literal_emulation_on_ragged_sizes_and_ties(oracle, n, B):
    """edge sizes the fixtures do not hold, and keys with many ties (a digit that fills whole rounds: the offset advances by the count)"""
    from tests.golden import glsl_emulation as glsl
    keys = np.random.RandomState(n * 7 + B).randint(0, 2 ** 32, size=n, dtype=np.uint32) & np.uint32(0x0F0F00FF)
    W = oracle.workgroup_count(n, B)
    cur = keys
    for i, (hist, offsets, out) in enumerate(glsl.multi_radixsort(keys, B, 64 if n % 2 else 32)):
        hist, offsets = hist.ravel(), offsets.ravel()
        assert np.array_equal(hist, oracle.histograms(cur, 8 * i, W, B)) and np.array_equal(offsets, oracle.offsets(hist, W))
        assert np.array_equal(out, oracle.scatter(cur, hist, 8 * i, W, B))
        cur = out
    assert np.array_equal(cur, np.sort(keys))


@pytest.mark.parametrize("n,B", [(0, 1), (1, 1), (255, 1), (256, 1), (257, 1), (1000, 32), (1000, 1), (65536, 4),
                                 (100003, 7), (8192, 32), (8193, 32), (50000, 4096)])
def test_oracle_equals_std_sort(oracle, n, B):
    keys = np.random.RandomState(n + B).randint(0, 2 ** 32, size=n, dtype=np.uint32)
    ref, _ = oracle.std_sort(keys)
    assert np.array_equal(ref, np.sort(keys))
    if n:
        assert np.array_equal(oracle.multi_radixsort(keys, B), ref)
        assert np.array_equal(oracle.single_radixsort(keys), ref)


def test_oracle_each_pass_is_stable_counting_sort(oracle):
    n, B = 30011, 5
    keys = np.random.RandomState(5).randint(0, 2 ** 32, size=n, dtype=np.uint32)
    W = oracle.workgroup_count(n, B)
    for shift in (0, 8, 16, 24):
        hist = oracle.histograms(keys, shift, W, B)
        out = oracle.scatter(keys, hist, shift, W, B)
        digits = (keys >> np.uint32(shift)) & np.uint32(255)
        assert np.array_equal(out, keys[np.argsort(digits, kind="stable")])


def test_oracle_pairs_equal_stable_sort(oracle):
    n, B = 20000, 3
    keys = np.random.RandomState(11).randint(0, 64, size=n, dtype=np.uint32) * np.uint32(0x01010101)  # many duplicates
    vals = np.arange(n, dtype=np.uint32)
    k, v = oracle.multi_radixsort(keys, B, vals)
    rk, rv, _ = oracle.stable_sort_pairs(keys, vals)
    assert np.array_equal(k, rk) and np.array_equal(v, rv)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(v, vals[order])


def test_test_sort_reports_first_mismatch(oracle):
    a = np.arange(10, dtype=np.uint32)
    b = a.copy()
    b[7] = 99
    assert oracle.test_sort(a, a.copy()) == -1
    assert oracle.test_sort(a, b) == 7
    assert oracle.test_sort(a, a[:9].copy()) == -2


@pytest.mark.parametrize("n,B", [(1, 1), (257, 1), (1000, 32), (100003, 7), (70000, 64)])
def test_oracle_u64_equals_std_sort(oracle, n, B):
    rs = np.random.RandomState(n + B)
    keys = (rs.randint(0, 2 ** 32, n, dtype=np.uint64) << np.uint64(32)) | rs.randint(0, 2 ** 32, n, dtype=np.uint64)
    ref, _ = oracle.std_sort_u64(keys)
    assert np.array_equal(ref, np.sort(keys))
    assert np.array_equal(oracle.multi_radixsort_u64(keys, B), ref)
    # the reference's own 64-bit key range [0, 0x0FFFFFFFFFFF] (MultiRadixSort.cpp:128)
    k44 = keys & np.uint64(0x0FFFFFFFFFFF)
    assert np.array_equal(oracle.multi_radixsort_u64(k44, B), np.sort(k44))
    vals = np.arange(n, dtype=np.uint32)
    dup = keys & np.uint64(0xFF00FF00FF)
    k, v = oracle.multi_radixsort_u64(dup, B, vals)
    order = np.argsort(dup, kind="stable")
    assert np.array_equal(k, dup[order]) and np.array_equal(v, vals[order])
