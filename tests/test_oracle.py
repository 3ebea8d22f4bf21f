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
