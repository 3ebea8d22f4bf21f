"""ctypes access to oracle/libvrs_oracle.so for the tests (never imported by the product package)."""
from __future__ import annotations

import ctypes
import subprocess
from ctypes import POINTER, c_char_p, c_double, c_int64, c_uint32, c_uint64, c_void_p

u64p = POINTER(c_uint64)
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
LIB = ORACLE_DIR / "libvrs_oracle.so"
u32p = POINTER(c_uint32)


def _p(a):
    return None if a is None else a.ctypes.data_as(u32p)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        L.vrs_oracle_global_invocation_size.restype = c_uint32
        L.vrs_oracle_global_invocation_size.argtypes = [c_uint32, c_uint32]
        L.vrs_oracle_workgroup_count.restype = c_uint32
        L.vrs_oracle_workgroup_count.argtypes = [c_uint32, c_uint32]
        L.vrs_oracle_histograms.restype = None
        L.vrs_oracle_histograms.argtypes = [u32p, u32p, c_uint32, c_uint32, c_uint32, c_uint32]
        L.vrs_oracle_offsets.restype = None
        L.vrs_oracle_offsets.argtypes = [u32p, u32p, c_uint32]
        L.vrs_oracle_scatter.restype = None
        L.vrs_oracle_scatter.argtypes = [u32p, u32p, u32p, u32p, u32p, c_uint32, c_uint32, c_uint32, c_uint32]
        L.vrs_oracle_multi_radixsort.restype = None
        L.vrs_oracle_multi_radixsort.argtypes = [u32p, u32p, u32p, c_uint32, c_uint32]
        L.vrs_oracle_multi_radixsort_pairs.restype = None
        L.vrs_oracle_multi_radixsort_pairs.argtypes = [u32p, u32p, u32p, u32p, u32p, c_uint32, c_uint32, c_void_p, c_void_p]
        L.vrs_oracle_single_radixsort.restype = None
        L.vrs_oracle_single_radixsort.argtypes = [u32p, u32p, c_uint32]
        L.vrs_oracle_mt19937_fill.restype = None
        L.vrs_oracle_mt19937_fill.argtypes = [c_uint32, u32p, c_uint64, c_uint32]
        L.vrs_stdsort_u32.restype = c_double
        L.vrs_stdsort_u32.argtypes = [u32p, c_uint64]
        L.vrs_parallel_sort_u32.restype = c_double
        L.vrs_parallel_sort_u32.argtypes = [u32p, c_uint64]
        L.vrs_stable_sort_pairs_u32.restype = c_double
        L.vrs_stable_sort_pairs_u32.argtypes = [u32p, u32p, c_uint64]
        L.vrs_test_sort.restype = c_int64
        L.vrs_test_sort.argtypes = [u32p, c_uint64, u32p, c_uint64]
        L.vrs_oracle_histograms_u64.restype = None
        L.vrs_oracle_histograms_u64.argtypes = [u64p, u32p, c_uint32, c_uint32, c_uint32, c_uint32]
        L.vrs_oracle_scatter_u64.restype = None
        L.vrs_oracle_scatter_u64.argtypes = [u64p, u64p, u32p, u32p, u32p, c_uint32, c_uint32, c_uint32, c_uint32]
        L.vrs_oracle_multi_radixsort_u64.restype = None
        L.vrs_oracle_multi_radixsort_u64.argtypes = [u64p, u64p, u32p, u32p, u32p, c_uint32, c_uint32]
        L.vrs_stdsort_u64.restype = c_double
        L.vrs_stdsort_u64.argtypes = [u64p, c_uint64]
        L.vrs_hardware_concurrency.restype = c_uint32
        L.vrs_cpu_model.argtypes = [c_char_p, c_uint32]

    def workgroup_count(self, n, B):
        return int(self.lib.vrs_oracle_workgroup_count(n, B))

    def mt19937(self, seed, n, top_bits_zeroed=0):
        out = np.empty(n, dtype=np.uint32)
        self.lib.vrs_oracle_mt19937_fill(seed, _p(out), n, top_bits_zeroed)
        return out

    def histograms(self, keys, shift, W, B):
        hist = np.empty(W * 256, dtype=np.uint32)
        self.lib.vrs_oracle_histograms(_p(keys), _p(hist), keys.size, shift, W, B)
        return hist

    def offsets(self, hist, W):
        off = np.empty(W * 256, dtype=np.uint32)
        self.lib.vrs_oracle_offsets(_p(hist), _p(off), W)
        return off

    def scatter(self, keys, hist, shift, W, B, values=None):
        out = np.zeros_like(keys)
        vout = None if values is None else np.zeros_like(values)
        self.lib.vrs_oracle_scatter(_p(keys), _p(out), _p(values), _p(vout), _p(hist), keys.size, shift, W, B)
        return out if values is None else (out, vout)

    def multi_radixsort(self, keys, B, values=None):
        """returns the content of buffer0 after the four passes (and values buffer0 for pairs)"""
        n = keys.size
        W = self.workgroup_count(n, B) if n else 0
        b0 = keys.copy()
        b1 = np.zeros_like(b0)
        hist = np.zeros(max(W, 1) * 256, dtype=np.uint32)
        if values is None:
            self.lib.vrs_oracle_multi_radixsort(_p(b0), _p(b1), _p(hist), n, B)
            return b0
        v0 = values.copy()
        v1 = np.zeros_like(v0)
        self.lib.vrs_oracle_multi_radixsort_pairs(_p(b0), _p(b1), _p(v0), _p(v1), _p(hist), n, B, None, None)
        return b0, v0

    def single_radixsort(self, keys):
        b0 = keys.copy()
        b1 = np.zeros_like(b0)
        self.lib.vrs_oracle_single_radixsort(_p(b0), _p(b1), keys.size)
        return b0

    # ---- 64-bit keys
    def histograms_u64(self, keys, shift, W, B):
        hist = np.empty(W * 256, dtype=np.uint32)
        self.lib.vrs_oracle_histograms_u64(keys.ctypes.data_as(u64p), _p(hist), keys.size, shift, W, B)
        return hist

    def scatter_u64(self, keys, hist, shift, W, B, values=None):
        out = np.zeros_like(keys)
        vout = None if values is None else np.zeros_like(values)
        self.lib.vrs_oracle_scatter_u64(keys.ctypes.data_as(u64p), out.ctypes.data_as(u64p), _p(values), _p(vout), _p(hist),
                                        keys.size, shift, W, B)
        return out if values is None else (out, vout)

    def multi_radixsort_u64(self, keys, B, values=None):
        n = keys.size
        W = self.workgroup_count(n, B) if n else 0
        b0, b1 = keys.copy(), np.zeros_like(keys)
        hist = np.zeros(max(W, 1) * 256, dtype=np.uint32)
        v0 = None if values is None else values.copy()
        v1 = None if values is None else np.zeros_like(values)
        self.lib.vrs_oracle_multi_radixsort_u64(b0.ctypes.data_as(u64p), b1.ctypes.data_as(u64p), _p(v0), _p(v1), _p(hist), n, B)
        return b0 if values is None else (b0, v0)

    def std_sort_u64(self, keys):
        out = keys.copy()
        ms = self.lib.vrs_stdsort_u64(out.ctypes.data_as(u64p), out.size)
        return out, ms

    def std_sort(self, keys):
        """the reference's verification path: returns (sorted copy, milliseconds)"""
        out = keys.copy()
        ms = self.lib.vrs_stdsort_u32(_p(out), out.size)
        return out, ms

    def stable_sort_pairs(self, keys, values):
        k, v = keys.copy(), values.copy()
        ms = self.lib.vrs_stable_sort_pairs_u32(_p(k), _p(v), k.size)
        return k, v, ms

    def test_sort(self, reference, out):
        """-1 == "Test passed.", -2 size mismatch, else first differing index (MultiRadixSort.cpp:148-161)"""
        return int(self.lib.vrs_test_sort(_p(reference), reference.size, _p(out), out.size))

    def cpu_info(self):
        buf = ctypes.create_string_buffer(256)
        self.lib.vrs_cpu_model(buf, 256)
        return int(self.lib.vrs_hardware_concurrency()), buf.value.decode()


def load() -> Oracle:
    srcs = [ORACLE_DIR / "vrs_oracle.c", ORACLE_DIR / "vrs_stdsort.cpp"]
    if not LIB.exists() or any(s.stat().st_mtime > LIB.stat().st_mtime for s in srcs):
        subprocess.run(["make", "-C", str(ORACLE_DIR), "clean", "all"], check=True, capture_output=True)
    return Oracle(ctypes.CDLL(str(LIB)))
