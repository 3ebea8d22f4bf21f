"""vkradixsort_amd -- MI355X-native multi-block LSD radix sort behind VkRadixSort's MultiRadixSort host API.

csrc/    hand-written gfx950 HIP kernels + the C ABI (include/vkradixsort_amd.h)
host/    C++ mirror of the reference's host classes (engine::GPUContext, Buffer, MultiRadixSortPass, ...)
engine   the same interface for Python callers (tests, bench.py), a thin ctypes layer over the C ABI
"""
from .capi import PushConstants, VrsError, load_library  # noqa: F401
from .engine import (Buffer, ComputePass, Extent3D, GPUContext, MultiRadixSort, MultiRadixSortPass,  # noqa: F401
                     SingleRadixSort, SingleRadixSortPass, generateRandomNumbers)

__version__ = "0.1.0"
