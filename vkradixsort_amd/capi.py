"""ctypes binding of the C ABI in include/vkradixsort_amd.h (the ONLY way Python reaches the GPU path).

There is no fallback: if libvkradixsort_amd.so has not been built, or no HIP device is present, the
calls raise.  Nothing here imports or knows about oracle/.
"""
from __future__ import annotations

import ctypes
from ctypes import (POINTER, Structure, byref, c_char_p, c_double, c_int, c_size_t, c_uint32, c_uint64,
                    c_void_p)
from pathlib import Path

import os as _os

# VRS_LIB: another build of the SAME library (the sanitizer builds: vkradixsort_amd/_build/san_asan/libvkradixsort_amd.so under a preloaded
# runtime, tools/asan_fuzz.sh) -- never another implementation: there is none
LIB_PATH = Path(_os.environ["VRS_LIB"]).resolve() if _os.environ.get("VRS_LIB") else Path(__file__).resolve().parent / "libvkradixsort_amd.so"

VRS_OK = 0
VRS_ERROR_INVALID_ARGUMENT = 1
VRS_ERROR_HIP = 2
VRS_ERROR_NO_DEVICE = 3
VRS_ERROR_OUT_OF_MEMORY = 4
VRS_ERROR_UNBALANCED = 5
VRS_ERROR_TIMEOUT = 6
VRS_ERROR_PEER = 7

VRS_KERNEL_HISTOGRAM = 0
VRS_KERNEL_PREFIX = 1
VRS_KERNEL_SCATTER = 2
VRS_KERNEL_SINGLE = 3
VRS_KERNEL_DIGIT_TABLES = 4
VRS_KERNEL_LOOKBACK_SCATTER = 5
VRS_KERNEL_LOCAL_SORT = 6
VRS_KERNEL_POOL_SAMPLE = 7
VRS_KERNEL_POOL_PASS_A = 8
VRS_KERNEL_POOL_PASS_B = 9
VRS_KERNEL_COUNT = 10
KERNEL_NAMES = {0: "histogram", 1: "prefix", 2: "scatter", 3: "single", 4: "digit_tables", 5: "lookback_scatter",
                6: "local_sort", 7: "pool_sample", 8: "pool_pass_a", 9: "pool_pass_b"}

VRS_KEYS_INT32 = 0
VRS_KEYS_FLOAT32_TO_SORTABLE = 1
VRS_KEYS_SORTABLE_TO_FLOAT32 = 2

VRS_TUNE_XCD_REMAP = 0
VRS_TUNE_SCATTER_VARIANT = 1
VRS_TUNE_FUSED_PREFIX = 2
VRS_TUNE_RANK_MODE = 3
VRS_TUNE_ONE_CALL_MIN_KEYS = 4
ONE_CALL_MIN_KEYS_DEFAULT = 1 << 13  # the library's default for VRS_TUNE_ONE_CALL_MIN_KEYS
VRS_TUNE_DEBUG_MISPLACE_STREAMS = 5
VRS_TUNE_LOOKBACK_SPIN_BUDGET = 6
VRS_TUNE_DEBUG_HOLD_TILE = 7
VRS_TUNE_DIGIT_TABLE_GROUPS = 8
VRS_TUNE_SINGLE_MAX_KEYS = 9
VRS_TUNE_FUSED_PLAN = 10
VRS_TUNE_HYBRID = 11
VRS_TUNE_HYBRID_MIN_KEYS = 12
VRS_TUNE_HYBRID_FAST_COUNT = 13
VRS_TUNE_ASYNC_SORT = 14
VRS_TUNE_PLAN_WAIT_MS = 15
VRS_TUNE_MSD_RESERVE = 16
VRS_TUNE_MSD_POOL = 17
VRS_TUNE_MSD_POOL_MIN_KEYS = 18
VRS_TUNE_DEBUG_XCC_STRAY_BLOCK = 19
VRS_TUNE_MSD_POOL_SUB_BITS = 20
VRS_TUNE_DEBUG_XCC_ROTATE = 21
VRS_TUNE_MSD_POOL_REUSE_LAYOUT = 22
VRS_TUNE_MSD_POOL_PAIRS = 23
VRS_TUNE_MSD_POOL_TOP_BITS = 24
VRS_TUNE_DEBUG_POOL_NO_MEMORY = 25
VRS_TUNE_MSD_POOL_PAIRS_PACKED = 26
FORM_NAMES = {0: "none", 1: "single", 2: "contract", 3: "lsd", 4: "counted", 5: "pool"}
FORM_KNOBS = ["single_max_keys", "one_call_min_keys", "hybrid_min_keys", "pool_min_keys", "hybrid", "pool", "pool_pairs", "reserve", "groups", "xcc_map_valid",
              "atomic_rank", "pool_skip", "pool_skip_n", "wide_refused", "wide_skipped", "no_pool", "no_hybrid"]
# keys the local sort of one top-14-bit bucket can hold (msd_local_capacity): uint32 keys with the 256- / 512-thread workgroup, pairs and 64-bit keys
LOCAL_SORT_SMALL_KEYS, LOCAL_SORT_MAX_KEYS = 7165, 14333
LOCAL_SORT_SMALL_PAIRS, LOCAL_SORT_MAX_PAIRS = 6656, 13312  # pairs and 64-bit keys: 512 / 1024-thread workgroups
HYBRID_MIN_KEYS_DEFAULT = 0  # vrs_capi.hip: os_hybrid_min_keys == 0: the measured crossovers (1.3e7 keys, 2.5e7 pairs, 2e7 64-bit keys)


class PushConstants(Structure):
    """vrs_push_constants == MultiRadixSortPass::PushConstants(Histograms), 16 bytes std430."""
    _fields_ = [("g_num_elements", c_uint32), ("g_shift", c_uint32), ("g_num_workgroups", c_uint32),
                ("g_num_blocks_per_workgroup", c_uint32)]


class VrsError(RuntimeError):
    """Every failure of the reference is a std::runtime_error; this is its Python spelling."""

    def __init__(self, code: int, message: str):
        super().__init__(f"vkradixsort_amd error {code}: {message}")
        self.code = code
        self.message = message


class DistTransport(ctypes.Structure):
    """vrs_dist_transport: the wire of the multi-GPU step as a table of functions (include/vkradixsort_amd.h)."""
    _fields_ = [("user", c_void_p), ("all_gather", c_void_p), ("all_reduce", c_void_p), ("group_start", c_void_p),
                ("send", c_void_p), ("recv", c_void_p), ("group_end", c_void_p), ("error_string", c_void_p)]


MSD_COUNT_WORDS = 16384 + 8 * 256 + 64
MSD_SHIFT_WORD = 16384 + 8 * 256

# every symbol include/vkradixsort_amd.h declares: (name, restype, argtypes)
_SIGNATURES = [
    ("vrs_version", c_char_p, []),
    ("vrs_device_count", c_int, [POINTER(c_int)]),
    ("vrs_context_create", c_int, [c_int, POINTER(c_void_p)]),
    ("vrs_context_create_on_stream", c_int, [c_int, c_void_p, POINTER(c_void_p)]),
    ("vrs_context_destroy", c_int, [c_void_p]),
    ("vrs_last_error", c_char_p, [c_void_p]),
    ("vrs_context_stream", c_void_p, [c_void_p]),
    ("vrs_device_info", c_int, [c_void_p, c_char_p, c_size_t, POINTER(c_int), POINTER(c_uint64)]),
    ("vrs_buffer_create", c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
    ("vrs_buffer_wrap", c_int, [c_void_p, c_void_p, c_size_t, POINTER(c_void_p)]),
    ("vrs_buffer_release", c_int, [c_void_p]),
    ("vrs_buffer_upload", c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    ("vrs_buffer_download", c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    ("vrs_buffer_copy", c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    ("vrs_buffer_device_ptr", c_void_p, [c_void_p]),
    ("vrs_buffer_size_bytes", c_size_t, [c_void_p]),
    ("vrs_global_invocation_size", c_uint32, [c_uint32, c_uint32]),
    ("vrs_workgroup_count", c_uint32, [c_uint32, c_uint32]),
    ("vrs_multi_radixsort_histograms", c_int, [c_void_p, c_void_p, c_void_p, POINTER(PushConstants)]),
    ("vrs_multi_radixsort", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(PushConstants)]),
    ("vrs_multi_radixsort_pairs", c_int,
     [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(PushConstants)]),
    ("vrs_multi_radixsort_histograms_u64", c_int, [c_void_p, c_void_p, c_void_p, POINTER(PushConstants)]),
    ("vrs_multi_radixsort_u64", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(PushConstants)]),
    ("vrs_multi_radixsort_pairs_u64", c_int,
     [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(PushConstants)]),
    ("vrs_range_partition", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_uint32, c_uint32]),
    ("vrs_multi_radixsort_digit_offsets", c_int, [c_void_p, c_void_p]),
    ("vrs_multi_radixsort_digit_offsets_device", c_int, [c_void_p, c_void_p]),
    ("vrs_queue_wait_idle", c_int, [c_void_p]),
    ("vrs_single_radixsort", c_int, [c_void_p, c_void_p, c_void_p, c_uint32]),
    ("vrs_sort_keys_u32", c_int, [c_void_p, c_void_p, c_void_p, c_uint32]),
    ("vrs_sort_keys_u32_ranged", c_int, [c_void_p, c_void_p, c_void_p, c_uint32, c_uint32]),
    ("vrs_sort_settle", c_int, [c_void_p]),
    ("vrs_sort_pending", c_int, [c_void_p]),
    ("vrs_sort_pairs_u32", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint32]),
    ("vrs_sort_pairs_u64", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint32]),
    ("vrs_sort_keys_u64", c_int, [c_void_p, c_void_p, c_void_p, c_uint32]),
    ("vrs_transform_keys", c_int, [c_void_p, c_void_p, c_uint32, c_int]),
    ("vrs_profile_enable", c_int, [c_void_p, c_int]),
    ("vrs_profile_enable_mask", c_int, [c_void_p, c_uint32]),
    ("vrs_verify_keys_u32", c_int, [c_void_p, c_void_p, c_uint32, POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint64)]),
    ("vrs_dist_create", c_int, [c_void_p, c_void_p, c_int, c_int, c_uint32, c_int, POINTER(c_void_p)]),
    ("vrs_dist_create_with_transport", c_int, [c_void_p, c_void_p, c_int, c_int, c_uint32, c_int, POINTER(c_void_p)]),
    ("vrs_dist_destroy", c_int, [c_void_p]),
    ("vrs_dist_stats", c_int, [c_void_p, POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint64)]),
    ("vrs_dist_plan_sampled_splitters", c_int, [POINTER(c_uint32), POINTER(c_uint64), c_int, c_uint32, c_int, POINTER(c_uint32)]),
    ("vrs_dist_grouped_rounds", c_int, [c_void_p, POINTER(c_uint64)]),
    ("vrs_dist_splitter_steps", c_int, [c_void_p, POINTER(c_uint64)]),
    ("vrs_dist_loopback_create", c_int, [c_int, POINTER(c_void_p)]),
    ("vrs_dist_loopback_create_host", c_int, [c_int, POINTER(c_void_p)]),
    ("vrs_dist_loopback_set_wire", c_int, [c_void_p, c_double, c_double]),
    ("vrs_dist_loopback_transport", c_int, [c_void_p, c_int, c_void_p]),
    ("vrs_dist_loopback_destroy", c_int, [c_void_p]),
    ("vrs_msd_partition_u32", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_uint32]),
    ("vrs_msd_finish_u32", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_uint32, c_uint32]),
    ("vrs_multi_radixsort_offsets_hook", c_int, [c_void_p, c_void_p, c_void_p]),
    ("vrs_msd_partition_signal_u32", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_uint32, c_void_p]),
    ("vrs_msd_finish_grouped_u32", c_int, [c_void_p, c_void_p, c_void_p, c_uint32, c_uint32, c_uint32]),
    ("vrs_msd_finish_grouped_counts_u32", c_int, [c_void_p, c_void_p, c_void_p, c_uint32, c_uint32, c_uint32, POINTER(c_uint32)]),
    ("vrs_msd_finish_grouped_split_u32", c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, c_uint32, c_uint32, c_uint32, POINTER(c_uint32), POINTER(c_uint32)]),
    ("vrs_msd_finish_status", c_int, [c_void_p, POINTER(c_int)]),
    ("vrs_msd_finish_ticket", c_int, [c_void_p, POINTER(c_uint32)]),
    ("vrs_msd_finish_status_at", c_int, [c_void_p, c_uint32, POINTER(c_int)]),
    ("vrs_context_device", c_int, [c_void_p]),
    ("vrs_dist_sort_keys_u32", c_int, [c_void_p, c_void_p, c_uint32, POINTER(c_void_p), POINTER(c_uint32)]),
    ("vrs_dist_plan_splitters", c_int, [POINTER(c_uint64), c_int, POINTER(c_uint32)]),
    ("vrs_dist_last_error", c_char_p, [c_void_p]),
    ("vrs_profile_reset", c_int, [c_void_p]),
    ("vrs_profile_query", c_int, [c_void_p, c_int, POINTER(c_uint64), POINTER(c_double)]),
    ("vrs_profile_query_launch", c_int, [c_void_p, c_int, c_uint64, POINTER(c_double)]),
    ("vrs_one_call_stats", c_int, [c_void_p, POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint64)]),
    ("vrs_one_call_relaunched_passes", c_int, [c_void_p, POINTER(c_uint64)]),
    ("vrs_one_call_hybrid_sorts", c_int, [c_void_p, POINTER(c_uint64)]),
    ("vrs_one_call_hybrid_recounts", c_int, [c_void_p, POINTER(c_uint64)]),
    ("vrs_one_call_pool_sorts", c_int, [c_void_p, POINTER(c_uint64), POINTER(c_uint64)]),
    ("vrs_one_call_pool_retries", c_int, [c_void_p, POINTER(c_uint64)]),
    ("vrs_one_call_pool_layouts", c_int, [c_void_p, POINTER(c_uint64), POINTER(c_uint64)]),
    ("vrs_pool_form_shape", c_int, [c_uint32, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint64)]),
    ("vrs_sort_form_for", c_int, [c_uint32, c_int, c_int, POINTER(ctypes.c_int64), c_int, POINTER(c_int), POINTER(ctypes.c_int64)]),
    ("vrs_pool_form_shape_ex", c_int, [c_uint32, c_int, c_int, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint64)]),
    ("vrs_context_trim_scratch", c_int, [c_void_p, POINTER(c_uint64)]),
    ("vrs_one_call_pool_no_memory", c_int, [c_void_p, POINTER(c_uint64)]),
    ("vrs_debug_xcc_placement", c_int, [c_void_p, POINTER(c_uint64), POINTER(c_uint64), POINTER(c_int)]),
    ("vrs_debug_download_offsets", c_int, [c_void_p, c_void_p, c_size_t]),
    ("vrs_debug_atomic_rank_selftest", c_int, [c_void_p, c_uint32, c_uint32, POINTER(c_uint64)]),
    ("vrs_rank_mode", c_int, [c_void_p]),
    ("vrs_set_tuning", c_int, [c_void_p, c_int, c_int]),
]

EXPORTED_SYMBOLS = [s[0] for s in _SIGNATURES]

_lib = None


def _preload_torch_hip_runtime() -> None:
    """One HIP runtime per process.  The PyTorch-ROCm wheel bundles its own libamdhip64.so (same SONAME as
    /opt/rocm's).  If this library pulled in the system copy first and torch were imported afterwards, the
    process would hold two runtimes and torch would report "No HIP GPUs are available".  So when torch is
    installed, map ITS runtime first (without importing torch); the SONAME match makes our library bind to it."""
    import importlib.util
    import os
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load_library() -> ctypes.CDLL:
    """dlopen the in-tree library and type every entry point.  Raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback.")
    _preload_torch_hip_runtime()
    lib = ctypes.CDLL(str(LIB_PATH))
    for name, restype, argtypes in _SIGNATURES:
        fn = getattr(lib, name)  # AttributeError if the header and the library ever diverge
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(ctx_handle, rc: int) -> None:
    if rc != VRS_OK:
        msg = load_library().vrs_last_error(ctx_handle)
        raise VrsError(rc, msg.decode() if msg else "unknown error")


def device_count() -> int:
    n = c_int(0)
    rc = load_library().vrs_device_count(byref(n))
    return n.value if rc == VRS_OK else 0
