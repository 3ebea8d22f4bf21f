"""CPU tests of the boundary: the C-ABI library loads, exports every symbol the header declares, does its
host-side arithmetic like the reference, and fails LOUDLY (no fallback) when there is no GPU."""
import ctypes
import re
from pathlib import Path

import pytest

import vkradixsort_amd as vrs
from vkradixsort_amd import capi

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "vkradixsort_amd.h"


@pytest.fixture(scope="module")
def lib():
    from vkradixsort_amd import build
    build.build_library()  # hipcc cross-compiles gfx950 without a GPU
    return capi.load_library()


def declared_symbols():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(vrs_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_all_exported_and_bound(lib):
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in the header but not exported"
    assert sorted(capi.EXPORTED_SYMBOLS) == syms, "ctypes binding and header diverge"


def test_push_constants_layout_is_16_bytes_std430():
    # MultiRadixSortPass.h:17-22 / :26-31 -- four uint32 in this order
    assert ctypes.sizeof(capi.PushConstants) == 16
    assert [f[0] for f in capi.PushConstants._fields_] == ["g_num_elements", "g_shift", "g_num_workgroups",
                                                           "g_num_blocks_per_workgroup"]
    assert capi.PushConstants.g_shift.offset == 4 and capi.PushConstants.g_num_blocks_per_workgroup.offset == 12


@pytest.mark.parametrize("n,B,W", [(10 ** 6, 32, 123), (10 ** 6, 1, 3907), (10 ** 6, 4096, 1), (10 ** 7, 32, 1221),
                                   (10 ** 7, 512, 77), (10 ** 8, 32, 12208), (10 ** 8, 128, 3052), (10 ** 8, 4096, 96),
                                   (1000, 32, 1), (257, 1, 2), (1, 1, 1)])
def test_workgroup_count(lib, oracle, n, B, W):
    assert lib.vrs_workgroup_count(n, B) == W == oracle.workgroup_count(n, B)
    assert lib.vrs_global_invocation_size(n, B) == -(-n // B)


def test_compute_pass_launch_shape_matches_c_abi(lib):
    p = vrs.ComputePass.__new__(vrs.ComputePass)
    for n, B in [(10 ** 6, 32), (10 ** 8, 4096), (1000, 7), (255, 1)]:
        gis = n // B + (1 if n % B else 0)
        d = vrs.ComputePass.getDispatchSize(gis, 1, 1, vrs.Extent3D(256, 1, 1))
        assert (d.width, d.height, d.depth) == (lib.vrs_workgroup_count(n, B), 1, 1)


def test_version_string(lib):
    assert b"gfx950" in lib.vrs_version()


def test_no_device_fails_loudly(lib):
    if capi.device_count() > 0:
        pytest.skip("a GPU is present; the no-device path is exercised on the CPU box")
    ctx = vrs.GPUContext(0)
    with pytest.raises(vrs.VrsError) as e:
        ctx.init()
    assert e.value.code == capi.VRS_ERROR_NO_DEVICE
    # and the high-level entry point does not quietly sort on the CPU
    with pytest.raises(vrs.VrsError):
        vrs.MultiRadixSort(1000, quiet=True).execute(vrs.GPUContext(0).__enter__())


def test_null_arguments_are_rejected(lib):
    assert lib.vrs_context_create(0, None) == capi.VRS_ERROR_INVALID_ARGUMENT
    assert lib.vrs_queue_wait_idle(None) == capi.VRS_ERROR_INVALID_ARGUMENT
    assert lib.vrs_multi_radixsort_histograms(None, None, None, None) == capi.VRS_ERROR_INVALID_ARGUMENT
    assert lib.vrs_buffer_release(None) == capi.VRS_OK  # idempotent like Buffer::release


def test_product_package_never_touches_the_oracle():
    for path in list((ROOT / "vkradixsort_amd").rglob("*.py")) + list((ROOT / "vkradixsort_amd").rglob("*.hip")) + \
            list((ROOT / "vkradixsort_amd").rglob("*.cpp")) + list((ROOT / "vkradixsort_amd").rglob("*.h")):
        text = path.read_text()
        assert "vrs_oracle" not in text and "libvrs_oracle" not in text and "_oracle" not in text, path


class _FakeContext:
    """just enough GPUContext for the binding-table logic (no device calls)"""

    def __init__(self):
        self.m_activeIndex = 0

    def getMultiBufferedCount(self):
        return 2

    def getActiveIndex(self):
        return self.m_activeIndex

    def incrementActiveIndex(self):
        self.m_activeIndex = (self.m_activeIndex + 1) % 2


def test_ping_pong_binding_table_matches_reference():
    """MultiRadixSort.cpp:33-46 + README.md:205-209: iterations 0,2 read buffer0 / write buffer1,
    iterations 1,3 the reverse; the histogram buffer is bound in both copies."""
    ctx = _FakeContext()
    m = vrs.MultiRadixSort(1000, quiet=True)
    m.m_gpuContext = ctx
    m.m_pass = vrs.MultiRadixSortPass(ctx)
    m.m_pass.create()
    m.m_buffers = ["buf0", "buf1", "hist"]
    m.bindBuffers()
    H, R = vrs.MultiRadixSortPass.RADIX_SORT_HISTOGRAMS, vrs.MultiRadixSortPass.RADIX_SORT
    seen = []
    for i in range(4):
        seen.append((m.m_pass._bound(H, 0), m.m_pass._bound(R, 0), m.m_pass._bound(R, 1), m.m_pass._bound(H, 1),
                     m.m_pass._bound(R, 2)))
        ctx.incrementActiveIndex()
    assert seen[0] == ("buf0", "buf0", "buf1", "hist", "hist") == seen[2]
    assert seen[1] == ("buf1", "buf1", "buf0", "hist", "hist") == seen[3]
    with pytest.raises(vrs.VrsError):
        m.m_pass._bound(R, 7)


def test_cpp_host_mirror_builds_and_fails_loudly_without_gpu():
    """g++ builds the engine:: classes against the C ABI; without a device main() reports and returns 1
    (the reference's catch-all: MultiRadixSortExample.cpp:20-23)."""
    import subprocess
    from vkradixsort_amd import build
    lib, exes = build.build_host()
    assert lib.exists() and all(e.exists() for e in exes)
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    for exe in exes:
        p = subprocess.run([str(exe), "1000"], capture_output=True, text=True, timeout=60)
        assert p.returncode == 1
        assert "no HIP device" in p.stderr


def test_cpp_host_logic_unit_test_binary():
    """vkradixsort_amd/host/test/host_logic_test.cpp: launch shapes, the mt19937 key generator, std::sort timing,
    testSort's "TEST FAILED." behaviour, push-constant layout, activeIndex toggling -- all without a device."""
    import subprocess
    from vkradixsort_amd import build
    exe = build.build_host_logic_test()
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "host logic ok" in p.stdout, p.stdout + p.stderr


def test_dist_plan_splitters_matches_the_python_orchestration():
    """vrs_dist_plan_splitters (C ABI, host only) and distributed.plan_splitters (Python) derive every rank's key
    ranges from the same gathered table: they must agree boundary for boundary."""
    import ctypes

    import numpy as np

    from vkradixsort_amd import capi
    from vkradixsort_amd.distributed import plan_splitters
    lib = capi.load_library()
    rs = np.random.RandomState(9)
    cases = [rs.randint(0, 10 ** 6, 256), np.zeros(256, np.int64), np.full(256, 7), np.eye(1, 256, 200, dtype=np.int64)[0] * 12345,
             (rs.rand(256) < 0.05) * rs.randint(1, 10 ** 7, 256), np.arange(256) ** 3, rs.randint(0, 3, 256)]
    for counts in cases:
        counts = np.asarray(counts, dtype=np.uint64)
        for parts in (1, 2, 3, 8, 16, 32, 64, 256):
            out = np.zeros(parts + 1, dtype=np.uint32)
            rc = lib.vrs_dist_plan_splitters(counts.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), parts,
                                             out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
            assert rc == 0
            assert np.array_equal(out.astype(np.int64), plan_splitters(counts.astype(np.int64), parts)), (counts[:8], parts)


def test_dist_sampled_splitters_cut_at_weighted_quantiles():
    """vrs_dist_plan_sampled_splitters (host only): what the multi-GPU step cuts at when top bytes are too concentrated for
    byte-aligned ranges -- the weighted quantiles of the pooled samples.  Simulated end to end on the CPU: shards of very different
    sizes (one empty) with small / clustered keys, 2048 evenly spaced samples per rank, then every part must hold its share of ALL keys
    within a few per cent; massive ties give equal cut keys (the caller then finds the parts unbalanced)."""
    import ctypes

    import numpy as np

    from vkradixsort_amd import capi
    lib = capi.load_library()
    S = 2048
    rs = np.random.RandomState(4)

    def shards_of(kind, sizes):
        out = []
        for i, m in enumerate(sizes):
            k = rs.randint(0, 2 ** 32, size=m, dtype=np.uint32)
            if kind == "16bit":
                k >>= np.uint32(16)
            elif kind == "clustered":
                c = rs.rand(m) < 0.75
                k[c] = (k[c] & np.uint32(0x00FFFFFF)) | np.uint32(0x40000000)
            elif kind == "per_rank_ranges":  # every rank holds another part of the key space: the weights matter
                k = (k >> np.uint32(4)) + np.uint32(i << 28)
            out.append(k)
        return out

    def cut(shards, parts):
        world = len(shards)
        samples = np.zeros(world * S, np.uint32)
        for q, k in enumerate(shards):
            if k.size:
                samples[q * S:(q + 1) * S] = k[(np.arange(S, dtype=np.uint64) * np.uint64(k.size - 1) // np.uint64(S - 1)).astype(np.int64)]
        sizes = np.array([k.size for k in shards], np.uint64)
        sp = np.zeros(max(parts - 1, 1), np.uint32)
        rc = lib.vrs_dist_plan_sampled_splitters(samples.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                                                 sizes.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), world, S, parts,
                                                 sp.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
        assert rc == 0
        return sp[:parts - 1]

    for kind, sizes, parts in [("16bit", [300000, 300000], 4), ("clustered", [400000, 50000, 0], 6), ("per_rank_ranges", [600000, 60000, 200000, 0], 8),
                               ("uniform", [100000] * 8, 32)]:
        shards = shards_of(kind, sizes)
        sp = cut(shards, parts)
        assert np.all(sp[1:] >= sp[:-1])
        allk = np.concatenate(shards)
        part_of = np.searchsorted(sp, allk, side="right")  # range = number of cut keys <= key
        counts = np.bincount(part_of, minlength=parts)
        ideal = allk.size / parts
        assert counts.max() <= 1.12 * ideal and counts.min() >= 0.88 * ideal, (kind, counts.tolist())
    # massive ties: three key values for four parts -- some cut keys coincide, some part is empty: the step reports UNBALANCED
    ties = [(rs.randint(0, 3, 200000).astype(np.uint32) * np.uint32(0x10000001)) for _ in range(2)]
    sp = cut(ties, 4)
    counts = np.bincount(np.searchsorted(sp, np.concatenate(ties), side="right"), minlength=4)
    assert counts.max() > 1.15 * (400000 / 4)
    # bad arguments
    assert lib.vrs_dist_plan_sampled_splitters(None, None, 2, S, 4, None) == capi.VRS_ERROR_INVALID_ARGUMENT


def test_round3_entry_points_reject_null_and_need_no_device(lib):
    """The enqueue-only sorts' settle / pending calls, the context accessor and the in-process transport hub of the multi-GPU
    step validate their arguments without touching a device (host-side objects only)."""
    assert lib.vrs_sort_settle(None) == capi.VRS_ERROR_INVALID_ARGUMENT
    assert lib.vrs_sort_pending(None) == 0
    assert lib.vrs_context_device(None) == -1
    took = ctypes.c_int(7)
    assert lib.vrs_msd_finish_status(None, ctypes.byref(took)) == capi.VRS_ERROR_INVALID_ARGUMENT
    ticket = ctypes.c_uint32(0)
    assert lib.vrs_msd_finish_ticket(None, ctypes.byref(ticket)) == capi.VRS_ERROR_INVALID_ARGUMENT
    assert lib.vrs_msd_finish_status_at(None, 1, ctypes.byref(took)) == capi.VRS_ERROR_INVALID_ARGUMENT
    hub = ctypes.c_void_p()
    assert lib.vrs_dist_loopback_create(0, ctypes.byref(hub)) == capi.VRS_ERROR_INVALID_ARGUMENT
    assert lib.vrs_dist_loopback_create(2, None) == capi.VRS_ERROR_INVALID_ARGUMENT
    assert lib.vrs_dist_loopback_create(2, ctypes.byref(hub)) == capi.VRS_OK and hub.value
    tr = capi.DistTransport()
    assert lib.vrs_dist_loopback_transport(hub, 2, ctypes.byref(tr)) == capi.VRS_ERROR_INVALID_ARGUMENT  # rank out of range
    assert lib.vrs_dist_loopback_destroy(hub) == capi.VRS_OK
    assert lib.vrs_dist_loopback_destroy(None) == capi.VRS_OK
    d = ctypes.c_void_p()
    assert lib.vrs_dist_create_with_transport(None, None, 0, 1, 1000, 1, ctypes.byref(d)) == capi.VRS_ERROR_INVALID_ARGUMENT
    assert b"context" in lib.vrs_dist_last_error(None)


def test_dist_transport_table_layout():
    # include/vkradixsort_amd.h: user + seven function pointers, in this order
    assert ctypes.sizeof(capi.DistTransport) == 8 * ctypes.sizeof(ctypes.c_void_p)
    assert [f[0] for f in capi.DistTransport._fields_] == ["user", "all_gather", "all_reduce", "group_start", "send", "recv",
                                                           "group_end", "error_string"]
    assert capi.MSD_COUNT_WORDS == 16384 + 2048 + 64 and capi.MSD_SHIFT_WORD == 16384 + 2048


def test_pool_form_shape_by_size(lib):
    """vrs_pool_form_shape (host only): the shape the pool form picks from the size alone -- one wave per bucket while uniform buckets
    (+ 5.5 deviations) stay below 1789 keys, then the 256-thread workgroups, seven bits in the second pass where six would need the
    512-thread one -- and the context scratch that costs (about 1.5 N slots of slack buffer)."""
    def shape(n):
        sb, cap, scratch = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint64()
        assert lib.vrs_pool_form_shape(n, ctypes.byref(sb), ctypes.byref(cap), ctypes.byref(scratch)) == 0
        return sb.value, cap.value, scratch.value
    assert shape(1000) == (0, 0, 0) and shape((1 << 22) - 1) == (0, 0, 0) and shape(300000000)[0] == 0
    assert shape(1 << 22)[:2] == (6, 1789) and shape(10 ** 7)[:2] == (6, 1789) and shape(2 * 10 ** 7)[:2] == (6, 1789)
    assert shape(4 * 10 ** 7)[:2] == (6, 4093) and shape(10 ** 8)[:2] == (6, 7165)
    assert shape(13 * 10 ** 7)[:2] == (7, 7165) and shape(2 * 10 ** 8)[:2] == (7, 7165) and shape(224000000)[0] == 7
    prev = 0
    for n in (1 << 22, 10 ** 7, 5 * 10 ** 7, 10 ** 8):  # the scratch grows with N; per key it shrinks (320 slots of floor per bucket weigh on small inputs)
        s = shape(n)[2]
        assert s > prev and 1.6 * 4 * n < s < 6.0 * 4 * n, (n, s)
        prev = s
    assert 1.65 * 4e8 < shape(10 ** 8)[2] < 1.75 * 4e8  # 1.54 N of slack buffer + 0.18 N of first-pass overflow room
    assert lib.vrs_pool_form_shape(10 ** 8, None, None, None) == 0


def test_pool_form_shape_ex_reports_the_cut_that_runs(lib):
    """vrs_pool_form_shape_ex (host only): the two passes' digits as one_read_enqueue_pool cuts them (7 + 7 by default: ADVICE r5 found
    vrs_pool_form_shape reporting '6' where the second pass takes 7 bits), pairs with their own shapes and the payloads' twins in the scratch,
    and no shape at all beyond the last size whose fullest uniform bucket fits (pairs: about 2.07e8 -- the candidate bound was wider)."""
    def ex(n, pairs=0, top=0):
        a, b, cap, scratch = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint64()
        assert lib.vrs_pool_form_shape_ex(n, pairs, top, ctypes.byref(a), ctypes.byref(b), ctypes.byref(cap), ctypes.byref(scratch)) == 0
        return a.value, b.value, cap.value, scratch.value
    assert ex(10 ** 8)[:3] == (7, 7, 7165) and ex(10 ** 8, top=8)[:3] == (8, 6, 7165) and ex(10 ** 8, top=6)[:3] == (6, 8, 7165)
    assert ex(13 * 10 ** 7)[:3] == (8, 7, 7165) and ex(10 ** 7)[:3] == (7, 7, 1789)
    assert ex(10 ** 8, pairs=1)[:3] == (7, 7, 6656) and ex(2 * 10 ** 8, pairs=1)[:3] == (8, 7, 6656)
    assert ex(1000)[:2] == (0, 0) and ex(300000000)[:2] == (0, 0)
    keys, pairs = ex(10 ** 8)[3], ex(10 ** 8, pairs=1)[3]
    assert 1.9 * keys < pairs < 2.05 * keys  # the payloads' twins of the slack buffer and the overflow room
    # the last pairs size with a shape: beyond it every sort would be refused after both passes ran
    lo, hi = 10 ** 8, 3 * 10 ** 8
    while hi - lo > 1:
        mid = (lo + hi) // 2
        lo, hi = (mid, hi) if ex(mid, pairs=1)[0] else (lo, mid)
    assert 2.0e8 < lo < 2.13e8 and ex(lo, pairs=1)[2] == 13312
    assert lib.vrs_pool_form_shape_ex(10 ** 8, 0, 5, None, None, None, None) != 0


def _form(lib, n, key_bytes=4, pairs=0, **knobs):
    """vrs_sort_form_for with named knobs (capi.FORM_KNOBS); returns (form name, [pool skip, wide skipped] afterwards)"""
    arr = (ctypes.c_int64 * len(capi.FORM_KNOBS))(*[-1] * len(capi.FORM_KNOBS))
    for name, value in knobs.items():
        arr[capi.FORM_KNOBS.index(name)] = int(value)
    form, mem = ctypes.c_int(-1), (ctypes.c_int64 * 2)()
    assert lib.vrs_sort_form_for(n, key_bytes, pairs, arr, len(capi.FORM_KNOBS), ctypes.byref(form), mem) == 0
    return capi.FORM_NAMES[form.value], list(mem)


def test_the_dispatchers_decision_table(lib):
    """Which form a one-call sort takes -- form x size x kind x settings x what an earlier refusal left behind -- through the dispatcher's
    own decision function (csrc/vrs_sort_form.hpp: sort_all_passes and one_read_enqueue ask the same one), without a device.  The reference
    has two fixed paths (single_radixsort for small inputs, multi_radixsort otherwise: README.md:18-21); here the sizes where each of the
    five forms takes over are the measured crossovers, and every one of them is pinned."""
    f = lambda *a, **k: _form(lib, *a, **k)[0]  # noqa: E731
    # ---- bare uint32 keys, a fresh context
    assert f(0) == "none" and f(1) == "single" and f(4096) == "single" and f(4097) == "contract" and f(8191) == "contract"
    assert f(8192) == "lsd" and f((1 << 22) - 1) == "lsd" and f(1 << 22) == "pool" and f(10 ** 8) == "pool" and f(224000000) == "pool"
    assert f(224000001) == "counted" and f(2 * 16384 * 14333) == "counted" and f(2 * 16384 * 14333 + 1) == "lsd"
    assert f((1 << 30) - 1) == "lsd" and f(1 << 30) == "contract" and f(4294967295) == "contract"
    # ---- settings
    assert f(10 ** 8, pool=0) == "counted" and f(12999999, pool=0) == "lsd" and f(13000000, pool=0) == "counted"
    assert f(10 ** 8, hybrid=0) == "lsd" and f(10 ** 8, atomic_rank=0) == "lsd" and f(10 ** 8, groups=16) == "lsd" and f(10 ** 8, groups=8) == "pool"
    assert f(10 ** 8, reserve=0) == "counted" and f(5 * 10 ** 6, reserve=0) == "lsd"  # (the pool form's passes reserve their output)
    assert f(10 ** 8, xcc_map_valid=0) == "contract" and f(3000, xcc_map_valid=0) == "single"
    assert f(10 ** 5, one_call_min_keys=0) == "contract" and f(5000, one_call_min_keys=4097) == "lsd" and f(100, single_max_keys=50) == "contract"
    assert f(5 * 10 ** 6, pool_min_keys=6 * 10 ** 6) == "lsd" and f(2 * 10 ** 7, pool_min_keys=3 * 10 ** 7) == "counted"
    assert f(5 * 10 ** 6, hybrid_min_keys=1 << 22, pool=0) == "counted"
    # ---- pairs (the stable pool form from the counted form's threshold on), 64-bit keys (never the pool form)
    assert f(100, pairs=1) == "contract" and f(8192, pairs=1) == "lsd" and f(24999999, pairs=1) == "lsd" and f(25 * 10 ** 6, pairs=1) == "pool"
    assert f(25 * 10 ** 6, pairs=1, pool_pairs=0) == "counted" and f(25 * 10 ** 6, pairs=1, reserve=0) == "pool"  # (pairs look back, they never reserve)
    assert f(2 * 10 ** 8, pairs=1) == "pool" and f(215 * 10 ** 6, pairs=1) == "counted" and f(2 * 16384 * 13312 + 1, pairs=1) == "lsd"
    assert f(100, 8) == "contract" and f(19999999, 8) == "lsd" and f(2 * 10 ** 7, 8) == "counted" and f(10 ** 8, 8, pairs=1) == "counted"
    assert f(2 * 16384 * 6656 + 1, 8, pairs=1) == "lsd" and f(2 * 16384 * 13312, 8) == "counted"
    # ---- a retry after a refusal: one form down
    assert f(10 ** 8, no_pool=1) == "counted" and f(5 * 10 ** 6, no_pool=1) == "lsd" and f(10 ** 8, no_pool=1, no_hybrid=1) == "lsd"
    assert f(10 ** 8, pairs=1, no_pool=1) == "counted" and f(10 ** 8, 8, no_hybrid=1) == "lsd"
    # ---- what the context remembers: after a refusal the next 15 pool candidates of that size class take the counted form ...
    form, mem = _form(lib, 10 ** 8, pool_skip=15, pool_skip_n=10 ** 8)
    assert form == "counted" and mem[0] == 14
    assert _form(lib, 10 ** 8, pool_skip=1, pool_skip_n=10 ** 8) == ("counted", [0, 0]) and f(10 ** 8, pool_skip=0, pool_skip_n=10 ** 8) == "pool"
    # ... another size class (beyond a factor of two) is another workload and starts afresh; VRS_TUNE_MSD_POOL = 2 never skips
    assert _form(lib, 4 * 10 ** 7, pool_skip=15, pool_skip_n=10 ** 8) == ("pool", [0, 0]) and _form(lib, 201 * 10 ** 6, pool_skip=15, pool_skip_n=10 ** 8)[0] == "pool"
    assert f(10 ** 8, pool=2, pool_skip=15, pool_skip_n=10 ** 8) == "pool"
    # (below the counted form's threshold a skipped pool candidate falls to the LSD passes, and the count stands)
    assert _form(lib, 5 * 10 ** 6, pool_skip=7, pool_skip_n=5 * 10 ** 6) == ("lsd", [7, 0])
    # 64-bit keys: after a refusal only every 16th sort tries the hybrid form again
    assert _form(lib, 5 * 10 ** 7, 8, wide_refused=1, wide_skipped=0) == ("lsd", [0, 1])
    assert _form(lib, 5 * 10 ** 7, 8, wide_refused=1, wide_skipped=15) == ("counted", [0, 16])
    assert _form(lib, 5 * 10 ** 7, 8, wide_refused=0, wide_skipped=3) == ("counted", [0, 3])
    # ---- arguments
    bad = ctypes.c_int()
    assert lib.vrs_sort_form_for(10, 3, 0, None, 0, ctypes.byref(bad), None) != 0 and lib.vrs_sort_form_for(10, 4, 0, None, 0, None, None) != 0
    assert lib.vrs_sort_form_for(10, 4, 0, None, 3, ctypes.byref(bad), None) != 0
    assert lib.vrs_sort_form_for(10 ** 8, 4, 0, None, 0, ctypes.byref(bad), None) == 0 and capi.FORM_NAMES[bad.value] == "pool"


def test_no_source_file_of_the_library_outgrows_its_seams():
    """round 5's vrs_capi.hip was one 2314-line file holding five sort forms; it is split along its seams now (context and buffers,
    contract stages, the one-call dispatcher and settle, the pool form's host side, the hybrid form's halves), and stays so"""
    from pathlib import Path
    csrc = Path(capi.__file__).resolve().parent / "csrc"
    sizes = {p.name: sum(1 for _ in p.open()) for p in csrc.iterdir() if p.suffix in (".hip", ".hpp", ".h")}
    assert max(sizes.values()) <= 1200, sizes
    assert {"vrs_capi.hip", "vrs_capi_contract.hip", "vrs_capi_sort.hip", "vrs_capi_pool.hip", "vrs_capi_msd.hip", "vrs_host.hpp", "vrs_sort_form.hpp"} <= set(sizes)
