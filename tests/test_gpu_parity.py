"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C ABI; the oracle is
only the checker.  Integer work: bit-exact or fail."""
import ctypes
from pathlib import Path

import numpy as np
import pytest

import vkradixsort_amd as vrs
from vkradixsort_amd import capi

pytestmark = pytest.mark.gpu
GOLDEN = sorted((Path(__file__).parent / "golden").glob("*.npz"))
S = vrs.Buffer.BufferSettings


def rand_keys(n, seed):
    return np.random.RandomState(seed).randint(0, 2 ** 32, size=n, dtype=np.uint32)


class StageRunner:
    """Drives the two stages through MultiRadixSortPass exactly like MultiRadixSort::execute, but keeps
    every intermediate so each stage can be compared with the oracle."""

    def __init__(self, ctx, keys, B, values=None):
        self.ctx, self.B, self.n = ctx, B, keys.size
        self.m = vrs.MultiRadixSort(NUM_BLOCKS_PER_WORKGROUP=B, keys=keys, values=values, quiet=True)
        self.m.setup(ctx)
        self.W = self.m.m_pass.m_pushConstants.g_num_workgroups

    def run_pass(self, i):
        p = self.m.m_pass
        p.m_pushConstantsHistogram.g_shift = 8 * i
        p.m_pushConstants.g_shift = 8 * i
        p.execute(None)
        self.ctx.incrementActiveIndex()
        self.ctx.waitIdle()
        hist = np.empty(self.W * 256, np.uint32)
        self.m.m_buffers[2].downloadWithStagingBuffer(hist)
        offsets = np.empty(self.W * 256, np.uint32)
        self.ctx.check(self.ctx.lib.vrs_debug_download_offsets(self.ctx.handle, offsets.ctypes.data_as(ctypes.c_void_p),
                                                               offsets.nbytes))
        out = np.empty(self.n, np.uint32)
        self.m.m_buffers[(i + 1) % 2].downloadWithStagingBuffer(out)
        return hist, offsets, out

    def close(self):
        self.m.releaseBuffers()
        self.m.m_pass.release()
        # leave the shared context's activeIndex where the next test expects it
        while self.ctx.getActiveIndex() != 0:
            self.ctx.incrementActiveIndex()


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: p.stem)
def test_every_stage_matches_golden(gpu_context, path):
    g = np.load(path)
    n, B, seed, tbz, W = (int(x) for x in g["meta"])
    r = StageRunner(gpu_context, g["keys"], B)
    assert r.W == W
    try:
        for i in range(4):
            hist, offsets, out = r.run_pass(i)
            assert np.array_equal(hist, g[f"hist{i}"]), f"histogram table, pass {i}"
            assert np.array_equal(offsets, g[f"offsets{i}"]), f"offset table, pass {i}"
            assert np.array_equal(out, g[f"pass{i}"]), f"scatter output, pass {i}"
        assert np.array_equal(r.m.download(), g["sorted"])  # result in buffer0
    finally:
        r.close()


@pytest.mark.parametrize("n,B", [(1, 1), (255, 1), (256, 1), (257, 1), (1000, 32), (1000, 1), (65536, 4), (100003, 7),
                                 (8192, 32), (8193, 32), (50000, 4096), (300000, 16), (300001, 8), (1 << 20, 32),
                                 (123457, 64), (99999, 33), (70000, 2)])
def test_every_stage_matches_oracle(gpu_context, oracle, n, B):
    keys = rand_keys(n, n * 31 + B)
    r = StageRunner(gpu_context, keys, B)
    assert r.W == oracle.workgroup_count(n, B)
    try:
        cur = keys
        for i in range(4):
            hist, offsets, out = r.run_pass(i)
            ohist = oracle.histograms(cur, 8 * i, r.W, B)
            assert np.array_equal(hist, ohist), f"histogram table, pass {i}"
            assert np.array_equal(offsets, oracle.offsets(ohist, r.W)), f"offset table, pass {i}"
            cur = oracle.scatter(cur, ohist, 8 * i, r.W, B)
            assert np.array_equal(out, cur), f"scatter output, pass {i}"
        ref, _ = oracle.std_sort(keys)
        assert oracle.test_sort(ref, r.m.download()) == -1
    finally:
        r.close()


def _distributions(n):
    rs = np.random.RandomState(1234)
    yield "all_equal", np.full(n, 0xDEADBEEF, np.uint32)
    yield "all_zero", np.zeros(n, np.uint32)
    yield "all_ones", np.full(n, 0xFFFFFFFF, np.uint32)  # collides with the kernel's padding key
    yield "sorted", np.sort(rs.randint(0, 2 ** 32, n, dtype=np.uint32))
    yield "reverse", np.sort(rs.randint(0, 2 ** 32, n, dtype=np.uint32))[::-1].copy()
    yield "28bit_reference_range", rs.randint(0, 2 ** 32, n, dtype=np.uint32) >> np.uint32(4)
    yield "two_values", rs.randint(0, 2, n).astype(np.uint32) * np.uint32(0x80000001)
    yield "one_bin_per_pass", (rs.randint(0, 4, n).astype(np.uint32) * np.uint32(0x01010101))
    yield "low_byte_only", rs.randint(0, 256, n).astype(np.uint32)
    yield "high_byte_only", rs.randint(0, 256, n).astype(np.uint32) << np.uint32(24)
    yield "max_at_tail", np.concatenate([rs.randint(0, 2 ** 32, n - 5, dtype=np.uint32), np.full(5, 0xFFFFFFFF, np.uint32)])


@pytest.mark.parametrize("n,B", [(200003, 32), (65536 + 100, 8)])
def test_distributions_end_to_end(gpu_context, oracle, n, B):
    for name, keys in _distributions(n):
        m = vrs.MultiRadixSort(NUM_BLOCKS_PER_WORKGROUP=B, keys=keys, quiet=True)
        m.execute(gpu_context)  # raises RuntimeError("TEST FAILED.") on mismatch like the reference
        ref, _ = oracle.std_sort(keys)
        assert oracle.test_sort(ref, m.sorted_keys) == -1, name
        assert np.array_equal(oracle.multi_radixsort(keys, B), m.sorted_keys), name


@pytest.mark.parametrize("n,B", [(1000, 32), (100003, 7), (1 << 20, 32), (333333, 16)])
def test_pairs_equal_stable_sort(gpu_context, oracle, n, B):
    keys = rand_keys(n, 77) & np.uint32(0x00FF00FF)  # plenty of equal keys -> stability is visible
    vals = np.arange(n, dtype=np.uint32)
    m = vrs.MultiRadixSort(NUM_BLOCKS_PER_WORKGROUP=B, keys=keys, values=vals, quiet=True)
    m.execute(gpu_context)
    rk, rv, _ = oracle.stable_sort_pairs(keys, vals)
    assert np.array_equal(m.sorted_keys, rk)
    assert np.array_equal(m.sorted_values, rv)
    ok, ov = oracle.multi_radixsort(keys, B, vals)
    assert np.array_equal(m.sorted_keys, ok) and np.array_equal(m.sorted_values, ov)


@pytest.mark.parametrize("n", [1, 2, 255, 256, 257, 1000, 1024, 1025, 5000, 70001])
def test_single_radixsort(gpu_context, oracle, n):
    keys = rand_keys(n, n)
    s = vrs.SingleRadixSort(keys=keys, quiet=True)
    s.execute(gpu_context)
    ref, _ = oracle.std_sort(keys)
    assert oracle.test_sort(ref, s.sorted_keys) == -1
    assert np.array_equal(oracle.single_radixsort(keys), s.sorted_keys)


def test_single_radixsort_config0_1000_keys(gpu_context, oracle):
    # BASELINE.json configs[0]: 1000 random uint32 keys, single_radixsort path, std::sort verification
    keys = oracle.mt19937(1, 1000)
    s = vrs.SingleRadixSort(keys=keys, quiet=True)
    s.execute(gpu_context)
    assert oracle.test_sort(oracle.std_sort(keys)[0], s.sorted_keys) == -1


def test_zero_elements_is_a_noop(gpu_context):
    ctx = gpu_context
    b0 = vrs.Buffer(ctx, S(0))
    b1 = vrs.Buffer(ctx, S(0))
    h = vrs.Buffer(ctx, S(1024))
    pc = vrs.PushConstants(0, 0, 0, 32)
    ctx.check(ctx.lib.vrs_multi_radixsort_histograms(ctx.handle, b0.handle, h.handle, ctypes.byref(pc)))
    ctx.check(ctx.lib.vrs_multi_radixsort(ctx.handle, b0.handle, b1.handle, h.handle, ctypes.byref(pc)))
    ctx.check(ctx.lib.vrs_single_radixsort(ctx.handle, b0.handle, b1.handle, 0))
    ctx.waitIdle()
    for b in (b0, b1, h):
        b.release()


def test_error_behaviour(gpu_context):
    ctx = gpu_context
    lib = ctx.lib
    n = 1000
    b0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), rand_keys(n, 1))
    b1 = vrs.Buffer(ctx, S(4 * n))
    small = vrs.Buffer(ctx, S(4 * (n - 1)))
    h = vrs.Buffer(ctx, S(1024))
    good = vrs.PushConstants(n, 0, 1, 32)
    bad = [vrs.PushConstants(n, 3, 1, 32),   # shift not a multiple of 8
           vrs.PushConstants(n, 32, 1, 32),  # shift out of range
           vrs.PushConstants(n, 0, 1, 0),    # B == 0
           vrs.PushConstants(n, 0, 1, 2),    # tiles do not cover N (would silently drop keys)
           vrs.PushConstants(n, 0, 3, 32)]   # more workgroups than ceil(ceil(N/B)/256)
    for pc in bad:
        assert lib.vrs_multi_radixsort_histograms(ctx.handle, b0.handle, h.handle, ctypes.byref(pc)) == capi.VRS_ERROR_INVALID_ARGUMENT
        assert lib.vrs_multi_radixsort(ctx.handle, b0.handle, b1.handle, h.handle, ctypes.byref(pc)) == capi.VRS_ERROR_INVALID_ARGUMENT
        assert lib.vrs_last_error(ctx.handle)
    # undersized / aliased buffers
    assert lib.vrs_multi_radixsort(ctx.handle, b0.handle, small.handle, h.handle, ctypes.byref(good)) == capi.VRS_ERROR_INVALID_ARGUMENT
    assert lib.vrs_multi_radixsort(ctx.handle, b0.handle, b0.handle, h.handle, ctypes.byref(good)) == capi.VRS_ERROR_INVALID_ARGUMENT
    assert lib.vrs_multi_radixsort_histograms(ctx.handle, b0.handle, small.handle, ctypes.byref(vrs.PushConstants(10 ** 6, 0, 123, 32))) == capi.VRS_ERROR_INVALID_ARGUMENT
    # released buffer -> exception through the wrapper (the reference throws std::runtime_error)
    small.release()
    small.release()  # idempotent
    with pytest.raises(vrs.VrsError):
        _ = small.handle
    # the good call still works afterwards
    ctx.check(lib.vrs_multi_radixsort_histograms(ctx.handle, b0.handle, h.handle, ctypes.byref(good)))
    ctx.waitIdle()
    for b in (b0, b1, h):
        b.release()


def test_wrapped_device_memory_own_usage(gpu_context, oracle):
    """README.md:151-241 "own usage": caller-owned device memory (here: torch tensors) sorted in place of Buffers."""
    torch = pytest.importorskip("torch")
    n, B = 500003, 32
    keys = rand_keys(n, 5)
    dev = torch.device("cuda", gpu_context.device_ordinal)
    t0 = torch.from_numpy(keys.view(np.int32)).to(dev)
    t1 = torch.empty_like(t0)
    W = gpu_context.lib.vrs_workgroup_count(n, B)
    th = torch.empty(W * 256, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    bufs = [vrs.Buffer(gpu_context, S(4 * n), device_ptr=t.data_ptr()) for t in (t0, t1)]
    hist = vrs.Buffer(gpu_context, S(W * 1024), device_ptr=th.data_ptr())
    pc = vrs.PushConstants(n, 0, W, B)
    for i in range(4):
        pc.g_shift = 8 * i
        src, dst = bufs[i % 2], bufs[(i + 1) % 2]
        gpu_context.check(gpu_context.lib.vrs_multi_radixsort_histograms(gpu_context.handle, src.handle, hist.handle, ctypes.byref(pc)))
        gpu_context.check(gpu_context.lib.vrs_multi_radixsort(gpu_context.handle, src.handle, dst.handle, hist.handle, ctypes.byref(pc)))
    gpu_context.waitIdle()
    out = t0.cpu().numpy().view(np.uint32)
    assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1
    for b in bufs + [hist]:
        b.release()


def test_xcd_remap_is_performance_only(gpu_context, oracle):
    keys = rand_keys(400000, 9)
    outs = []
    for flag in (0, 1):
        gpu_context.setTuning(capi.VRS_TUNE_XCD_REMAP, flag)
        m = vrs.MultiRadixSort(NUM_BLOCKS_PER_WORKGROUP=8, keys=keys, quiet=True)
        m.execute(gpu_context)
        outs.append(m.sorted_keys)
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(outs[0], np.sort(keys))


@pytest.mark.parametrize("n,B", [(10 ** 7, 32), (10 ** 7 + 123, 32)])
def test_config1_1e7_keys(gpu_context, oracle, n, B):
    # BASELINE.json configs[1]: 10^7 random uint32 keys, multi_radixsort (seeds 1,2,3)
    for seed in (1, 2, 3):
        keys = rand_keys(n, seed)
        m = vrs.MultiRadixSort(NUM_BLOCKS_PER_WORKGROUP=B, keys=keys, quiet=True)
        m.execute(gpu_context)
        assert oracle.test_sort(oracle.std_sort(keys)[0], m.sorted_keys) == -1


def _check_sorted_permutation(keys, out):
    # size-independent properties at the full size: sortedness + multiset equality (byte-wise digit
    # histograms of all four bytes and an order-independent checksum)
    assert out.size == keys.size
    assert np.all(out[1:] >= out[:-1]), "not sorted"
    for shift in (0, 8, 16, 24):
        a = np.bincount((keys >> np.uint32(shift)) & np.uint32(255), minlength=256)
        b = np.bincount((out >> np.uint32(shift)) & np.uint32(255), minlength=256)
        assert np.array_equal(a, b)
    assert int(keys.astype(np.uint64).sum()) == int(out.astype(np.uint64).sum())
    assert int(np.bitwise_xor.reduce(keys)) == int(np.bitwise_xor.reduce(out))


def test_config2_1e8_keys_roofline_size(gpu_context, oracle):
    # BASELINE.json configs[2]: 10^8 random uint32 keys -- bit-exact vs std::sort for seed 1
    n = 10 ** 8
    keys = rand_keys(n, 1)
    m = vrs.MultiRadixSort(NUM_BLOCKS_PER_WORKGROUP=32, keys=keys, quiet=True)
    m.setup(gpu_context)
    m.enqueueSort()
    gpu_context.waitIdle()
    out = m.download()
    m.releaseBuffers()
    m.m_pass.release()
    _check_sorted_permutation(keys, out)
    ref, _ = oracle.std_sort(keys)
    assert oracle.test_sort(ref, out) == -1


@pytest.mark.parametrize("one_call", [False, True], ids=["stages", "one_call"])
@pytest.mark.parametrize("n", [(1 << 25) - 1, 1 << 25, (1 << 25) + 8193])
def test_inputs_around_the_streaming_threshold(gpu_context, oracle, n, one_call):
    """Inputs of 128 MB and more are read with nontemporal loads (histogram stage, counting read, scatter passes; the pool form, which
    also streams its sorted output, starts at 3.2e7 keys): 2^25 uint32 keys are exactly 128 MB -- one key less, that many, a ragged
    tile more, through the stages and through the one-call sort, bit for bit against std::sort."""
    keys = rand_keys(n, n % 1000)
    m = vrs.MultiRadixSort(NUM_BLOCKS_PER_WORKGROUP=32, keys=keys, quiet=True)
    m.m_oneCallSort = one_call
    m.setup(gpu_context)
    m.enqueueSort()
    gpu_context.waitIdle()
    out = m.download()
    m.releaseBuffers()
    m.m_pass.release()
    ref, _ = oracle.std_sort(keys)
    assert oracle.test_sort(ref, out) == -1


@pytest.mark.parametrize("one_call", [False, True], ids=["stages", "one_call"])
def test_config3_1e8_pairs(gpu_context, oracle, one_call):
    # BASELINE.json configs[3]: 10^8 key+payload pairs; payload[i] = i; verified by properties:
    # keys sorted, keys[payload] reproduces the output keys (payload is a permutation that follows its key),
    # and equal keys keep increasing payloads (stability) -- which together ARE std::stable_sort's result; the one-call
    # way is also compared with the oracle's std::stable_sort bit for bit, keys and payloads.  Both ways to run the passes.
    n = 10 ** 8
    keys = rand_keys(n, 2)
    vals = np.arange(n, dtype=np.uint32)
    m = vrs.MultiRadixSort(NUM_BLOCKS_PER_WORKGROUP=32, keys=keys, values=vals, quiet=True)
    m.m_oneCallSort = one_call
    m.setup(gpu_context)
    took, refused = ctypes.c_uint64(), ctypes.c_uint64()
    gpu_context.check(gpu_context.lib.vrs_one_call_pool_sorts(gpu_context.handle, ctypes.byref(took), ctypes.byref(refused)))
    before = (took.value, refused.value)
    m.enqueueSort()
    gpu_context.waitIdle()
    gpu_context.check(gpu_context.lib.vrs_one_call_pool_sorts(gpu_context.handle, ctypes.byref(took), ctypes.byref(refused)))
    if one_call:  # with the defaults 10^8 pairs take the (stable) pool form: what the bit-for-bit compare below is a compare OF
        assert (took.value - before[0], refused.value - before[1]) == (1, 0)
    ok = m.download()
    ov = np.empty(n, np.uint32)
    m.m_valueBuffers[0].downloadWithStagingBuffer(ov)
    m.releaseBuffers()
    m.m_pass.release()
    assert np.all(ok[1:] >= ok[:-1])
    assert np.array_equal(keys[ov], ok)
    same = ok[1:] == ok[:-1]
    assert np.all(ov[1:][same] > ov[:-1][same])
    seen = np.zeros(n, np.bool_)
    seen[ov] = True
    assert seen.all()
    if one_call:
        rk, rv, _ = oracle.stable_sort_pairs(keys, vals)
        assert np.array_equal(rk, ok) and np.array_equal(rv, ov)


@pytest.mark.parametrize("mode", [1, 2], ids=["ballot", "atomic"])
def test_both_ranking_methods_are_bit_exact(gpu_context, oracle, mode):
    """RANK_MODE 1 = __ballot match-any (architecturally ordered), 2 = returning LDS atomics (lane order
    verified by the device self-test).  Both must reproduce every stage of the oracle."""
    mm = ctypes.c_uint64(1)
    gpu_context.check(gpu_context.lib.vrs_debug_atomic_rank_selftest(gpu_context.handle, 2048, 777, ctypes.byref(mm)))
    assert mm.value == 0, "this device does not serve same-address LDS atomics in lane order"
    gpu_context.setTuning(capi.VRS_TUNE_RANK_MODE, mode)
    assert gpu_context.lib.vrs_rank_mode(gpu_context.handle) == mode
    try:
        for n, B in [(300001, 32), (100003, 16), (65536, 64), (12345, 8)]:
            keys = rand_keys(n, n + mode) & np.uint32(0x0F0FFFFF)  # duplicates in the upper digits
            r = StageRunner(gpu_context, keys, B)
            try:
                cur = keys
                for i in range(4):
                    hist, offsets, out = r.run_pass(i)
                    ohist = oracle.histograms(cur, 8 * i, r.W, B)
                    cur = oracle.scatter(cur, ohist, 8 * i, r.W, B)
                    assert np.array_equal(hist, ohist) and np.array_equal(out, cur), (mode, n, B, i)
            finally:
                r.close()
        vals = np.arange(200000, dtype=np.uint32)
        keys = rand_keys(200000, 3) & np.uint32(0xFF)
        m = vrs.MultiRadixSort(NUM_BLOCKS_PER_WORKGROUP=32, keys=keys, values=vals, quiet=True)
        m.execute(gpu_context)
        rk, rv, _ = oracle.stable_sort_pairs(keys, vals)
        assert np.array_equal(m.sorted_keys, rk) and np.array_equal(m.sorted_values, rv)
    finally:
        gpu_context.setTuning(capi.VRS_TUNE_RANK_MODE, 0)


def test_cpp_host_examples_run_and_verify():
    """The C++ drop-in (engine::MultiRadixSort over the C ABI) prints the reference's lines and exits 0."""
    import subprocess
    from vkradixsort_amd import build
    _, exes = build.build_host()
    multi, single = exes
    for args in (["1000000"], ["1000", "1"], ["100003", "7", "5"], ["2000000", "32", "2", "3", "28bit"],
                 ["3000000", "32", "1", "2", "full", "32bit", "onecall"],  # m_oneCallSort: the library runs the passes
                 # BASELINE.json configs[3] through the C++ drop-in (m_sortPairs): payloads ride at bindings (1,3)/(1,4),
                 # keys and payloads verified against std::stable_sort -- stage by stage and one-call
                 ["2500003", "32", "4", "2", "28bit", "32bit", "stages", "pairs"],
                 ["2500003", "32", "5", "1", "full", "32bit", "onecall", "pairs"],
                 ["300000", "64", "6", "1", "28bit", "64bit", "stages", "pairs"]):
        p = subprocess.run([str(multi), *args], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        bits = "64" if len(args) > 5 and args[5] == "64bit" else "32"
        assert "[MultiRadixSort] Sorting " + str(int(float(args[0]))) + " " + bits + "bit numbers." in p.stdout
        assert "GPU sort finished in" in p.stdout and "CPU sort finished in" in p.stdout
        assert "[MultiRadixSort] Test passed." in p.stdout
        assert ("Payloads follow their keys (stable)." in p.stdout) == ("pairs" in args)
    p = subprocess.run([str(single), "1000"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "[SingleRadixSort] Test passed." in p.stdout, p.stdout + p.stderr


def test_range_sharded_sort_product_backend_world1_rccl(oracle):
    """The multi-GPU path end to end with the PRODUCT backend (C ABI on torch's stream) and RCCL, at world
    size 1 (one GPU per box here; world size 2 is covered on CPU with gloo in test_distributed_cpu.py)."""
    import os
    import socket
    torch = pytest.importorskip("torch")
    import torch.distributed as dist
    from vkradixsort_amd.distributed import HipLocalSortBackend, RangeShardedSort
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        n = 3000017
        keys = rand_keys(n, 4242)
        cap = int(n * 1.25) + 4096
        backend = HipLocalSortBackend(0, capacity=cap, blocks_per_workgroup=32)
        sorter = RangeShardedSort(backend, recv_capacity=cap, make_empty=lambda m: torch.empty(m, dtype=torch.int32, device=dev))
        t = torch.from_numpy(keys.view(np.int32)).to(dev)
        res = sorter.step(t, n)
        torch.cuda.synchronize()
        out = res.keys[:res.count].cpu().numpy().view(np.uint32)
        assert res.count == n and res.bounds.tolist() == [0, 256]
        assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1
        backend.close()
    finally:
        dist.destroy_process_group()


def _sort_with_transform(ctx, raw_u32, to_mode, from_mode, B=32):
    """transform -> four passes -> inverse transform, all on the device through the C ABI"""
    n = raw_u32.size
    lib = ctx.lib
    b0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), raw_u32)
    b1 = vrs.Buffer(ctx, S(4 * n))
    W = lib.vrs_workgroup_count(n, B)
    h = vrs.Buffer(ctx, S(W * 1024))
    ctx.check(lib.vrs_transform_keys(ctx.handle, b0.handle, n, to_mode))
    pc = vrs.PushConstants(n, 0, W, B)
    bufs = [b0, b1]
    for i in range(4):
        pc.g_shift = 8 * i
        ctx.check(lib.vrs_multi_radixsort_histograms(ctx.handle, bufs[i % 2].handle, h.handle, ctypes.byref(pc)))
        ctx.check(lib.vrs_multi_radixsort(ctx.handle, bufs[i % 2].handle, bufs[(i + 1) % 2].handle, h.handle, ctypes.byref(pc)))
    ctx.check(lib.vrs_transform_keys(ctx.handle, b0.handle, n, from_mode))
    out = np.empty(n, np.uint32)
    b0.downloadWithStagingBuffer(out)
    for b in (b0, b1, h):
        b.release()
    return out


@pytest.mark.parametrize("n", [1, 3, 1000, 100003, 1 << 20])
def test_int32_keys_via_sign_flip(gpu_context, n):
    vals = np.random.RandomState(n).randint(-2 ** 31, 2 ** 31, size=n, dtype=np.int64).astype(np.int32)
    out = _sort_with_transform(gpu_context, vals.view(np.uint32), capi.VRS_KEYS_INT32, capi.VRS_KEYS_INT32)
    assert np.array_equal(out.view(np.int32), np.sort(vals))


@pytest.mark.parametrize("n", [2, 1000, 100003, 1 << 20])
def test_float32_keys_total_order(gpu_context, n):
    rs = np.random.RandomState(n)
    vals = (rs.standard_normal(n) * 10.0 ** rs.randint(-30, 30, n)).astype(np.float32)
    special = np.array([0.0, -0.0, np.inf, -np.inf, 1e-45, -1e-45, 3.4e38, -3.4e38], dtype=np.float32)
    vals[:min(n, special.size)] = special[:min(n, special.size)]
    out = _sort_with_transform(gpu_context, vals.view(np.uint32), capi.VRS_KEYS_FLOAT32_TO_SORTABLE,
                               capi.VRS_KEYS_SORTABLE_TO_FLOAT32).view(np.float32)
    ref = np.sort(vals)  # no NaNs in the input: numpy's order == IEEE total order except for the sign of zero
    assert np.array_equal(out, ref)
    if n >= 2:  # -0.0 sorts before +0.0 (bit-exact total order), which numpy does not promise
        zeros = np.flatnonzero(out == 0.0)
        signs = np.signbit(out[zeros])
        assert np.all(signs[:-1] >= signs[1:])
    assert np.array_equal(np.sort(out.view(np.uint32)), np.sort(vals.view(np.uint32)))  # a permutation of the input bits


def test_float32_nans_sort_to_the_ends(gpu_context):
    vals = np.array([np.nan, 1.0, -np.nan, -1.0, np.inf, -np.inf, 0.0], dtype=np.float32)
    vals.view(np.uint32)[2] |= np.uint32(0x80000000)  # a negative NaN
    out = _sort_with_transform(gpu_context, vals.view(np.uint32), capi.VRS_KEYS_FLOAT32_TO_SORTABLE,
                               capi.VRS_KEYS_SORTABLE_TO_FLOAT32).view(np.float32)
    assert np.isnan(out[0]) and np.signbit(out[0]) and np.isnan(out[-1]) and not np.signbit(out[-1])
    assert out[1:-1].tolist() == [-np.inf, -1.0, 0.0, 1.0, np.inf]


@pytest.mark.parametrize("n", [1, 257, 100003, 3000001])
def test_one_call_sort_entry_points(gpu_context, oracle, n):
    ctx, lib = gpu_context, gpu_context.lib
    keys = rand_keys(n, n) & np.uint32(0xFFFF00FF)
    vals = np.arange(n, dtype=np.uint32)
    k0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), keys)
    k1 = vrs.Buffer(ctx, S(4 * n))
    ctx.check(lib.vrs_sort_keys_u32(ctx.handle, k0.handle, k1.handle, n))
    out = np.empty(n, np.uint32)
    k0.downloadWithStagingBuffer(out)
    assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1
    k0.release()
    k0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), keys)
    v0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), vals)
    v1 = vrs.Buffer(ctx, S(4 * n))
    ctx.check(lib.vrs_sort_pairs_u32(ctx.handle, k0.handle, k1.handle, v0.handle, v1.handle, n))
    ov = np.empty(n, np.uint32)
    k0.downloadWithStagingBuffer(out)
    v0.downloadWithStagingBuffer(ov)
    rk, rv, _ = oracle.stable_sort_pairs(keys, vals)
    assert np.array_equal(out, rk) and np.array_equal(ov, rv)
    for b in (k0, k1, v0, v1):
        b.release()


@pytest.mark.parametrize("n,B", [(300007, 64), (300007, 32), (300007, 8), (1 << 20, 256), (50001, 4096), (9000, 1)])
def test_sort_stage_accepts_a_caller_filled_histogram_table(gpu_context, oracle, n, B):
    """The RADIX_SORT stage consumes only the [W][256] buffer (multi_radixsort.comp:56-63): a caller may fill it
    without ever running the histogram stage here -- the sub-tile shortcut for large B must not be assumed."""
    ctx, lib = gpu_context, gpu_context.lib
    keys = rand_keys(n, n + B)
    W = oracle.workgroup_count(n, B)
    for shift in (0, 16):
        ohist = oracle.histograms(keys, shift, W, B)
        b0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), keys)
        b1 = vrs.Buffer(ctx, S(4 * n))
        h = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(W * 1024), ohist)
        pc = vrs.PushConstants(n, shift, W, B)
        ctx.check(lib.vrs_multi_radixsort(ctx.handle, b0.handle, b1.handle, h.handle, ctypes.byref(pc)))
        out = np.empty(n, np.uint32)
        b1.downloadWithStagingBuffer(out)
        assert np.array_equal(out, oracle.scatter(keys, ohist, shift, W, B)), (n, B, shift)
        # and a histogram stage for OTHER keys in between must not leak its cached sub-tile table
        other = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), keys[::-1].copy())
        h2 = vrs.Buffer(ctx, S(W * 1024))
        ctx.check(lib.vrs_multi_radixsort_histograms(ctx.handle, other.handle, h2.handle, ctypes.byref(pc)))
        ctx.check(lib.vrs_multi_radixsort(ctx.handle, b0.handle, b1.handle, h.handle, ctypes.byref(pc)))
        b1.downloadWithStagingBuffer(out)
        assert np.array_equal(out, oracle.scatter(keys, ohist, shift, W, B)), (n, B, shift, "after foreign histogram stage")
        for b in (b0, b1, h, other, h2):
            b.release()


@pytest.mark.parametrize("B", [64, 256])
def test_kept_sub_tile_table_never_outlives_its_keys(gpu_context, oracle, B):
    """For NUM_BLOCKS_PER_WORKGROUP > 32 the histogram stage keeps an 8192-key sub-tile table for the sort stage.  It must
    be dropped as soon as the keys it describes may have changed (upload, device copy, key transform, a one-call sort)
    or another table is bound: the sort stage then reads the table it is GIVEN, as the reference's does
    (multi_radixsort.comp:56-63)."""
    ctx, lib = gpu_context, gpu_context.lib
    n = 400003
    a, b = rand_keys(n, 11), rand_keys(n, 12)
    W = oracle.workgroup_count(n, B)
    pc = vrs.PushConstants(n, 8, W, B)
    b0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), a)
    b1 = vrs.Buffer(ctx, S(4 * n))
    h = vrs.Buffer(ctx, S(W * 1024))
    out = np.empty(n, np.uint32)
    # (1) keys rewritten in place between the two stages, table refilled by the caller for the NEW keys
    ctx.check(lib.vrs_multi_radixsort_histograms(ctx.handle, b0.handle, h.handle, ctypes.byref(pc)))
    ctx.check(lib.vrs_buffer_upload(ctx.handle, b0.handle, b.ctypes.data_as(ctypes.c_void_p), b.nbytes))
    hb = oracle.histograms(b, 8, W, B)
    ctx.check(lib.vrs_buffer_upload(ctx.handle, h.handle, hb.ctypes.data_as(ctypes.c_void_p), hb.nbytes))
    ctx.check(lib.vrs_multi_radixsort(ctx.handle, b0.handle, b1.handle, h.handle, ctypes.byref(pc)))
    b1.downloadWithStagingBuffer(out)
    assert np.array_equal(out, oracle.scatter(b, hb, 8, W, B)), "stale sub-tile table after an upload"
    # (2) same keys, but the sort stage is handed ANOTHER table buffer (filled by the caller)
    ctx.check(lib.vrs_multi_radixsort_histograms(ctx.handle, b0.handle, h.handle, ctypes.byref(pc)))
    h2 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(W * 1024), hb)
    ctx.check(lib.vrs_multi_radixsort(ctx.handle, b0.handle, b1.handle, h2.handle, ctypes.byref(pc)))
    b1.downloadWithStagingBuffer(out)
    assert np.array_equal(out, oracle.scatter(b, hb, 8, W, B)), "another table bound to the sort stage"
    # (3) a device copy over the keys between the stages
    src = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), a)
    ctx.check(lib.vrs_multi_radixsort_histograms(ctx.handle, b0.handle, h.handle, ctypes.byref(pc)))  # table of b
    b0.copyFrom(src)  # keys are a again
    ha = oracle.histograms(a, 8, W, B)
    ctx.check(lib.vrs_buffer_upload(ctx.handle, h.handle, ha.ctypes.data_as(ctypes.c_void_p), ha.nbytes))
    ctx.check(lib.vrs_multi_radixsort(ctx.handle, b0.handle, b1.handle, h.handle, ctypes.byref(pc)))
    b1.downloadWithStagingBuffer(out)
    assert np.array_equal(out, oracle.scatter(a, ha, 8, W, B)), "stale sub-tile table after a device copy"
    for x in (b0, b1, h, h2, src):
        x.release()


def test_verify_keys_on_device(gpu_context):
    """vrs_verify_keys_u32, the on-device form of MultiRadixSort::verify / testSort (MultiRadixSort.cpp:97-102,148-161):
    descents == 0 iff ascending; the two fingerprints do not depend on the order and change with the multiset."""
    ctx = gpu_context
    n = 1000003
    keys = rand_keys(n, 5)
    kb = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), keys)
    d, sm, mx = kb.verifyKeys(n)
    assert d == int(np.count_nonzero(keys[:-1] > keys[1:]))
    assert sm == int(keys.astype(np.uint64).sum())
    sb = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), np.sort(keys))
    d2, sm2, mx2 = sb.verifyKeys(n)
    assert d2 == 0 and (sm2, mx2) == (sm, mx)
    tampered = np.sort(keys)
    tampered[1234] += 1  # still ascending or not, the multiset changed: the fingerprints must notice
    tampered[1235] -= 1 if tampered[1235] > 0 else 0
    tb = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), tampered)
    _, sm3, mx3 = tb.verifyKeys(n)
    assert mx3 != mx
    assert kb.verifyKeys(0) == (0, 0, 0)
    # 4-byte aligned sub-ranges of every phase and a few ragged lengths
    for off in (1, 2, 3):
        for m in (1, 2, 3, 4, 5, 1000, n - 7):
            view = vrs.Buffer(ctx, S(4 * m), device_ptr=kb.getDeviceAddress() + 4 * off)
            sub = keys[off:off + m]
            assert view.verifyKeys(m) == (int(np.count_nonzero(sub[:-1] > sub[1:])), int(sub.astype(np.uint64).sum()),
                                          view.verifyKeys(m)[2]), (off, m)
            view.release()
    for x in (kb, sb, tb):
        x.release()


@pytest.mark.parametrize("B", [1, 2, 4, 8, 16, 64, 128, 512, 4096, 16384])
def test_every_block_count_of_the_reference_sweeps(gpu_context, oracle, B):
    """NUM_BLOCKS_PER_WORKGROUP values of the reference's timing plots (README.md:253-265): tables and outputs per stage."""
    n = 700001
    keys = rand_keys(n, B)
    r = StageRunner(gpu_context, keys, B)
    try:
        cur = keys
        for i in range(4):
            hist, offsets, out = r.run_pass(i)
            ohist = oracle.histograms(cur, 8 * i, r.W, B)
            assert np.array_equal(hist, ohist), f"histogram table, pass {i}"
            assert np.array_equal(offsets, oracle.offsets(ohist, r.W)), f"offset table, pass {i}"
            cur = oracle.scatter(cur, ohist, 8 * i, r.W, B)
            assert np.array_equal(out, cur), f"scatter output, pass {i}"
    finally:
        r.close()


def rand_keys_u64(n, seed):
    rs = np.random.RandomState(seed)
    return (rs.randint(0, 2 ** 32, n, dtype=np.uint64) << np.uint64(32)) | rs.randint(0, 2 ** 32, n, dtype=np.uint64)


@pytest.mark.parametrize("n,B", [(1, 1), (255, 1), (4097, 16), (100003, 7), (300007, 32), (700001, 64), (50001, 4096), (123457, 8)])
def test_u64_every_stage_matches_oracle(gpu_context, oracle, n, B):
    """SORT_64_BIT (MultiRadixSort.h:10-18): eight passes, shifts 0..56, same [W][256] table and push constants."""
    ctx, lib = gpu_context, gpu_context.lib
    keys = rand_keys_u64(n, n + B)
    if n > 1000:
        keys[:500] &= np.uint64(0x0FFFFFFFFFFF)  # some keys in the reference's own 44-bit range
    W = oracle.workgroup_count(n, B)
    b = [vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(8 * n), keys), vrs.Buffer(ctx, S(8 * n))]
    h = vrs.Buffer(ctx, S(W * 1024))
    cur = keys
    for i in range(8):
        pc = vrs.PushConstants(n, 8 * i, W, B)
        ctx.check(lib.vrs_multi_radixsort_histograms_u64(ctx.handle, b[i % 2].handle, h.handle, ctypes.byref(pc)))
        ctx.check(lib.vrs_multi_radixsort_u64(ctx.handle, b[i % 2].handle, b[(i + 1) % 2].handle, h.handle, ctypes.byref(pc)))
        hist = np.empty(W * 256, np.uint32)
        h.downloadWithStagingBuffer(hist)
        out = np.empty(n, np.uint64)
        b[(i + 1) % 2].downloadWithStagingBuffer(out)
        ohist = oracle.histograms_u64(cur, 8 * i, W, B)
        assert np.array_equal(hist, ohist), f"histogram table, pass {i}"
        cur = oracle.scatter_u64(cur, ohist, 8 * i, W, B)
        assert np.array_equal(out, cur), f"scatter output, pass {i}"
    ref, _ = oracle.std_sort_u64(keys)
    final = np.empty(n, np.uint64)
    b[0].downloadWithStagingBuffer(final)  # eight passes: back in buffer 0
    assert np.array_equal(final, ref)
    bad = vrs.PushConstants(n, 64, W, B)
    assert lib.vrs_multi_radixsort_histograms_u64(ctx.handle, b[0].handle, h.handle, ctypes.byref(bad)) == capi.VRS_ERROR_INVALID_ARGUMENT
    for x in b + [h]:
        x.release()


@pytest.mark.parametrize("n", [1, 1000, 1000003, 5000000])
def test_u64_one_call_sort_and_pairs(gpu_context, oracle, n):
    ctx, lib = gpu_context, gpu_context.lib
    keys = rand_keys_u64(n, n)
    k0 = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(8 * n), keys)
    k1 = vrs.Buffer(ctx, S(8 * n))
    ctx.check(lib.vrs_sort_keys_u64(ctx.handle, k0.handle, k1.handle, n))
    out = np.empty(n, np.uint64)
    k0.downloadWithStagingBuffer(out)
    assert np.array_equal(out, oracle.std_sort_u64(keys)[0])
    # pairs: uint64 keys with many duplicates + uint32 payloads, stable
    dup = keys & np.uint64(0x00FF0000FF0000FF)
    vals = np.arange(n, dtype=np.uint32)
    B = 16
    W = oracle.workgroup_count(n, B)
    kb = [vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(8 * n), dup), vrs.Buffer(ctx, S(8 * n))]
    vb = [vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), vals), vrs.Buffer(ctx, S(4 * n))]
    h = vrs.Buffer(ctx, S(W * 1024))
    for i in range(8):
        pc = vrs.PushConstants(n, 8 * i, W, B)
        ctx.check(lib.vrs_multi_radixsort_histograms_u64(ctx.handle, kb[i % 2].handle, h.handle, ctypes.byref(pc)))
        ctx.check(lib.vrs_multi_radixsort_pairs_u64(ctx.handle, kb[i % 2].handle, kb[(i + 1) % 2].handle, vb[i % 2].handle,
                                                    vb[(i + 1) % 2].handle, h.handle, ctypes.byref(pc)))
    ov = np.empty(n, np.uint32)
    kb[0].downloadWithStagingBuffer(out)
    vb[0].downloadWithStagingBuffer(ov)
    order = np.argsort(dup, kind="stable")
    assert np.array_equal(out, dup[order]) and np.array_equal(ov, vals[order])
    for x in [k0, k1, h] + kb + vb:
        x.release()


def test_u64_through_the_host_mirrors(gpu_context, oracle):
    """engine.MultiRadixSort with uint64 keys (SORT_64_BIT) and the C++ MultiRadixSort64 example."""
    import subprocess
    from vkradixsort_amd import build
    n = 400003
    keys = rand_keys_u64(n, 64)
    m = vrs.MultiRadixSort(NUM_BLOCKS_PER_WORKGROUP=32, keys=keys, quiet=True)
    m.execute(gpu_context)
    assert m.sorted_keys.dtype == np.uint64 and np.array_equal(m.sorted_keys, oracle.std_sort_u64(keys)[0])
    vals = np.arange(n, dtype=np.uint32)
    dup = keys & np.uint64(0xFFFF)
    m = vrs.MultiRadixSort(NUM_BLOCKS_PER_WORKGROUP=64, keys=dup, values=vals, quiet=True)
    m.execute(gpu_context)
    order = np.argsort(dup, kind="stable")
    assert np.array_equal(m.sorted_keys, dup[order]) and np.array_equal(m.sorted_values, vals[order])
    _, exes = build.build_host()
    for args in (["1000000", "32", "1", "1", "full", "64bit"], ["300001", "7", "2", "1", "28bit", "64bit"]):
        p = subprocess.run([str(exes[0]), *args], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and "64bit numbers." in p.stdout and "[MultiRadixSort] Test passed." in p.stdout, p.stdout + p.stderr


@pytest.mark.parametrize("n", [2 ** 30 - 1, 2 ** 31 + 12345, 2 ** 32 - 1])
def test_maximum_sizes_index_arithmetic(gpu_context, n):
    """g_num_elements is a uint32 (push constant): sort up to 2^32 - 1 keys (17 GB per buffer; 2^30 - 1 is the largest
    count the one-read form of the one-call sort takes, above it come the contract passes) and check the
    size-independent properties on the device: strictly increasing (the input is a bijection image, so all keys
    are distinct) and the same order-independent fingerprint as the input."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", gpu_context.device_ordinal)
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 12 * n + (8 << 30):
        pytest.skip("not enough free HBM for the maximum-size test")
    chunks = []
    step = 1 << 28
    for a in range(0, n, step):  # keys[i] = (i * 2654435761) mod 2^32, an injective map on [0, 2^32)
        i = torch.arange(a, min(a + step, n), dtype=torch.int64, device=dev)
        chunks.append(((i * 2654435761) & 0xFFFFFFFF).to(torch.int32))  # wraps into the int32 bit pattern
        del i
    keys = torch.cat(chunks)
    del chunks
    mask = torch.tensor(0xFFFFFFFF, dtype=torch.int64, device=dev)

    def sums(t):  # (sum, sum of squares mod 2^61-ish): an order-independent fingerprint of the multiset
        s, q = 0, 0
        for a in range(0, n, step):
            u = t[a:a + step].to(torch.int64) & mask
            s += int(u.sum().item())
            q += int(((u & 0xFFFF) * (u >> 16)).sum().item())
            del u
        return s, q

    before = sums(keys)
    tmp = torch.empty_like(keys)
    torch.cuda.synchronize()
    k0 = vrs.Buffer(gpu_context, S(4 * n), device_ptr=keys.data_ptr())
    k1 = vrs.Buffer(gpu_context, S(4 * n), device_ptr=tmp.data_ptr())
    gpu_context.check(gpu_context.lib.vrs_sort_keys_u32(gpu_context.handle, k0.handle, k1.handle, n))
    gpu_context.waitIdle()
    assert sums(keys) == before
    flip = torch.tensor(-2 ** 31, dtype=torch.int32, device=dev)
    for a in range(0, n - 1, step):
        seg = keys[a:min(a + step + 1, n)] ^ flip  # unsigned order on int32 storage
        assert bool((seg[1:] > seg[:-1]).all().item()), f"not strictly increasing in [{a}, {a + step}]"
        del seg
    k0.release()
    k1.release()


@pytest.mark.parametrize("n,nsplit", [(1, 0), (1000, 1), (100003, 7), (1 << 20, 31), (3000017, 255), (500000, 3)])
def test_range_partition_by_splitters(gpu_context, n, nsplit):
    """vrs_range_partition: stable grouping by range r = #splitters <= key (the multi-GPU exchange's robust local step)."""
    ctx, lib = gpu_context, gpu_context.lib
    rs = np.random.RandomState(n + nsplit)
    keys = rs.randint(0, 2 ** 32, n, dtype=np.uint32)
    if n > 1000:
        keys[: n // 3] = rs.randint(0, 1 << 12, n // 3, dtype=np.uint32)  # heavy skew towards small keys
        keys[-5:] = 0xFFFFFFFF  # collides with the padding key
    splitters = np.sort(rs.choice(keys, size=nsplit, replace=True)).astype(np.uint32) if nsplit else np.zeros(0, np.uint32)
    if nsplit >= 7:
        splitters[2] = splitters[3]  # an empty range
    kin = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * n), keys)
    kout = vrs.Buffer(ctx, S(4 * n))
    sp = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * max(nsplit, 1)), np.concatenate([splitters, np.zeros(1, np.uint32)])[:max(nsplit, 1)])
    ctx.check(lib.vrs_range_partition(ctx.handle, kin.handle, kout.handle, sp.handle, nsplit, n))
    out = np.empty(n, np.uint32)
    kout.downloadWithStagingBuffer(out)
    base = np.empty(256, np.uint32)
    ctx.check(lib.vrs_multi_radixsort_digit_offsets(ctx.handle, base.ctypes.data_as(ctypes.c_void_p)))
    bucket = np.searchsorted(splitters, keys, side="right")
    assert np.array_equal(out, keys[np.argsort(bucket, kind="stable")])
    counts = np.bincount(bucket, minlength=256)
    assert np.array_equal(base, np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint32))
    for b in (kin, kout, sp):
        b.release()


def test_range_sharded_sort_small_keys_take_the_sampled_splitter_path(oracle):
    """Keys below 2^20 share one top byte: byte-aligned cuts are useless, the step must switch to sampled splitters
    and vrs_range_partition (product backend, RCCL, world size 1 with 4 rounds = 4 ranges)."""
    import os
    import socket
    torch = pytest.importorskip("torch")
    import torch.distributed as dist
    from vkradixsort_amd.distributed import HipLocalSortBackend, RangeShardedSort
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        n = 2000003
        keys = np.random.RandomState(7).randint(0, 2 ** 20, n, dtype=np.uint32)
        cap = int(n * 1.25) + 4096
        backend = HipLocalSortBackend(0, capacity=cap, blocks_per_workgroup=32)
        # max_imbalance < 1 forces the decision even at world size 1 (one rank always holds 100 %)
        sorter = RangeShardedSort(backend, recv_capacity=cap, make_empty=lambda m: torch.empty(m, dtype=torch.int32, device=dev),
                                  rounds=4, max_imbalance=0.5)
        res = sorter.step(torch.from_numpy(keys.view(np.int32)).to(dev), n)
        torch.cuda.synchronize()
        out = res.keys[:res.count].cpu().numpy().view(np.uint32)
        assert res.count == n and res.bounds.tolist() == [0, 4]  # part-index space: the sampled path was taken
        assert oracle.test_sort(oracle.std_sort(keys)[0], out) == -1
        backend.close()
    finally:
        dist.destroy_process_group()


def test_product_backend_handles_an_empty_shard():
    """a rank may hold no keys.  Both C-ABI stages launch nothing for N == 0, so the backend must not read
    an offset row that was never written: every top byte starts at position 0."""
    torch = pytest.importorskip("torch")
    from vkradixsort_amd.distributed import HipLocalSortBackend
    backend = HipLocalSortBackend(0, capacity=1 << 20)
    try:
        keys = torch.from_numpy(rand_keys(300000, 3).view(np.int32)).cuda()
        grouped, base = backend.group_by_top_byte(keys, 300000)  # leaves a non-trivial offset row behind
        torch.cuda.synchronize()
        assert int(base[-1].item()) > 0
        grouped, base = backend.group_by_top_byte(keys[:0], 0)
        torch.cuda.synchronize()
        assert int(base.abs().sum().item()) == 0
        g2, first = backend.partition_by_splitters(keys[:0], 0, np.array([10, 20, 30], dtype=np.uint32))
        assert first.tolist() == [0, 0, 0, 0, 0]
        assert backend.sort(keys[:0], 0).numel() == 0
    finally:
        backend.close()


def _dist_step(ctx, comm, keys, rounds, capacity=None):
    """one vrs_dist_sort_keys_u32 step at world size 1 -> (sorted range as numpy, count)"""
    lib = ctx.lib
    n = keys.size
    cap = capacity or int(n * 1.25) + 4096
    d = ctypes.c_void_p()
    assert lib.vrs_dist_create(ctx.handle, comm, 0, 1, cap, rounds, ctypes.byref(d)) == 0, lib.vrs_dist_last_error(None)
    try:
        kb = vrs.Buffer.fillDeviceWithStagingBuffer(ctx, S(4 * max(n, 1)), keys if n else np.zeros(1, np.uint32))
        out_buf, out_n = ctypes.c_void_p(), ctypes.c_uint32()
        rc = lib.vrs_dist_sort_keys_u32(d, kb.handle, n, ctypes.byref(out_buf), ctypes.byref(out_n))
        if rc:
            return rc, lib.vrs_dist_last_error(d).decode()
        ctx.waitIdle()
        out = np.empty(out_n.value, np.uint32)
        if out_n.value:
            ctx.check(lib.vrs_buffer_download(ctx.handle, out_buf, out.ctypes.data_as(ctypes.c_void_p), out.nbytes))
        kb.release()
        return 0, out
    finally:
        lib.vrs_dist_destroy(d)


@pytest.mark.parametrize("rounds", [1, 4, 8])
@pytest.mark.parametrize("n", [0, 1, 5000, (1 << 21) + 77, 6000001])
def test_dist_step_c_abi_world1(gpu_context, n, rounds):
    """The multi-GPU step behind the C ABI (vrs_dist_*), exercised end to end at world size 1 without a communicator:
    top-byte partition pass, splitters, `rounds` sub-ranges moved by device copies, one vrs_sort_keys_u32 per
    sub-range -- the concatenation must be the sorted shard."""
    keys = rand_keys(n, 31 + rounds) if n else np.zeros(0, np.uint32)
    rc, out = _dist_step(gpu_context, None, keys, rounds)
    assert rc == 0, out
    assert np.array_equal(out, np.sort(keys))


def test_dist_step_c_abi_reports_unbalanced_key_ranges(gpu_context):
    """All keys below 2^20 share top byte 0: with two ranks no byte-aligned cut can balance them.  At world size 1
    there is nothing to balance (one range), so force the check through a tiny capacity instead."""
    keys = rand_keys(300000, 5) >> np.uint32(12)
    rc, msg = _dist_step(gpu_context, None, keys, 2, capacity=300000)
    assert rc == 0  # one rank: its range is everything, and it fits
    lib = gpu_context.lib
    d = ctypes.c_void_p()
    assert lib.vrs_dist_create(gpu_context.handle, None, 0, 2, 1000, 1, ctypes.byref(d)) != 0  # world 2 needs a communicator
    assert b"communicator" in lib.vrs_dist_last_error(None)


def test_dist_step_c_abi_with_a_real_rccl_communicator(gpu_context):
    """Same step with a single-rank RCCL communicator made by the RCCL copy in this process (PyTorch's): the count
    exchange runs as a real ncclAllGather on the context's stream; the library binds RCCL at run time."""
    torch = pytest.importorskip("torch")
    import os
    rccl_path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if not os.path.exists(rccl_path):
        pytest.skip("no librccl.so next to torch")
    rccl = ctypes.CDLL(rccl_path, mode=ctypes.RTLD_GLOBAL)

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        # world size 1 WITH a communicator: vrs_dist_create binds RCCL, the all-gather is the collective itself
        lib = gpu_context.lib
        keys = rand_keys(3000001, 77)
        d = ctypes.c_void_p()
        assert lib.vrs_dist_create(gpu_context.handle, comm, 0, 1, 4000000, 4, ctypes.byref(d)) == 0, lib.vrs_dist_last_error(None)
        kb = vrs.Buffer.fillDeviceWithStagingBuffer(gpu_context, S(4 * keys.size), keys)
        out_buf, out_n = ctypes.c_void_p(), ctypes.c_uint32()
        for _ in range(2):  # twice: the receive buffer is reused across steps
            rc = lib.vrs_dist_sort_keys_u32(d, kb.handle, keys.size, ctypes.byref(out_buf), ctypes.byref(out_n))
            assert rc == 0, lib.vrs_dist_last_error(d)
            gpu_context.waitIdle()
            out = np.empty(out_n.value, np.uint32)
            gpu_context.check(lib.vrs_buffer_download(gpu_context.handle, out_buf, out.ctypes.data_as(ctypes.c_void_p), out.nbytes))
            assert np.array_equal(out, np.sort(keys))
        kb.release()
        lib.vrs_dist_destroy(d)
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)
