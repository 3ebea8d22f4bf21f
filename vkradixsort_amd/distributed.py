"""Key-range sharded sort across the GPUs of one node (BASELINE.json config 5).

The reference is single-GPU; this is the build's multi-GPU path.  One process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  Per step:

  1. local step   one radix pass on the TOP byte (histograms + stable scatter, shift 24) groups the shard
                  by top byte, so every key range [byte lo, byte hi) is one contiguous slice
  2. splitters    ONE all-gather of every rank's 256 top-byte counts (world x 2 KiB, latency bound); each rank
                  derives the same world-1 byte boundaries plus its own send and receive counts from it
  3. exchange     ONE variable-size all-to-all of the keys (dist.all_to_all_single with split sizes: grouped
                  send/recv on every xGMI link at once)
  4. merge step   the received runs are sorted locally by the four-pass multi_radixsort

Rank g ends up holding range g in ascending order; the global result is the concatenation of ranks 0..W-1.
HBM bytes per key: 12 (step 1) + 48 (step 4) = 60, plus one trip over xGMI for (world-1)/world of the keys.

The device work is behind `LocalSortBackend`; the product backend drives the C ABI on torch's current stream.
The CPU tests substitute a numpy backend to exercise steps 2-3 under gloo (the product never does).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import numpy as np

RADIX_SORT_BINS = 256


def plan_splitters(global_counts: np.ndarray, world_size: int) -> np.ndarray:
    """Byte boundaries b[0]=0 <= b[1] <= ... <= b[world]=256: rank q owns top bytes [b[q], b[q+1]).
    Greedy: boundary q is the byte at which the cumulative count first reaches q/world of the total,
    snapped to whichever side is closer.  Deterministic, identical on every rank."""
    counts = np.asarray(global_counts, dtype=np.int64)
    assert counts.shape == (RADIX_SORT_BINS,)
    cum = np.concatenate([[0], np.cumsum(counts)])
    total = int(cum[-1])
    bounds = np.zeros(world_size + 1, dtype=np.int64)
    bounds[world_size] = RADIX_SORT_BINS
    for q in range(1, world_size):
        target = total * q / world_size
        hi = int(np.searchsorted(cum, target, side="left"))
        hi = min(max(hi, 0), RADIX_SORT_BINS)
        lo = max(hi - 1, 0)
        pick = lo if abs(cum[lo] - target) <= abs(cum[hi] - target) else hi
        bounds[q] = max(pick, bounds[q - 1])
    return bounds


def send_counts_from_digit_base(digit_base: np.ndarray, n_local: int, bounds: np.ndarray) -> np.ndarray:
    """digit_base[d] = first position of top byte d in the grouped shard (exclusive prefix); returns the
    number of keys going to each rank."""
    base = np.concatenate([np.asarray(digit_base, dtype=np.int64), [n_local]])
    return (base[bounds[1:]] - base[bounds[:-1]]).astype(np.int64)


class LocalSortBackend:
    """Device work of steps 1 and 4 on one rank."""

    def group_by_top_byte(self, keys, n: int):
        """-> (grouped keys tensor (len >= n), digit_base uint32[256])"""
        raise NotImplementedError

    def sort(self, keys, n: int):
        """-> sorted keys tensor (first n entries)"""
        raise NotImplementedError


@dataclass
class StepResult:
    keys: object  # tensor holding this rank's range, ascending, first `count` entries valid
    count: int
    bounds: np.ndarray
    send_counts: np.ndarray
    recv_counts: np.ndarray


class HipLocalSortBackend(LocalSortBackend):
    """The product backend: C-ABI stages on torch's current stream, torch tensors as device memory."""

    def __init__(self, device_index: int, capacity: int, blocks_per_workgroup: int = 32):
        import torch

        from . import engine
        self.torch = torch
        self.engine = engine
        self.B = blocks_per_workgroup
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        self.ctx = engine.GPUContext(device_index, stream=torch.cuda.current_stream().cuda_stream)
        self.ctx.init()
        self.capacity = int(capacity)
        self.scratch = torch.empty(self.capacity, dtype=torch.int32, device=self.device)
        w_max = self.ctx.lib.vrs_workgroup_count(self.capacity, self.B)
        self.hist = torch.empty(max(w_max, 1) * RADIX_SORT_BINS, dtype=torch.int32, device=self.device)
        self._wrapped = {}

    def close(self):
        for b in self._wrapped.values():
            b[0].release()
        self._wrapped.clear()
        self.ctx.shutdown()

    def _buf(self, t):
        key = (t.data_ptr(), t.numel())
        if key not in self._wrapped:
            S = self.engine.Buffer.BufferSettings
            self._wrapped[key] = (self.engine.Buffer(self.ctx, S(t.numel() * 4), device_ptr=t.data_ptr()), t)
        return self._wrapped[key][0]

    def _pass(self, src, dst, n, shift):
        lib, ctx = self.ctx.lib, self.ctx
        pc = self.engine.PushConstants(n, shift, lib.vrs_workgroup_count(n, self.B), self.B)
        h = self._buf(self.hist)
        ctx.check(lib.vrs_multi_radixsort_histograms(ctx.handle, self._buf(src).handle, h.handle, ctypes.byref(pc)))
        ctx.check(lib.vrs_multi_radixsort(ctx.handle, self._buf(src).handle, self._buf(dst).handle, h.handle,
                                          ctypes.byref(pc)))

    def group_by_top_byte(self, keys, n):
        if n > self.capacity:
            raise ValueError("shard larger than the backend capacity")
        self._pass(keys, self.scratch, n, 24)
        digit_base = np.empty(RADIX_SORT_BINS, dtype=np.uint32)
        self.ctx.check(self.ctx.lib.vrs_multi_radixsort_digit_offsets(self.ctx.handle,
                                                                     digit_base.ctypes.data_as(ctypes.c_void_p)))
        return self.scratch, digit_base

    def sort(self, keys, n):
        if n == 0:
            return keys
        if n > self.capacity or keys.numel() < n:
            raise ValueError("received more keys than the backend capacity")
        a, b = keys, self.scratch
        for i in range(4):
            self._pass(a, b, n, 8 * i)
            a, b = b, a
        return a  # four passes: back in `keys`


class RangeShardedSort:
    """Steps 1-4 over a torch.distributed process group."""

    def __init__(self, backend: LocalSortBackend, recv_capacity: int, make_empty, process_group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.backend = backend
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.recv_capacity = int(recv_capacity)
        self.recv = make_empty(self.recv_capacity)  # int32 storage for uint32 keys
        self.device = self.recv.device

    def step(self, keys, n: int) -> StepResult:
        torch, dist = self.torch, self.dist
        # 1. local step
        grouped, digit_base = self.backend.group_by_top_byte(keys, n)
        base = np.concatenate([digit_base.astype(np.int64), [n]])
        local_counts = np.diff(base)
        # 2. ONE small collective: everybody learns everybody's 256 top-byte counts (world x 2 KiB), from which
        #    each rank derives the same splitters, its send counts and its receive counts without further traffic
        mine = torch.from_numpy(local_counts.copy()).to(self.device)
        table = torch.empty(self.world * RADIX_SORT_BINS, dtype=mine.dtype, device=self.device)
        dist.all_gather_into_tensor(table, mine, group=self.group)
        all_counts = table.cpu().numpy().reshape(self.world, RADIX_SORT_BINS)
        bounds = plan_splitters(all_counts.sum(axis=0), self.world)
        send_counts = send_counts_from_digit_base(digit_base, n, bounds)
        lo, hi = int(bounds[self.rank]), int(bounds[self.rank + 1])
        recv_counts = all_counts[:, lo:hi].sum(axis=1).astype(np.int64)
        total = int(recv_counts.sum())
        if total > self.recv_capacity:
            raise RuntimeError(f"rank {self.rank}: receives {total} keys, capacity {self.recv_capacity}")
        # 3. exchange the keys: one variable-size all-to-all (grouped send/recv over every xGMI link at once)
        dist.all_to_all_single(self.recv[:total], grouped[:n], output_split_sizes=[int(c) for c in recv_counts],
                               input_split_sizes=[int(c) for c in send_counts], group=self.group)
        # 4. merge step: local four-pass sort of the received runs
        out = self.backend.sort(self.recv, total)
        return StepResult(out, total, bounds, send_counts, recv_counts)
