"""Key-range sharded sort across the GPUs of one node (BASELINE.json configs[4]) -- the Python
orchestration.  The step ALSO exists behind the C ABI (csrc/vrs_dist.hip: vrs_dist_sort_keys_u32 over an RCCL communicator
or any injected transport), and that one has the leaner shape -- the single-GPU hybrid sort with the exchange between its two
MSD passes, 28 B/key per GPU; bench.py --gpus N times it by default (--dist-path python times this module).  What this
module adds on top of the same C entry points: sampled splitters for keys that byte-aligned ranges cannot balance, and a
gather path for totals too small to amortise an exchange.

The reference is single-GPU; this is the build's multi-GPU path.  One process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  Per step:

  1. local step   one radix pass on the TOP byte (histograms + stable scatter, shift 24) groups the shard
                  by top byte, so every key range [byte lo, byte hi) is one contiguous slice
  2. splitters    ONE all-gather of every rank's 256 top-byte counts (world x 2 KiB, latency bound); each rank
                  derives the same world-1 byte boundaries plus its own send and receive counts from it.
                  If byte-aligned cuts would overload a rank by more than 15 % (small or clustered keys), the ranks
                  pool 2048 sampled keys each, cut at sample quantiles and redo step 1 as a range partition
                  (vrs_range_partition: bucket = number of splitters <= key) -- any distribution without massive ties
  3. exchange     the range of every rank is cut into R sub-ranges ("rounds", still top-byte boundaries); round r
                  moves every rank's keys of sub-range r with one batch of grouped send/recv (all xGMI links at once)
  4. merge step   round r's keys are sorted by the four-pass multi_radixsort WHILE round r+1 is on the wire;
                  the sub-ranges are disjoint and ascending, so their concatenation is the sorted range -- no merge

Rank g ends up holding range g in ascending order; the global result is the concatenation of ranks 0..W-1.
HBM bytes per key: 12 (step 1) + what vrs_sort_keys_u32 moves for a received sub-range in step 4 (28 in its hybrid form from
1.3e7 keys on, 36 in its LSD form below) = 40 ... 48, plus one trip over xGMI for (world-1)/world of the keys; only the first
round's transfer and the last round's sort are not overlapped.

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
        """-> (grouped keys tensor (len >= n), digit_base[256]: first position of every top byte in it -- a host
        uint32 array, or a device int32 tensor when the backend can hand it over without a host round trip)"""
        raise NotImplementedError

    def partition_by_splitters(self, keys, n: int, splitters: np.ndarray):
        """stable grouping by range r = #splitters <= key -> (grouped keys tensor, first position of every range,
        int64[len(splitters) + 2] with the total appended)"""
        raise NotImplementedError

    def sort(self, keys, n: int, key_floor: int = 0):
        """-> sorted keys tensor (first n entries).  key_floor: every key is >= this (a hint the product backend hands to
        vrs_sort_keys_u32_ranged: a received sub-range is bucketed from its own first key, not from 0)"""
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
        self.grouped = torch.empty(self.capacity, dtype=torch.int32, device=self.device)  # output of step 1, read by the sends
        self.scratch = torch.empty(self.capacity, dtype=torch.int32, device=self.device)  # ping-pong partner of the local sorts
        w_max = self.ctx.lib.vrs_workgroup_count(self.capacity, self.B)
        self.hist = torch.empty(max(w_max, 1) * RADIX_SORT_BINS, dtype=torch.int32, device=self.device)
        self.digit_base = torch.empty(RADIX_SORT_BINS, dtype=torch.int32, device=self.device)
        self._wrapped = {}
        self._in_use = set()

    def close(self):
        for b in self._wrapped.values():
            b[0].release()
        self._wrapped.clear()
        self.ctx.shutdown()

    def _buf(self, t):
        """C-ABI wrapper of a tensor's memory, cached.  The backend's own buffers (hist, grouped, scratch, digit_base)
        stay wrapped for its whole life; the wrappers of callers' tensors and sub-range views (they differ from step
        to step) are dropped oldest-first once there are more than 256 -- never one handed out in the current call."""
        key = (t.data_ptr(), t.numel())
        hit = self._wrapped.get(key)
        if hit is not None:
            return hit[0]
        S = self.engine.Buffer.BufferSettings
        buf = self.engine.Buffer(self.ctx, S(t.numel() * 4), device_ptr=t.data_ptr())
        pinned = any(t is own for own in (self.hist, self.grouped, self.scratch, self.digit_base))
        self._wrapped[key] = (buf, t, pinned)
        if len(self._wrapped) > 256:
            keep = self._in_use | {key}
            for k in [k for k, v in self._wrapped.items() if not v[2] and k not in keep][:64]:
                self._wrapped.pop(k)[0].release()
        return buf

    def _handles(self, *tensors):
        """Handles of several tensors for ONE C-ABI call: none of them is evicted while the others are resolved."""
        self._in_use = {(t.data_ptr(), t.numel()) for t in tensors}
        try:
            return [self._buf(t).handle for t in tensors]
        finally:
            self._in_use = set()

    def _pass(self, src, dst, n, shift):
        lib, ctx = self.ctx.lib, self.ctx
        pc = self.engine.PushConstants(n, shift, lib.vrs_workgroup_count(n, self.B), self.B)
        s, d, h = self._handles(src, dst, self.hist)
        ctx.check(lib.vrs_multi_radixsort_histograms(ctx.handle, s, h, ctypes.byref(pc)))
        ctx.check(lib.vrs_multi_radixsort(ctx.handle, s, d, h, ctypes.byref(pc)))

    def group_by_top_byte(self, keys, n):
        if n > self.capacity:
            raise ValueError("shard larger than the backend capacity")
        if n == 0:
            # an empty shard: both stages return without launching anything, so there is no offset row to read --
            # every top byte starts at position 0
            self.digit_base.zero_()
            return self.grouped, self.digit_base
        self._pass(keys, self.grouped, n, 24)
        # stays on the device: the step feeds it into its count all-gather and synchronises once, after that
        self.ctx.check(self.ctx.lib.vrs_multi_radixsort_digit_offsets_device(self.ctx.handle, self._buf(self.digit_base).handle))
        return self.grouped, self.digit_base

    def partition_by_splitters(self, keys, n, splitters):
        if n > self.capacity:
            raise ValueError("shard larger than the backend capacity")
        torch = self.torch
        sp = np.ascontiguousarray(splitters, dtype=np.uint32)
        sp_t = torch.from_numpy(np.concatenate([sp, np.zeros(1, np.uint32)]).view(np.int32)).to(self.device)
        torch.cuda.current_stream().synchronize()  # the upload ran on torch's stream == the context's stream; keep it simple
        lib, ctx = self.ctx.lib, self.ctx
        spb = self.engine.Buffer(ctx, self.engine.Buffer.BufferSettings(sp_t.numel() * 4), device_ptr=sp_t.data_ptr())
        if n == 0:  # nothing is launched for an empty shard: every range starts (and ends) at 0
            spb.release()
            return self.grouped, np.zeros(sp.size + 2, dtype=np.int64)
        k, g = self._handles(keys, self.grouped)
        ctx.check(lib.vrs_range_partition(ctx.handle, k, g, spb.handle, sp.size, n))
        first = np.empty(RADIX_SORT_BINS, dtype=np.uint32)
        ctx.check(lib.vrs_multi_radixsort_digit_offsets(ctx.handle, first.ctypes.data_as(ctypes.c_void_p)))  # synchronous
        spb.release()
        return self.grouped, np.concatenate([first[:sp.size + 1].astype(np.int64), [n]])

    def sort(self, keys, n, key_floor=0):
        if n == 0:
            return keys
        if n > self.capacity or keys.numel() < n:
            raise ValueError("received more keys than the backend capacity")
        # the library's own loop over the passes (one counting read + look-back scatters from 2^13 keys on, the hybrid form
        # from 1.3e7 keys on -- bucketed from key_floor, so a sub-range of the key space fills its 16384 buckets evenly); the
        # result is back in `keys`, `scratch` is the ping-pong partner
        ctx = self.ctx
        k, t = self._handles(keys, self.scratch)
        ctx.check(ctx.lib.vrs_sort_keys_u32_ranged(ctx.handle, k, t, n, int(key_floor) & 0xFFFFFFFF))
        return keys


class RangeShardedSort:
    """Steps 1-4 over a torch.distributed process group."""

    def __init__(self, backend: LocalSortBackend, recv_capacity: int, make_empty, process_group=None, rounds: int = 4,
                 max_imbalance: float = 1.15, samples_per_rank: int = 2048, small_total: int = 1 << 21):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.backend = backend
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.rounds = max(1, min(int(rounds), RADIX_SORT_BINS // max(self.world, 1)))
        self.small_total = int(small_total)  # below this many keys in total the exchange does not pay: gather + local sort
        self.max_imbalance = float(max_imbalance)
        self.samples_per_rank = int(samples_per_rank)
        self.recv_capacity = int(recv_capacity)
        self.recv = make_empty(self.recv_capacity)  # int32 storage for uint32 keys
        self.device = self.recv.device

    def _step_small(self, keys, n: int, sizes: np.ndarray) -> StepResult:
        """Too few keys to amortise a partition pass, an exchange and per-range sorts (fixed costs of ~0.3 ms): every rank
        gathers all keys (one all-gather), sorts them locally and keeps its 1/world slice by position."""
        torch, dist = self.torch, self.dist
        world, me = self.world, self.rank
        m = int(sizes.max())
        total = int(sizes.sum())
        if total > self.recv_capacity or world * m > self.recv_capacity:
            raise RuntimeError("receive buffer too small for the gather path")
        padded = keys[:n] if n == m else torch.cat([keys[:n], torch.zeros(m - n, dtype=keys.dtype, device=self.device)])
        pool = torch.empty(world * m, dtype=keys.dtype, device=self.device)
        dist.all_gather_into_tensor(pool, padded.contiguous(), group=self.group)
        off = 0
        for q in range(world):  # compact the ragged shards
            c = int(sizes[q])
            self.recv[off:off + c].copy_(pool[q * m:q * m + c])
            off += c
        self.backend.sort(self.recv[:total], total)
        cuts = (np.arange(world + 1) * total) // world
        lo, hi = int(cuts[me]), int(cuts[me + 1])
        mine = self.recv[lo:hi].clone()
        self.recv[:hi - lo].copy_(mine)
        return StepResult(self.recv, hi - lo, cuts, np.zeros(world, np.int64), np.zeros(world, np.int64))

    def step(self, keys, n: int, n_total_hint: int | None = None) -> StepResult:
        """n_total_hint: total number of keys over all ranks if the caller knows it (avoids a size collective);
        totals below `small_total` take the gather path -- the all-to-all only runs where N amortises it."""
        torch, dist = self.torch, self.dist
        world, R, me = self.world, self.rounds, self.rank
        if world > 1 and n_total_hint is not None and n_total_hint < self.small_total:
            sz = torch.tensor([n], dtype=torch.int64, device=self.device)
            all_sz = torch.empty(world, dtype=torch.int64, device=self.device)
            dist.all_gather_into_tensor(all_sz, sz, group=self.group)
            return self._step_small(keys, n, all_sz.cpu().numpy())
        # 1. local step
        grouped, digit_base = self.backend.group_by_top_byte(keys, n)
        # 2. ONE small collective: everybody learns everybody's 256 top-byte counts (world x 2 KiB), from which
        #    each rank derives the same splitters, its send counts and its receive counts without further traffic
        if torch.is_tensor(digit_base):  # device-resident prefix: counts by differencing on the device, one host sync below
            b64 = digit_base.to(torch.int64) & 0xFFFFFFFF
            mine = torch.empty(RADIX_SORT_BINS, dtype=torch.int64, device=self.device)
            mine[:-1] = b64[1:] - b64[:-1]
            mine[-1] = n - b64[-1]  # scalar operand: no host-to-device copy
        else:
            mine = torch.from_numpy(np.diff(np.concatenate([digit_base.astype(np.int64), [n]]))).to(self.device)
        table = torch.empty(world * RADIX_SORT_BINS, dtype=mine.dtype, device=self.device)
        dist.all_gather_into_tensor(table, mine.contiguous(), group=self.group)
        all_counts = table.cpu().numpy().reshape(world, RADIX_SORT_BINS)
        base = np.concatenate([[0], np.cumsum(all_counts[me])]).astype(np.int64)  # my own prefix, back from the table
        # world*R parts of (almost) equal size; part q*R + r = rank q, round r
        parts = plan_splitters(all_counts.sum(axis=0), world * R)
        floors = np.minimum(parts[:-1].astype(np.int64), 255) << 24  # part p starts at top byte parts[p]
        bounds = parts[::R]
        per_rank = np.array([all_counts[:, bounds[q]:bounds[q + 1]].sum() for q in range(world)], dtype=np.int64)
        ideal = max(int(all_counts.sum()) / world, 1.0)
        if per_rank.max() > self.max_imbalance * ideal or per_rank.max() > self.recv_capacity:
            # top bytes too concentrated for byte-aligned cuts (small keys, clustered keys): cut at sampled key
            # values instead.  Same decision on every rank (it only depends on the gathered table).
            S = self.samples_per_rank
            idx = torch.linspace(0, max(n - 1, 0), S, device=self.device).long()
            mine_s = keys[:n][idx] if n else torch.zeros(S, dtype=keys.dtype, device=self.device)
            pool = torch.empty(world * S, dtype=keys.dtype, device=self.device)
            dist.all_gather_into_tensor(pool, mine_s.contiguous(), group=self.group)
            # sort the pooled sample where it is (unsigned order on int32 storage: flip the sign bit), bring back only
            # the P - 1 splitters
            P = world * R
            flipped = torch.sort(pool.to(torch.int32) ^ (-2 ** 31)).values
            picks = torch.from_numpy((np.arange(1, P) * pool.numel()) // P).to(self.device)
            splitters = (flipped[picks] ^ (-2 ** 31)).cpu().numpy().view(np.uint32)
            grouped, base = self.backend.partition_by_splitters(keys, n, splitters)
            local = np.diff(base)  # P ranges
            mine2 = torch.from_numpy(local.astype(np.int64)).to(self.device)
            table2 = torch.empty(world * P, dtype=mine2.dtype, device=self.device)
            dist.all_gather_into_tensor(table2, mine2, group=self.group)
            all_counts = table2.cpu().numpy().reshape(world, P)
            parts = np.arange(P + 1, dtype=np.int64)
            floors = np.concatenate([[0], splitters.astype(np.int64)])  # first key value of every range
            bounds = parts[::R]
        send_counts = (base[bounds[1:]] - base[bounds[:-1]]).astype(np.int64)
        # what I receive in round r from source s, and where it lands: rounds ascending, sources ascending
        my_parts = parts[me * R:(me + 1) * R + 1]
        recv_rs = np.stack([all_counts[:, my_parts[r]:my_parts[r + 1]].sum(axis=1) for r in range(R)]).astype(np.int64)
        round_total = recv_rs.sum(axis=1)
        round_off = np.concatenate([[0], np.cumsum(round_total)])
        total = int(round_off[-1])
        if total > self.recv_capacity:
            raise RuntimeError(f"rank {self.rank}: receives {total} keys, capacity {self.recv_capacity} "
                               "(too many equal keys to balance by key range)")

        def issue(r):
            """round r: grouped send/recv with every peer; my own slice is a device copy"""
            ops = []
            off = int(round_off[r])
            for src in range(world):
                cnt = int(recv_rs[r, src])
                if cnt and src != me:
                    ops.append(dist.P2POp(dist.irecv, self.recv[off:off + cnt], src, group=self.group))
                if cnt and src == me:
                    a = int(base[my_parts[r]])
                    self.recv[off:off + cnt].copy_(grouped[a:a + cnt])
                off += cnt
            for dst in range(world):
                lo, hi = int(parts[dst * R + r]), int(parts[dst * R + r + 1])
                a, b = int(base[lo]), int(base[hi])
                if b > a and dst != me:
                    ops.append(dist.P2POp(dist.isend, grouped[a:b], dst, group=self.group))
            return dist.batch_isend_irecv(ops) if ops else []

        # 3 + 4. round r+1 is put on the wire before round r is sorted
        pending = issue(0)
        for r in range(R):
            nxt = issue(r + 1) if r + 1 < R else []
            for w in pending:
                w.wait()
            cnt = int(round_total[r])
            if cnt:
                off = int(round_off[r])
                self.backend.sort(self.recv[off:off + cnt], cnt, int(floors[me * R + r]))
            pending = nxt
        return StepResult(self.recv, total, bounds, send_counts, recv_rs.sum(axis=0))
