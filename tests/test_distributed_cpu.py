"""World-size-2 gloo tests of the key-range exchange (steps 2-3 of vkradixsort_amd.distributed) on CPU.
The device work (steps 1 and 4) is substituted by a numpy backend that lives ONLY here; the product
backend (HipLocalSortBackend) needs a GPU and is covered by the -m gpu run."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from vkradixsort_amd.distributed import plan_splitters, send_counts_from_digit_base  # noqa: E402


def test_plan_splitters_uniform_and_skewed():
    c = np.full(256, 1000, dtype=np.int64)
    assert plan_splitters(c, 8).tolist() == [0, 32, 64, 96, 128, 160, 192, 224, 256]
    assert plan_splitters(c, 2).tolist() == [0, 128, 256]
    assert plan_splitters(c, 1).tolist() == [0, 256]
    # everything in one byte: one rank takes it all, boundaries stay monotone
    s = np.zeros(256, dtype=np.int64)
    s[17] = 10 ** 6
    b = plan_splitters(s, 4)
    assert b[0] == 0 and b[-1] == 256 and np.all(np.diff(b) >= 0)
    # the reference's 28-bit keys: only top bytes 0..15 are populated
    k = np.zeros(256, dtype=np.int64)
    k[:16] = 500
    assert plan_splitters(k, 8).tolist() == [0, 2, 4, 6, 8, 10, 12, 14, 256]
    # empty input
    assert plan_splitters(np.zeros(256, dtype=np.int64), 4)[-1] == 256


def test_send_counts_cover_the_shard():
    rs = np.random.RandomState(0)
    counts = rs.randint(0, 50, 256)
    base = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint32)
    n = int(counts.sum())
    bounds = plan_splitters(counts, 8)
    sc = send_counts_from_digit_base(base, n, bounds)
    assert sc.sum() == n and len(sc) == 8
    for q in range(8):
        assert sc[q] == counts[bounds[q]:bounds[q + 1]].sum()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_per_rank, mode, q, rounds=4):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch
        import torch.distributed as dist

        from vkradixsort_amd.distributed import LocalSortBackend, RangeShardedSort

        class NumpyBackend(LocalSortBackend):  # test double for the device work
            def group_by_top_byte(self, keys, n):
                k = keys[:n].numpy().view(np.uint32)
                top = k >> np.uint32(24)
                order = np.argsort(top, kind="stable")
                counts = np.bincount(top, minlength=256)
                base = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint32)
                return torch.from_numpy(k[order].view(np.int32).copy()), base

            def partition_by_splitters(self, keys, n, splitters):
                k = keys[:n].numpy().view(np.uint32)
                bucket = np.searchsorted(splitters, k, side="right")
                order = np.argsort(bucket, kind="stable")
                counts = np.bincount(bucket, minlength=len(splitters) + 1)
                base = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
                return torch.from_numpy(k[order].view(np.int32).copy()), base

            def sort(self, keys, n, key_floor=0):
                assert n == 0 or int(keys[:n].numpy().view(np.uint32).min()) >= key_floor  # the hint must hold
                k = np.sort(keys[:n].numpy().view(np.uint32))
                keys[:n] = torch.from_numpy(k.view(np.int32))
                return keys

        dist.init_process_group("gloo", rank=rank, world_size=world)
        rs = np.random.RandomState(1000 + rank)
        if mode == "uniform":
            shard = rs.randint(0, 2 ** 32, n_per_rank, dtype=np.uint32)
        elif mode == "28bit":
            shard = rs.randint(0, 2 ** 32, n_per_rank, dtype=np.uint32) >> np.uint32(4)
        elif mode == "tiny":  # below RangeShardedSort.small_total: gather path, ragged shards
            shard = rs.randint(0, 2 ** 32, 500 + 37 * rank, dtype=np.uint32)
        elif mode == "small":  # every key below 2^20: one top byte holds everything -> sampled splitters
            shard = rs.randint(0, 2 ** 20, n_per_rank, dtype=np.uint32)
        elif mode == "clustered":  # two narrow clusters far apart
            shard = (rs.randint(0, 5000, n_per_rank, dtype=np.uint32) + np.where(rs.rand(n_per_rank) < 0.7, 3_000_000_000, 17)).astype(np.uint32)
        else:  # skewed: rank 0 holds only large keys, rank 1 only small ones
            shard = rs.randint(0, 2 ** 31, n_per_rank, dtype=np.uint32) + (np.uint32(2 ** 31) if rank == 0 else np.uint32(0))
        sorter = RangeShardedSort(NumpyBackend(), recv_capacity=2 * n_per_rank * world,
                                  make_empty=lambda n: torch.empty(n, dtype=torch.int32), rounds=rounds)
        res = sorter.step(torch.from_numpy(shard.view(np.int32).copy()), shard.size,
                          n_total_hint=(shard.size * world if mode == "tiny" else None))
        out = res.keys[:res.count].numpy().view(np.uint32).copy()
        gathered = [None] * world
        dist.all_gather_object(gathered, (shard, out))
        dist.destroy_process_group()
        if rank == 0:
            all_in = np.concatenate([g[0] for g in gathered])
            all_out = np.concatenate([g[1] for g in gathered])
            ok = bool(np.array_equal(all_out, np.sort(all_in)))
            sizes = [len(g[1]) for g in gathered]
            q.put((ok, sizes, res.bounds.tolist()))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put(("error", traceback.format_exc(), str(e)))
        raise


@pytest.mark.parametrize("mode,world,rounds", [("uniform", 2, 4), ("28bit", 2, 4), ("skewed", 2, 4), ("uniform", 2, 1),
                                               ("uniform", 3, 2), ("small", 2, 4), ("clustered", 3, 2), ("tiny", 3, 4)])
def test_range_sharded_sort_gloo(mode, world, rounds):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n = 20000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, mode, q, rounds)) for r in range(world)]
    for p in procs:
        p.start()
    result = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
    assert result[0] is True, result
    ok, sizes, bounds = result
    if mode == "tiny":
        assert sum(sizes) == sum(500 + 37 * r for r in range(world)) and max(sizes) - min(sizes) <= 1
        return
    assert sum(sizes) == world * n and bounds[0] == 0 and bounds[-1] in (256, world * rounds)  # byte cuts or sampled parts
    if mode in ("uniform", "small", "clustered"):
        assert max(sizes) < 1.3 * n, sizes  # balanced ranges (byte cuts or sampled splitters)
