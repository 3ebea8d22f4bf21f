"""Host-side profile of RangeShardedSort.step at world size 1 (where does the wall time of a step go?)"""
import cProfile
import os
import pstats
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from vkradixsort_amd.distributed import HipLocalSortBackend, RangeShardedSort  # noqa: E402

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)
n = 10 ** 8
keys = torch.from_numpy(np.random.RandomState(1).randint(0, 2 ** 32, n, dtype=np.uint32).view(np.int32)).to(dev)
cap = int(n * 1.25) + 4096
backend = HipLocalSortBackend(0, capacity=cap)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sorter = RangeShardedSort(backend, recv_capacity=cap, make_empty=lambda m: torch.empty(m, dtype=torch.int32, device=dev), rounds=R)
work = keys.clone()
for _ in range(3):
    work.copy_(keys)
    sorter.step(work, n, n_total_hint=n)
torch.cuda.synchronize()
# GPU-busy time of one step from events around it vs host wall
ts = []
for _ in range(5):
    work.copy_(keys)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sorter.step(work, n, n_total_hint=n)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
print("host-return ms, complete ms:", [(round(a, 3), round(b, 3)) for a, b in ts])
pr = cProfile.Profile()
work.copy_(keys)
torch.cuda.synchronize()
pr.enable()
for _ in range(10):
    sorter.step(work, n, n_total_hint=n)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
dist.destroy_process_group()
