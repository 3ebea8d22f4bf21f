"""Time the key+payload sort (BASELINE.json configs[3]) and small-N sorts (development aid)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vkradixsort_amd as vrs  # noqa: E402
from vkradixsort_amd import capi  # noqa: E402


def run(gpu, n, B, pairs, reps=5, variant=0):
    keys = np.random.RandomState(1).randint(0, 2 ** 32, size=n, dtype=np.uint32)
    vals = np.arange(n, dtype=np.uint32) if pairs else None
    gpu.setTuning(capi.VRS_TUNE_SCATTER_VARIANT, variant)
    m = vrs.MultiRadixSort(NUM_BLOCKS_PER_WORKGROUP=B, keys=keys, values=vals, quiet=True)
    m.setup(gpu)
    S = vrs.Buffer.BufferSettings
    src = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), keys)
    times = []
    for r in range(reps + 2):
        m.m_buffers[0].copyFrom(src)
        gpu.waitIdle()
        if r == 2:
            gpu.profileReset()
            gpu.profileEnable(True)
        t0 = time.perf_counter()
        m.enqueueSort()
        gpu.waitIdle()
        if r >= 2:
            times.append(time.perf_counter() - t0)
    gpu.profileEnable(False)
    t = min(times)
    bpk = 80 if pairs else 48
    line = f"N={n} B={B} pairs={pairs} variant={variant} min={t*1e3:.3f}ms {n/t/1e9:.2f} G/s {bpk*n/t/8e12*100:.1f}%roof({bpk}B)"
    for kid, name in capi.KERNEL_NAMES.items():
        cnt, ms = gpu.profileQuery(kid)
        if cnt:
            line += f" | {name}: {ms/cnt*1e3:.1f}us"
    print(line, flush=True)
    src.release()
    m.releaseBuffers()
    m.m_pass.release()
    while gpu.getActiveIndex() != 0:
        gpu.incrementActiveIndex()


with vrs.GPUContext(0) as gpu:
    run(gpu, 10 ** 8, 32, True)
    run(gpu, 10 ** 8, 16, True)
    run(gpu, 10 ** 8, 32, False)
