"""NUM_BLOCKS_PER_WORKGROUP sweep, the MI355X counterpart of the reference's timings/radixsort_multi_*.png
(README.md:253-265).  Emits the CSV the reference only had commented out (MultiRadixSort.cpp:78-80):
    NUM_ELEMENTS NUM_BLOCKS_PER_WORKGROUP gpuSortTime[ms] cpuSortTime[ms]
plus the single_radixsort time per N.  GPU time = first pass enqueue -> queue idle (the reference's timed region),
minimum of 5 repetitions; CPU time = numpy's sort of the same keys (the C++ example prints std::sort's)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import vkradixsort_amd as vrs  # noqa: E402

S = vrs.Buffer.BufferSettings


def gpu_time(gpu, keys, B, reps=5):
    n = keys.size
    m = vrs.MultiRadixSort(NUM_BLOCKS_PER_WORKGROUP=B, keys=keys, quiet=True)
    m.setup(gpu)
    src = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), keys)
    best = 1e9
    for r in range(reps + 1):
        m.m_buffers[0].copyFrom(src)
        gpu.waitIdle()
        t0 = time.perf_counter()
        m.enqueueSort()
        gpu.waitIdle()
        if r:
            best = min(best, time.perf_counter() - t0)
    out = m.download()
    ok = bool(np.all(out[1:] >= out[:-1]))
    src.release()
    m.releaseBuffers()
    m.m_pass.release()
    return best * 1e3, ok


def single_time(gpu, keys, reps=3):
    n = keys.size
    b0 = vrs.Buffer(gpu, S(4 * n))
    b1 = vrs.Buffer(gpu, S(4 * n))
    src = vrs.Buffer.fillDeviceWithStagingBuffer(gpu, S(4 * n), keys)
    best = 1e9
    for r in range(reps + 1):
        b0.copyFrom(src)
        gpu.waitIdle()
        t0 = time.perf_counter()
        gpu.check(gpu.lib.vrs_single_radixsort(gpu.handle, b0.handle, b1.handle, n))
        gpu.waitIdle()
        if r:
            best = min(best, time.perf_counter() - t0)
    for b in (b0, b1, src):
        b.release()
    return best * 1e3


def main():
    out = Path(sys.argv[1]) if len(sys.argv) > 1 else Path("gpurun_out/block_sweep.csv")
    out.parent.mkdir(parents=True, exist_ok=True)
    lines = ["# NUM_ELEMENTS NUM_BLOCKS_PER_WORKGROUP gpuSortTime_ms cpuSortTime_ms (B=0: single_radixsort)"]
    with vrs.GPUContext(0) as gpu:
        for e in range(2, 9):
            n = 10 ** e
            keys = np.random.RandomState(1).randint(0, 2 ** 32, size=n, dtype=np.uint32)
            t0 = time.perf_counter()
            np.sort(keys)
            cpu_ms = (time.perf_counter() - t0) * 1e3
            if n <= 10 ** 6:
                lines.append(f"{n} 0 {single_time(gpu, keys):.4f} {cpu_ms:.4f}")
            B = 1
            while B <= 16384:
                W = gpu.lib.vrs_workgroup_count(n, B)
                if W <= 400000 and (B == 1 or n // (256 * B) >= 1 or B <= 32):
                    ms, ok = gpu_time(gpu, keys, B)
                    lines.append(f"{n} {B} {ms:.4f} {cpu_ms:.4f}" + ("" if ok else " NOT_SORTED"))
                    print(lines[-1], flush=True)
                B *= 2
    out.write_text("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
