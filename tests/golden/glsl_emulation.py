"""A second, LITERAL restatement of the reference's two compute shaders -- test infrastructure, like oracle/: an invocation-by-invocation
emulation of what a Vulkan device executes, written to be independent of oracle/vrs_oracle.c (which restates the NET EFFECT of the same
shaders: a running counter per digit) so that the stage-level fixtures are the agreement of two restatements instead of the word of one.

What is emulated, and where it is in the reference (multiradixsort/resources/shaders/):
  multi_radixsort_histograms.comp:31-55   one workgroup of 256 invocations per tile; a shared histogram of 256 counters zeroed by the
                                          invocations with lID < 256, a barrier, B rounds of "element wID*B*256 + index*256 + lID, if it
                                          exists, atomically increments the counter of its digit", a barrier, the histogram written to row wID.
  multi_radixsort.comp:45-77              every invocation lID < 256 walks ALL rows of the histogram table for digit lID: the keys of the rows
                                          before its own workgroup (local_histogram) and of all rows (histogram_count); subgroupAdd /
                                          subgroupExclusiveAdd over the subgroup, the elected invocation stores the subgroup's sum in sums[sID];
                                          barrier; subgroupExclusiveAdd over sums[lsID] broadcast from invocation sID; global_offsets[lID].
  multi_radixsort.comp:80-127             B rounds: the bin_flags bit masks zeroed (256 digits x 8 words), barrier, every invocation with an
                                          element sets ITS bit (word lID / 32, bit lID % 32) in its digit's mask with an atomic add and reads
                                          its digit's offset, barrier, rank = bitCount of the mask below its bit, the element written to
                                          offset + rank, the LAST invocation of a digit (rank == count - 1) advances the digit's offset by the
                                          count, barrier.
SUBGROUP_SIZE is a compile-time constant of the shader ("32 NVIDIA; 64 AMD", multi_radixsort.comp:13): both are emulated, and must agree.
With 32 the shader reads sums[lsID] for lsID up to 31 although sums has 256 / 32 = 8 entries; the exclusive add broadcast from invocation
sID < 8 never depends on those reads -- they are emulated as a poison value to show it.

Pure Python on purpose (small cases only: the fixtures have at most 20 000 keys); nothing here is imported by the product or shipped to the
GPU box beyond the .npz files it validated (tests/golden/make_golden.py, tests/test_oracle.py).
"""
from __future__ import annotations

import numpy as np

WORKGROUP_SIZE = 256
RADIX_SORT_BINS = 256
POISON = 0xDEADBEEF  # what an out-of-bounds read of `sums` returns here
MASK32 = 0xFFFFFFFF


def workgroup_count(num_elements: int, blocks_per_workgroup: int) -> int:
    """ComputePass.h:24-29 with MultiRadixSortPass.cpp's global size ceil(N / B): ceil(ceil(N / B) / 256) workgroups."""
    global_size = -(-num_elements // blocks_per_workgroup)
    return -(-global_size // WORKGROUP_SIZE)


def histograms_stage(keys, shift: int, num_workgroups: int, blocks_per_workgroup: int) -> np.ndarray:
    n = len(keys)
    table = np.zeros((num_workgroups, RADIX_SORT_BINS), dtype=np.uint32)
    for w_id in range(num_workgroups):
        histogram = [0] * RADIX_SORT_BINS  # shared; zeroed by lID < 256, then barrier()
        for index in range(blocks_per_workgroup):
            for l_id in range(WORKGROUP_SIZE):  # the invocations of the workgroup, in any order: the adds commute
                element_id = w_id * blocks_per_workgroup * WORKGROUP_SIZE + index * WORKGROUP_SIZE + l_id
                if element_id < n:
                    bin_ = (int(keys[element_id]) >> shift) & (RADIX_SORT_BINS - 1)
                    histogram[bin_] += 1  # atomicAdd
        # barrier(); every lID < 256 writes its counter
        table[w_id, :] = histogram
    return table


def _subgroup_add(values, subgroup_size):
    out = [0] * len(values)
    for base in range(0, len(values), subgroup_size):
        total = sum(values[base:base + subgroup_size]) & MASK32
        for lane in range(base, min(base + subgroup_size, len(values))):
            out[lane] = total
    return out


def _subgroup_exclusive_add(values, subgroup_size):
    out = [0] * len(values)
    for base in range(0, len(values), subgroup_size):
        run = 0
        for lane in range(base, min(base + subgroup_size, len(values))):
            out[lane] = run
            run = (run + values[lane]) & MASK32
    return out


def sort_stage(keys, table, shift: int, num_workgroups: int, blocks_per_workgroup: int, subgroup_size: int = 32):
    """Returns (global_offsets of every workgroup as they stand BEFORE its first round -- the per-workgroup offset table --, elements_out)."""
    assert subgroup_size in (32, 64) and RADIX_SORT_BINS % subgroup_size == 0
    n = len(keys)
    out = np.zeros(n, dtype=np.uint32)
    written = np.zeros(n, dtype=bool)
    offsets_table = np.zeros((num_workgroups, RADIX_SORT_BINS), dtype=np.uint32)
    n_sums = RADIX_SORT_BINS // subgroup_size
    for w_id in range(num_workgroups):
        # ---- :56-69, per invocation lID (all 256 pass `lID < RADIX_SORT_BINS`)
        local_histogram = [0] * WORKGROUP_SIZE
        histogram_count = [0] * WORKGROUP_SIZE
        for l_id in range(WORKGROUP_SIZE):
            count = 0
            for j in range(num_workgroups):
                t = int(table[j, l_id])
                if j == w_id:
                    local_histogram[l_id] = count
                count = (count + t) & MASK32
            histogram_count[l_id] = count
        sub_sum = _subgroup_add(histogram_count, subgroup_size)
        prefix_sum = _subgroup_exclusive_add(histogram_count, subgroup_size)
        sums = [0] * n_sums
        for l_id in range(WORKGROUP_SIZE):
            s_id, ls_id = divmod(l_id, subgroup_size)
            if ls_id == 0:  # subgroupElect(): the lowest active invocation of the subgroup
                sums[s_id] = sub_sum[l_id]
        # barrier()
        # ---- :72-76
        global_offsets = [0] * RADIX_SORT_BINS
        for s_id in range(WORKGROUP_SIZE // subgroup_size):
            lane_values = [sums[ls] if ls < n_sums else POISON for ls in range(subgroup_size)]  # sums[lsID]
            scanned = _subgroup_exclusive_add(lane_values, subgroup_size)
            sums_prefix_sum = scanned[s_id]  # subgroupBroadcast(..., sID)
            for ls_id in range(subgroup_size):
                l_id = s_id * subgroup_size + ls_id
                global_histogram = (sums_prefix_sum + prefix_sum[l_id]) & MASK32
                global_offsets[l_id] = (global_histogram + local_histogram[l_id]) & MASK32
        offsets_table[w_id, :] = global_offsets
        # ---- :80-127
        for index in range(blocks_per_workgroup):
            bin_flags = [[0] * (WORKGROUP_SIZE // 32) for _ in range(RADIX_SORT_BINS)]  # zeroed by lID < 256; barrier()
            element_in = [0] * WORKGROUP_SIZE
            bin_id = [0] * WORKGROUP_SIZE
            bin_offset = [0] * WORKGROUP_SIZE
            has = [False] * WORKGROUP_SIZE
            for l_id in range(WORKGROUP_SIZE):
                element_id = w_id * blocks_per_workgroup * WORKGROUP_SIZE + index * WORKGROUP_SIZE + l_id
                if element_id < n:
                    has[l_id] = True
                    element_in[l_id] = int(keys[element_id])
                    bin_id[l_id] = (element_in[l_id] >> shift) & (RADIX_SORT_BINS - 1)
                    bin_offset[l_id] = global_offsets[bin_id[l_id]]  # read before the barrier: nobody has advanced it in this round yet
                    word, bit = l_id // 32, 1 << (l_id % 32)
                    assert bin_flags[bin_id[l_id]][word] & bit == 0  # every invocation owns its bit: the atomic add sets it
                    bin_flags[bin_id[l_id]][word] = (bin_flags[bin_id[l_id]][word] + bit) & MASK32
            # barrier()
            advance = []
            for l_id in range(WORKGROUP_SIZE):
                if not has[l_id]:
                    continue
                flags_bin, flags_bit = l_id // 32, 1 << (l_id % 32)
                prefix = count = 0
                for i in range(WORKGROUP_SIZE // 32):
                    bits = bin_flags[bin_id[l_id]][i]
                    full_count = bin(bits).count("1")
                    partial_count = bin(bits & (flags_bit - 1)).count("1")
                    if i < flags_bin:
                        prefix += full_count
                    if i == flags_bin:
                        prefix += partial_count
                    count += full_count
                dst = (bin_offset[l_id] + prefix) & MASK32
                assert dst < n and not written[dst], "two elements for one slot, or a slot outside the buffer"
                out[dst] = element_in[l_id]
                written[dst] = True
                if prefix == count - 1:
                    advance.append((bin_id[l_id], count))
            for b, c in advance:  # atomicAdd(global_offsets[binID], count): one per digit that occurs in the round
                global_offsets[b] = (global_offsets[b] + c) & MASK32
            assert len({b for b, _ in advance}) == len(advance)
            # barrier()
    assert written.all()
    return offsets_table, out


def multi_radixsort(keys, blocks_per_workgroup: int, subgroup_size: int = 32):
    """MultiRadixSort::execute (multiradixsort/src/MultiRadixSort.cpp:37-61): four rounds of the two stages over two ping-pong buffers;
    yields (hist, offsets, pass output) per round."""
    cur = np.asarray(keys, dtype=np.uint32).copy()
    w = workgroup_count(len(cur), blocks_per_workgroup)
    for i in range(4):
        hist = histograms_stage(cur, 8 * i, w, blocks_per_workgroup)
        offsets, cur = sort_stage(cur, hist, 8 * i, w, blocks_per_workgroup, subgroup_size)
        yield hist, offsets, cur
