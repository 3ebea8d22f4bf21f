/*
 * vrs_oracle.c -- TEST INFRASTRUCTURE ONLY. Not part of the product path.
 *
 * A plain-C, single-threaded CPU restatement of the reference's multi_radixsort /
 * single_radixsort algorithm (VkRadixSort, GLSL compute shaders), written from the
 * behaviour of the shaders, stage by stage, so that every intermediate the HIP path
 * produces (the [W][256] histogram table, the per-workgroup offset table, each pass's
 * output buffer) can be compared bit-for-bit.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product library (vkradixsort_amd/csrc) never links, loads or calls it.
 *
 * PARITY PINNING: the reference ships no golden vectors, fixtures or unit tests
 *  and cannot be built in this image (needs Vulkan headers/loader, glslc,
 * the un-vendored SPIRV-Reflect submodule and a GPU).  Its one acceptance check is
 * "GPU output == std::sort of the same input, element for element"
 * (multiradixsort/src/MultiRadixSort.cpp:141-161).  The END-TO-END output of this
 * restatement is pinned to exactly that criterion (oracle/vrs_stdsort.cpp is that
 * std::sort path; tests/test_oracle.py checks equality on every fixture).  The
 * STAGE-LEVEL tables (histograms / offsets) are "parity unpinned": they follow the
 * shader text cited below but no reference-produced vector exists to pin them.
 * What stands in for one since round 6: tests/golden/glsl_emulation.py, a second,
 * independent restatement (a literal per-invocation emulation of both shaders: LDS
 * atomicAdd, bin_flags masks, bitCount, subgroup operations for SUBGROUP_SIZE 32 and 64),
 * asserted equal to this one on every fixture (make_golden.py, tests/test_oracle.py).
 *
 * Citations are file:line into /root/reference.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define VRS_WORKGROUP_SIZE 256u /* multi_radixsort.comp:11, multi_radixsort_histograms.comp:8 */
#define VRS_RADIX_SORT_BINS 256u /* multi_radixsort.comp:12 */

/* ceil-div as ComputePass::getDispatchSize, engine/include/engine/passes/ComputePass.h:24-29 */
static uint32_t ceil_div_u32(uint32_t a, uint32_t b) { return (a + b - 1u) / b; }

/*
 * Global invocation size and workgroup count exactly as the host computes them:
 * multiradixsort/src/MultiRadixSort.cpp:12-17 (gis = N / B, +1 if remainder) and
 * ComputePass.h:16-29 (W = ceil(gis / 256)).
 */
uint32_t vrs_oracle_global_invocation_size(uint32_t num_elements, uint32_t blocks_per_workgroup) {
    uint32_t gis = num_elements / blocks_per_workgroup;
    uint32_t rem = num_elements % blocks_per_workgroup;
    gis += rem > 0 ? 1u : 0u;
    return gis;
}

uint32_t vrs_oracle_workgroup_count(uint32_t num_elements, uint32_t blocks_per_workgroup) {
    return ceil_div_u32(vrs_oracle_global_invocation_size(num_elements, blocks_per_workgroup),
                        VRS_WORKGROUP_SIZE);
}

/*
 * Stage RADIX_SORT_HISTOGRAMS: multi_radixsort_histograms.comp:31-55.
 * hist[256*w + d] = number of keys i in workgroup w's tile with ((key>>shift)&255)==d,
 * tile(w) = { w*B*256 + b*256 + l : b<B, l<256 } intersected with [0,N)   (:42-50).
 * Every entry of the W*256 table is written (:53-55).
 */
void vrs_oracle_histograms(const uint32_t *keys_in, uint32_t *hist, uint32_t num_elements,
                           uint32_t shift, uint32_t num_workgroups, uint32_t blocks_per_workgroup) {
    for (uint32_t w = 0; w < num_workgroups; ++w) {
        uint32_t *h = hist + (size_t)VRS_RADIX_SORT_BINS * w;
        memset(h, 0, VRS_RADIX_SORT_BINS * sizeof(uint32_t)); /* :37-39 */
        for (uint32_t index = 0; index < blocks_per_workgroup; ++index) {
            for (uint32_t l = 0; l < VRS_WORKGROUP_SIZE; ++l) {
                /* :43, evaluated in 64 bit so the oracle itself cannot wrap */
                uint64_t e = (uint64_t)w * blocks_per_workgroup * VRS_WORKGROUP_SIZE +
                             (uint64_t)index * VRS_WORKGROUP_SIZE + l;
                if (e < num_elements) {
                    uint32_t bin = (keys_in[e] >> shift) & (VRS_RADIX_SORT_BINS - 1u); /* :46 */
                    h[bin] += 1u;                                                       /* :48 */
                }
            }
        }
    }
}

/*
 * Stage RADIX_SORT, part 1: multi_radixsort.comp:56-77.
 * For workgroup w and digit d:
 *   count_d  = sum_j hist[j][d]            (:58-63)
 *   local_d  = sum_{j<w} hist[j][d]        (:60)
 *   global_d = exclusive scan of count over digits (:64-75)
 *   offsets[w][d] = global_d + local_d     (:76)
 * Written for all W workgroups into offsets[W*256].
 */
void vrs_oracle_offsets(const uint32_t *hist, uint32_t *offsets, uint32_t num_workgroups) {
    uint32_t count[VRS_RADIX_SORT_BINS];
    memset(count, 0, sizeof(count));
    for (uint32_t w = 0; w < num_workgroups; ++w)
        for (uint32_t d = 0; d < VRS_RADIX_SORT_BINS; ++d)
            count[d] += hist[(size_t)VRS_RADIX_SORT_BINS * w + d];
    uint32_t global_prefix[VRS_RADIX_SORT_BINS];
    uint32_t run = 0;
    for (uint32_t d = 0; d < VRS_RADIX_SORT_BINS; ++d) {
        global_prefix[d] = run;
        run += count[d];
    }
    uint32_t local[VRS_RADIX_SORT_BINS];
    memset(local, 0, sizeof(local));
    for (uint32_t w = 0; w < num_workgroups; ++w) {
        for (uint32_t d = 0; d < VRS_RADIX_SORT_BINS; ++d) {
            offsets[(size_t)VRS_RADIX_SORT_BINS * w + d] = global_prefix[d] + local[d];
            local[d] += hist[(size_t)VRS_RADIX_SORT_BINS * w + d];
        }
    }
}

/*
 * Stage RADIX_SORT, part 2: multi_radixsort.comp:80-126.
 * For workgroup w, for block index 0..B-1 (:83), thread l handles element
 * e = w*B*256 + index*256 + l (:84) if e < N.  Its rank inside the block is the number
 * of lower-numbered threads of the SAME block with the same digit (:111-118, popcount
 * over the per-bin 256-bit flag mask below the thread's own bit); it is stored at
 * offset[d] + rank (:119) and afterwards offset[d] advances by the block's count of d
 * (:120-122).  Net effect: a stable counting sort on the digit.
 * `values_*` (may be NULL) ride along with the keys: the build's key+payload extension
 * (the reference has no payload buffers).
 */
void vrs_oracle_scatter(const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *values_in,
                        uint32_t *values_out, const uint32_t *hist, uint32_t num_elements,
                        uint32_t shift, uint32_t num_workgroups, uint32_t blocks_per_workgroup) {
    uint32_t *offsets = (uint32_t *)malloc((size_t)num_workgroups * VRS_RADIX_SORT_BINS * sizeof(uint32_t));
    vrs_oracle_offsets(hist, offsets, num_workgroups);
    for (uint32_t w = 0; w < num_workgroups; ++w) {
        uint32_t *global_offsets = offsets + (size_t)VRS_RADIX_SORT_BINS * w; /* LDS array :38 */
        for (uint32_t index = 0; index < blocks_per_workgroup; ++index) {
            uint32_t block_count[VRS_RADIX_SORT_BINS];
            memset(block_count, 0, sizeof(block_count)); /* bin_flags cleared :87-91 */
            for (uint32_t l = 0; l < VRS_WORKGROUP_SIZE; ++l) {
                uint64_t e = (uint64_t)w * blocks_per_workgroup * VRS_WORKGROUP_SIZE +
                             (uint64_t)index * VRS_WORKGROUP_SIZE + l;
                if (e < num_elements) {
                    uint32_t key = keys_in[e];
                    uint32_t bin = (key >> shift) & (VRS_RADIX_SORT_BINS - 1u); /* :97 */
                    /* threads visited in increasing l: block_count[bin] == #lower threads
                       with this bin == `prefix` of :111-118 */
                    uint32_t dst = global_offsets[bin] + block_count[bin]; /* :119 */
                    keys_out[dst] = key;
                    if (values_in) values_out[dst] = values_in[e];
                    block_count[bin] += 1u;
                }
            }
            for (uint32_t d = 0; d < VRS_RADIX_SORT_BINS; ++d) global_offsets[d] += block_count[d]; /* :120-122 */
        }
    }
    free(offsets);
}

/*
 * The host loop: MultiRadixSort::execute, multiradixsort/src/MultiRadixSort.cpp:37-61.
 * Four iterations (SORT_32BIT, :50-51), g_shift = 8*i (:57-58); iterations 0 and 2 read
 * buffer0 and write buffer1, iterations 1 and 3 the reverse (:37-43); the result is in
 * buffer0 (:99).  `hist` must hold W*256 uint32 (prepareBuffers :93).
 * `stage_cb` (may be NULL) is called after each stage so tests can snapshot tables:
 *   stage 0 = after histograms, stage 1 = after scatter.
 */
typedef void (*vrs_oracle_stage_cb)(void *user, uint32_t iteration, uint32_t stage);

void vrs_oracle_multi_radixsort_pairs(uint32_t *kbuf0, uint32_t *kbuf1, uint32_t *vbuf0, uint32_t *vbuf1,
                                      uint32_t *hist, uint32_t num_elements, uint32_t blocks_per_workgroup,
                                      vrs_oracle_stage_cb stage_cb, void *user) {
    uint32_t W = vrs_oracle_workgroup_count(num_elements, blocks_per_workgroup);
    for (uint32_t i = 0; i < 4u; ++i) {
        uint32_t shift = 8u * i;
        uint32_t *kin = (i % 2u == 0u) ? kbuf0 : kbuf1;
        uint32_t *kout = (i % 2u == 0u) ? kbuf1 : kbuf0;
        uint32_t *vin = vbuf0 ? ((i % 2u == 0u) ? vbuf0 : vbuf1) : NULL;
        uint32_t *vout = vbuf0 ? ((i % 2u == 0u) ? vbuf1 : vbuf0) : NULL;
        vrs_oracle_histograms(kin, hist, num_elements, shift, W, blocks_per_workgroup);
        if (stage_cb) stage_cb(user, i, 0);
        vrs_oracle_scatter(kin, kout, vin, vout, hist, num_elements, shift, W, blocks_per_workgroup);
        if (stage_cb) stage_cb(user, i, 1);
    }
}

void vrs_oracle_multi_radixsort(uint32_t *buf0, uint32_t *buf1, uint32_t *hist, uint32_t num_elements,
                                uint32_t blocks_per_workgroup) {
    vrs_oracle_multi_radixsort_pairs(buf0, buf1, NULL, NULL, hist, num_elements, blocks_per_workgroup, NULL, NULL);
}

/*
 * single_radixsort path: singleradixsort/resources/shaders/single_radixsort.comp:42-140.
 * One workgroup; per iteration (:47): histogram of the whole input (:50-62), exclusive
 * scan over the 256 bins (:64-86), then block-wise (256 keys per block, :91) stable
 * scatter with the same rank rule as the multi path (:106-137).  Even iterations read
 * buffer0 / write buffer1, odd the reverse (ELEMENT_IN macro :40, writes :129-133);
 * after 4 iterations the result is in buffer0 (SingleRadixSort.cpp:22-23).
 */
void vrs_oracle_single_radixsort(uint32_t *buf0, uint32_t *buf1, uint32_t num_elements) {
    for (uint32_t iteration = 0; iteration < 4u; ++iteration) {
        uint32_t shift = 8u * iteration;
        const uint32_t *in = (iteration % 2u == 0u) ? buf0 : buf1;
        uint32_t *out = (iteration % 2u == 0u) ? buf1 : buf0;
        uint32_t histogram[VRS_RADIX_SORT_BINS];
        memset(histogram, 0, sizeof(histogram));
        for (uint32_t id = 0; id < num_elements; ++id) histogram[(in[id] >> shift) & 255u] += 1u;
        uint32_t global_offsets[VRS_RADIX_SORT_BINS];
        uint32_t run = 0;
        for (uint32_t d = 0; d < VRS_RADIX_SORT_BINS; ++d) {
            global_offsets[d] = run;
            run += histogram[d];
        }
        for (uint32_t block = 0; block < num_elements; block += VRS_WORKGROUP_SIZE) {
            uint32_t block_count[VRS_RADIX_SORT_BINS];
            memset(block_count, 0, sizeof(block_count));
            for (uint32_t l = 0; l < VRS_WORKGROUP_SIZE; ++l) {
                uint32_t id = block + l;
                if (id < num_elements) {
                    uint32_t key = in[id];
                    uint32_t bin = (key >> shift) & 255u;
                    out[global_offsets[bin] + block_count[bin]] = key;
                    block_count[bin] += 1u;
                }
            }
            for (uint32_t d = 0; d < VRS_RADIX_SORT_BINS; ++d) global_offsets[d] += block_count[d];
        }
    }
}

/*
 * std::mt19937 (MT19937, Matsumoto & Nishimura 1998; the C++ standard fixes its parameters
 * and requires the 10000th output of a default-seeded engine to be 4123659995).
 * The reference draws keys from std::mt19937 through
 * uniform_int_distribution<uint32_t>(0, 0x0FFFFFFF) (MultiRadixSort.cpp:121-133), seeded
 * from random_device (not reproducible).  The build's fixtures use the raw 32-bit outputs
 * of mt19937(seed) (full range); `top_bits_zeroed` = 4 reproduces the
 * reference's 28-bit keys (libstdc++ maps that distribution to raw >> 4).
 */
void vrs_oracle_mt19937_fill(uint32_t seed, uint32_t *out, uint64_t n, uint32_t top_bits_zeroed) {
    uint32_t mt[624];
    mt[0] = seed;
    for (uint32_t i = 1; i < 624u; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i;
    uint32_t idx = 624u;
    for (uint64_t k = 0; k < n; ++k) {
        if (idx >= 624u) {
            for (uint32_t i = 0; i < 624u; ++i) {
                uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1u) % 624u] & 0x7fffffffu);
                uint32_t v = mt[(i + 397u) % 624u] ^ (y >> 1);
                if (y & 1u) v ^= 0x9908b0dfu;
                mt[i] = v;
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        out[k] = y >> top_bits_zeroed;
    }
}

/*
 * 64-bit keys: the reference's SORT_64_BIT stub (MultiRadixSort.h:10-18, NUM_ITERATIONS = 8 at
 * MultiRadixSort.cpp:51-55; its generator draws uniform_int_distribution<uint64_t>(0, 0x0FFFFFFFFFFF), :128).
 * The shaders were never adapted ("requires changes in the two shaders"), so these restatements apply the
 * SAME stage definitions as above to uint64 keys and shifts 0..56 -- parity unpinned by any reference run,
 * end-to-end pinned to std::sort of the same uint64 input.
 */
void vrs_oracle_histograms_u64(const uint64_t *keys_in, uint32_t *hist, uint32_t num_elements, uint32_t shift,
                               uint32_t num_workgroups, uint32_t blocks_per_workgroup) {
    for (uint32_t w = 0; w < num_workgroups; ++w) {
        uint32_t *h = hist + (size_t)VRS_RADIX_SORT_BINS * w;
        memset(h, 0, VRS_RADIX_SORT_BINS * sizeof(uint32_t));
        uint64_t begin = (uint64_t)w * blocks_per_workgroup * VRS_WORKGROUP_SIZE;
        uint64_t end = begin + (uint64_t)blocks_per_workgroup * VRS_WORKGROUP_SIZE;
        if (end > num_elements) end = num_elements;
        for (uint64_t e = begin; e < end; ++e) h[(keys_in[e] >> shift) & 255u] += 1u;
    }
}

void vrs_oracle_scatter_u64(const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *values_in,
                            uint32_t *values_out, const uint32_t *hist, uint32_t num_elements, uint32_t shift,
                            uint32_t num_workgroups, uint32_t blocks_per_workgroup) {
    uint32_t *offsets = (uint32_t *)malloc((size_t)num_workgroups * VRS_RADIX_SORT_BINS * sizeof(uint32_t));
    vrs_oracle_offsets(hist, offsets, num_workgroups);
    for (uint32_t w = 0; w < num_workgroups; ++w) {
        uint32_t *global_offsets = offsets + (size_t)VRS_RADIX_SORT_BINS * w;
        uint64_t begin = (uint64_t)w * blocks_per_workgroup * VRS_WORKGROUP_SIZE;
        uint64_t end = begin + (uint64_t)blocks_per_workgroup * VRS_WORKGROUP_SIZE;
        if (end > num_elements) end = num_elements;
        /* blocks of 256 in order, threads in order: the running offset per bin is the stable rank */
        for (uint64_t e = begin; e < end; ++e) {
            uint32_t bin = (uint32_t)(keys_in[e] >> shift) & 255u;
            uint32_t dst = global_offsets[bin]++;
            keys_out[dst] = keys_in[e];
            if (values_in) values_out[dst] = values_in[e];
        }
    }
    free(offsets);
}

void vrs_oracle_multi_radixsort_u64(uint64_t *kbuf0, uint64_t *kbuf1, uint32_t *vbuf0, uint32_t *vbuf1, uint32_t *hist,
                                    uint32_t num_elements, uint32_t blocks_per_workgroup) {
    uint32_t W = vrs_oracle_workgroup_count(num_elements, blocks_per_workgroup);
    for (uint32_t i = 0; i < 8u; ++i) { /* NUM_ITERATIONS = 8, MultiRadixSort.cpp:54 */
        uint64_t *kin = (i % 2u == 0u) ? kbuf0 : kbuf1, *kout = (i % 2u == 0u) ? kbuf1 : kbuf0;
        uint32_t *vin = vbuf0 ? ((i % 2u == 0u) ? vbuf0 : vbuf1) : NULL;
        uint32_t *vout = vbuf0 ? ((i % 2u == 0u) ? vbuf1 : vbuf0) : NULL;
        vrs_oracle_histograms_u64(kin, hist, num_elements, 8u * i, W, blocks_per_workgroup);
        vrs_oracle_scatter_u64(kin, kout, vin, vout, hist, num_elements, 8u * i, W, blocks_per_workgroup);
    }
}
