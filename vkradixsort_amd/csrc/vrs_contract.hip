// vrs_contract.hip -- hand-written CDNA4 (gfx950, wave64) kernels of the reference's stage contract: the multi-block LSD radix sort
// stage by stage (K1-K3), single_radixsort (K4), and the small utility kernels (self-test, placement probe, verification).
// The one-call sort's kernels: vrs_one_call.hip (K5), vrs_msd_hybrid.hip (K5b), vrs_msd_pool.hip.
//
// What the reference computes per pass (VkRadixSort @ v2):
//   multi_radixsort_histograms.comp:31-55   hist[w][d]   = #keys of tile w with digit d
//   multi_radixsort.comp:56-77              offset[w][d] = excl_scan_d(sum_j hist[j][d]) + sum_{j<w} hist[j][d]
//   multi_radixsort.comp:80-126             stable scatter of tile w's keys to offset[w][digit]++
// How it is computed here is NOT how the shaders do it (no per-bin flag masks, no O(W^2) table walk,
// no 4-byte isolated stores): see DESIGN.md "Kernels".
//
//   K1 histogram_kernel   one workgroup per contract tile; 16-byte coalesced loads; LDS counters;
//                         a __ballot vote collapses wave-uniform digits into one LDS add.
//   K2 chunk_sum_kernel + offsets_kernel   two-level prefix over the [W][256] table, O(W*256).
//   K3 scatter_kernel     coalesced tile load (wave-striped), wave64 match-any ranking with
//                         __ballot / mbcnt / __popcll against per-wave LDS digit counters,
//                         LDS re-bucketing, then digit-contiguous global stores.  Stable.
//   K4 single_kernel      the single_radixsort path: four passes inside one workgroup.
#include "vrs_device.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace vrs {

// ---------------------------------------------------------------------------------------------
// K1: per-tile digit histogram.  One workgroup per contract tile, 16-byte coalesced loads, one LDS
// counter per digit.  The kernel must stay HBM-bound (a bare 400 MB read takes ~64 us on this
// chip), so the per-key instruction count matters: 2 VALU + 1 ds_add per key on the plain path.
// Skew guard: a __ballot vote per 4-key vector detects the wave whose 256 keys all carry ONE digit
// (sorted / constant / zero upper bytes -- the reference's own 28-bit keys make pass 3 mostly that)
// and collapses 256 same-address LDS atomics into a single ds_add of the population count.
// (single-key form, used by the single_radixsort kernel)
__device__ __forceinline__ void histogram_count(uint32_t *s_hist, uint32_t key, uint32_t shift, bool valid) {
    const uint32_t d = digit_of(key, shift);
    const uint64_t active = __ballot(valid);
    if (active == 0) return;  // wave-uniform
    const uint32_t first = static_cast<uint32_t>(__ffsll(static_cast<long long>(active))) - 1u;
    const uint32_t d0 = __builtin_amdgcn_readlane(d, first);
    const uint64_t same = __ballot(valid && d == d0);
    if (same == active) {  // wave-uniform: every valid lane votes for the same digit
        if (lane_id() == first) atomicAdd(&s_hist[d0], static_cast<uint32_t>(__popcll(active)));
    } else if (valid) {
        atomicAdd(&s_hist[d], 1u);
    }
}

// all 64 lanes hold a valid 16-byte vector of keys
template <typename K, typename DG>
__device__ __forceinline__ void histogram_count_vec(uint32_t *s_hist, const typename KeyVec<K>::type &q, const DG &dg) {
    constexpr int V = KeyVec<K>::kKeys;
    uint32_t d[V];
#pragma unroll
    for (int i = 0; i < V; ++i) d[i] = dg(KeyVec<K>::get(q, i));
    const uint32_t d0 = __builtin_amdgcn_readfirstlane(d[0]);
    uint32_t diff = 0;
#pragma unroll
    for (int i = 0; i < V; ++i) diff |= d[i] ^ d0;
    if (__ballot(diff == 0u) == ~0ull) {  // wave-uniform branch
        if (lane_id() == 0u) atomicAdd(&s_hist[d0], 64u * V);
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) atomicAdd(&s_hist[d[i]], 1u);
    }
}

template <typename K, int UNROLL, bool SPLIT>
__global__ __launch_bounds__(kThreads) void histogram_kernel(const K *__restrict__ keys,
                                                             uint32_t *__restrict__ hist, uint32_t n,
                                                             uint32_t shift, uint32_t W, uint32_t B,
                                                             const uint32_t *__restrict__ tile_order,
                                                             const K *__restrict__ splitters, uint32_t num_splitters) {
    using Vec = typename KeyVec<K>::type;
    using DG = typename std::conditional<SPLIT, SplitDigit<K>, RadixDigit<K>>::type;
    constexpr uint32_t V = KeyVec<K>::kKeys;
    __shared__ uint32_t s_hist[kBins];
    __shared__ K s_split[SPLIT ? 256 : 1];
    DG dg;
    if constexpr (SPLIT) {
        stage_splitters(s_split, splitters, num_splitters);
        dg.splitters = s_split;
        dg.count = num_splitters;
    } else {
        dg.shift = shift;
    }
    const uint32_t tid = threadIdx.x;
    // which tile this workgroup takes is a pure scheduling choice (cache residency), never a result
    const uint32_t w = tile_order ? tile_order[blockIdx.x] : blockIdx.x;
    s_hist[tid] = 0;
    __syncthreads();

    const uint64_t tile_begin = static_cast<uint64_t>(w) * B * kThreads;
    if (tile_begin < n) {
        const uint64_t tile_keys = static_cast<uint64_t>(B) * kThreads;
        const uint32_t len = static_cast<uint32_t>(tile_begin + tile_keys <= n ? tile_keys : n - tile_begin);
        // 16-byte loads need a 16-byte aligned address; tile_begin is a multiple of 256 keys, so the misalignment
        // is that of the buffer base (a sub-range of a larger allocation may start anywhere): peel `head` keys.
        const K *tile = keys + tile_begin;
        const uint32_t mis = static_cast<uint32_t>((reinterpret_cast<uintptr_t>(tile) / sizeof(K)) % V);
        const uint32_t head = min(mis ? V - mis : 0u, len);
        if (tid < head) atomicAdd(&s_hist[dg(tile[tid])], 1u);
        const Vec *v = reinterpret_cast<const Vec *>(tile + head);
        const uint32_t nvec = (len - head) / V;
        constexpr uint32_t kStep = kThreads * UNROLL;  // vectors per fully unrolled step
        uint32_t i0 = 0;
        const bool stream_in = static_cast<size_t>(n) * sizeof(K) >= kStreamInBytes;  // (below, everything fits the caches: plain loads)
        for (; i0 + kStep <= nvec; i0 += kStep) {  // every lane of every wave holds valid vectors
            Vec q[UNROLL];
            if (stream_in) {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) q[u] = load_stream16(v + i0 + u * kThreads + tid);
            } else {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) q[u] = v[i0 + u * kThreads + tid];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) histogram_count_vec<K>(s_hist, q[u], dg);
        }
        for (uint32_t i = i0 + tid; i < nvec; i += kThreads) {  // ragged remainder of the tile
            const Vec q = v[i];
#pragma unroll
            for (int k = 0; k < static_cast<int>(V); ++k) atomicAdd(&s_hist[dg(KeyVec<K>::get(q, k))], 1u);
        }
        const uint32_t tail = head + nvec * V + tid;  // at most V-1 keys
        if (tail < len) atomicAdd(&s_hist[dg(tile[tail])], 1u);
    }
    __syncthreads();
    hist[static_cast<size_t>(w) * kBins + tid] = s_hist[tid];
}

// ---------------------------------------------------------------------------------------------
// K2: offsets from the [W][256] table in O(W*256) (the reference re-sums the whole table in every
// workgroup: O(W^2*256), multi_radixsort.comp:58-62).  Tiles are grouped in chunks of C rows.
//   chunk_sum_kernel : chunk_sums[g][d] = sum of the hist rows of chunk g
//   offsets_kernel   : base_d = excl_scan_d(sum_g chunk_sums[g][d]); offsets[w][d] = base_d + (rows before w)
// Both are latency-bound (a few MB), so each workgroup is 1024 threads = 4 row groups x 256 digits
// and every thread issues all of its independent row loads before it consumes any.
constexpr int kPrefixThreads = 1024;
constexpr int kPrefixGroups = kPrefixThreads / kBins;

// sum of rows r0, r0+stride, ... < r1 of a [rows][256] table, column d; kDepth independent loads in flight
constexpr int kDepth = 16;
__device__ __forceinline__ uint32_t column_sum(const uint32_t *__restrict__ p, uint32_t r0, uint32_t r1,
                                               uint32_t stride) {
    uint32_t s = 0;
    uint32_t r = r0;
    for (; r + (kDepth - 1) * stride < r1; r += kDepth * stride) {
        uint32_t t[kDepth];
#pragma unroll
        for (int u = 0; u < kDepth; ++u) t[u] = p[static_cast<size_t>(r + u * stride) * kBins];
#pragma unroll
        for (int u = 0; u < kDepth; ++u) s += t[u];
    }
    if (r < r1) {  // remainder: still one batch of predicated independent loads
        uint32_t t[kDepth];
#pragma unroll
        for (int u = 0; u < kDepth; ++u) t[u] = (r + u * stride < r1) ? p[static_cast<size_t>(r + u * stride) * kBins] : 0u;
#pragma unroll
        for (int u = 0; u < kDepth; ++u) s += t[u];
    }
    return s;
}

__global__ __launch_bounds__(kPrefixThreads) void chunk_sum_kernel(const uint32_t *__restrict__ hist,
                                                                   uint32_t *__restrict__ chunk_sums, uint32_t W,
                                                                   uint32_t C) {
    __shared__ uint32_t s_part[kPrefixGroups][kBins];
    const uint32_t d = threadIdx.x & (kBins - 1), grp = threadIdx.x >> 8;
    const uint32_t row0 = blockIdx.x * C;
    const uint32_t rows = min(C, W - row0);
    s_part[grp][d] = column_sum(hist + static_cast<size_t>(row0) * kBins + d, grp, rows, kPrefixGroups);
    __syncthreads();
    if (grp == 0) {
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < kPrefixGroups; ++k) s += s_part[k][d];
        chunk_sums[static_cast<size_t>(blockIdx.x) * kBins + d] = s;
    }
}

__global__ __launch_bounds__(kPrefixThreads) void offsets_kernel(const uint32_t *__restrict__ hist,
                                                                 const uint32_t *__restrict__ chunk_sums,
                                                                 uint32_t *__restrict__ offsets, uint32_t W, uint32_t C,
                                                                 uint32_t G) {
    __shared__ uint32_t s_before[kPrefixGroups][kBins];
    __shared__ uint32_t s_after[kPrefixGroups][kBins];
    __shared__ uint32_t s_quarter[kPrefixGroups][kBins];
    __shared__ uint32_t s_base[kBins];
    __shared__ uint32_t s_tmp[kPrefixThreads / 64];
    const uint32_t d = threadIdx.x & (kBins - 1), grp = threadIdx.x >> 8;
    const uint32_t g = blockIdx.x;
    // (1) chunk totals before this chunk / from this chunk on, split over the 4 row groups
    s_before[grp][d] = column_sum(chunk_sums + d, grp, g, kPrefixGroups);
    s_after[grp][d] = column_sum(chunk_sums + d, g + grp, G, kPrefixGroups);
    // (2) this chunk's rows in 4 contiguous quarters: quarter sums
    const uint32_t row0 = g * C;
    const uint32_t rows = min(C, W - row0);
    const uint32_t per = (rows + kPrefixGroups - 1) / kPrefixGroups;
    const uint32_t q0 = min(grp * per, rows), q1 = min(q0 + per, rows);
    const uint32_t *p = hist + static_cast<size_t>(row0) * kBins + d;
    s_quarter[grp][d] = column_sum(p, q0, q1, 1);
    __syncthreads();
    uint32_t before = 0, total = 0;
    if (grp == 0) {
#pragma unroll
        for (int k = 0; k < kPrefixGroups; ++k) {
            before += s_before[k][d];
            total += s_before[k][d] + s_after[k][d];
        }
    }
    // exclusive scan of the 256 digit totals (waves 0..3 carry them, the rest carry zeros)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = total;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if (lane >= static_cast<uint32_t>(o)) incl += up;
    }
    if (lane == 63u) s_tmp[wave] = incl;
    __syncthreads();
    if (grp == 0) {
        uint32_t base = incl - total;
#pragma unroll
        for (int j = 0; j < 4; ++j) base += (static_cast<uint32_t>(j) < wave) ? s_tmp[j] : 0u;
        s_base[d] = base + before;
    }
    __syncthreads();
    // (3) each quarter walks its rows again (L2-warm) and writes the exclusive offsets
    uint32_t run = s_base[d];
    for (uint32_t k = 0; k < grp; ++k) run += s_quarter[k][d];
    uint32_t *o = offsets + static_cast<size_t>(row0) * kBins + d;
    for (uint32_t r = q0; r < q1; r += kDepth) {
        uint32_t t[kDepth];
#pragma unroll
        for (int u = 0; u < kDepth; ++u) t[u] = (r + u < q1) ? p[static_cast<size_t>(r + u) * kBins] : 0u;
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {
            if (r + u < q1) o[static_cast<size_t>(r + u) * kBins] = run;
            run += t[u];
        }
    }
}

// K2, fused form (default when every chunk workgroup can be resident at once): ONE launch.  Each chunk
// workgroup publishes its 256 digit sums as 8-byte {epoch tag, value} granules with write-through (sc1)
// agent-scope stores, then gathers every chunk's granules by polling them with relaxed agent-scope loads --
// the data is its own flag, so no fence and no placement assumption is involved (a granule is written by
// one aligned 8-byte store).  Saves a kernel boundary and the second read of the chunk sums through HBM.
// Progress never depends on another workgroup: a granule that does not show up within the spin budget is
// recomputed locally from the histogram rows it summarises (written by the previous kernel, hence visible).
__global__ __launch_bounds__(kPrefixThreads) void prefix_fused_kernel(const uint32_t *__restrict__ hist,
                                                                      unsigned long long *granules,
                                                                      uint32_t *__restrict__ offsets, uint32_t W,
                                                                      uint32_t C, uint32_t G, uint32_t epoch) {
    __shared__ uint32_t s_a[kPrefixGroups][kBins];
    __shared__ uint32_t s_b[kPrefixGroups][kBins];
    __shared__ uint32_t s_quarter[kPrefixGroups][kBins];
    __shared__ uint32_t s_base[kBins];
    __shared__ uint32_t s_tmp[kPrefixThreads / 64];
    const uint32_t d = threadIdx.x & (kBins - 1), grp = threadIdx.x >> 8;
    const uint32_t g = blockIdx.x;
    const uint32_t row0 = g * C;
    const uint32_t rows = min(C, W - row0);
    const uint32_t per = (rows + kPrefixGroups - 1) / kPrefixGroups;
    const uint32_t q0 = min(grp * per, rows), q1 = min(q0 + per, rows);
    const uint32_t *p = hist + static_cast<size_t>(row0) * kBins + d;

    // (1) my quarter of this chunk's rows -> quarter sums -> chunk sum, published as granules
    const uint32_t quarter = column_sum(p, q0, q1, 1);
    s_quarter[grp][d] = quarter;
    __syncthreads();
    if (grp == 0) {
        uint32_t a = 0;
#pragma unroll
        for (int k = 0; k < kPrefixGroups; ++k) a += s_quarter[k][d];
        __hip_atomic_store(&granules[static_cast<size_t>(g) * kBins + d],
                           (static_cast<unsigned long long>(epoch) << 32) | a, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }

    // (2) gather: chunk j's digit-d sum for j = grp, grp+4, ... ; `before` = chunks ahead of mine.
    //     (Polling one granule after the other measured faster than sweeping them in batches.)
    uint32_t before = 0, after = 0;
    for (uint32_t j = grp; j < G; j += kPrefixGroups) {
        const unsigned long long *gp = &granules[static_cast<size_t>(j) * kBins + d];
        unsigned long long x = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (uint32_t spin = 0; static_cast<uint32_t>(x >> 32) != epoch && spin < 20000u; ++spin) {
            __builtin_amdgcn_s_sleep(2);
            x = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        uint32_t v;
        if (static_cast<uint32_t>(x >> 32) == epoch) {
            v = static_cast<uint32_t>(x);
        } else {  // never seen in practice: chunk j's workgroup is not running; do its sum ourselves
            const uint32_t r0 = j * C;
            v = column_sum(hist + static_cast<size_t>(r0) * kBins + d, 0, min(C, W - r0), 1);
        }
        if (j < g) before += v;
        else after += v;
    }
    s_a[grp][d] = before;
    s_b[grp][d] = after;
    __syncthreads();
    uint32_t bsum = 0, total = 0;
    if (grp == 0) {
#pragma unroll
        for (int k = 0; k < kPrefixGroups; ++k) {
            bsum += s_a[k][d];
            total += s_a[k][d] + s_b[k][d];
        }
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = total;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if (lane >= static_cast<uint32_t>(o)) incl += up;
    }
    if (lane == 63u) s_tmp[wave] = incl;
    __syncthreads();
    if (grp == 0) {
        uint32_t base = incl - total;
#pragma unroll
        for (int j = 0; j < 4; ++j) base += (static_cast<uint32_t>(j) < wave) ? s_tmp[j] : 0u;
        s_base[d] = base + bsum;
    }
    __syncthreads();

    // (3) each quarter walks its rows again (L2-warm) and writes the exclusive offsets
    uint32_t run = s_base[d];
    for (uint32_t k = 0; k < grp; ++k) run += s_quarter[k][d];
    uint32_t *o = offsets + static_cast<size_t>(row0) * kBins + d;
    for (uint32_t r = q0; r < q1; r += kDepth) {
        uint32_t t[kDepth];
#pragma unroll
        for (int u = 0; u < kDepth; ++u) t[u] = (r + u < q1) ? p[static_cast<size_t>(r + u) * kBins] : 0u;
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {
            if (r + u < q1) o[static_cast<size_t>(r + u) * kBins] = run;
            run += t[u];
        }
    }
}

// the fused form pays off while the gather is short: measured 6 vs 12 us at G = 32 (N = 10^7), a tie at G = 96
constexpr uint32_t kFusedMaxChunks = 48;

template <typename K, int ITEMS, int WAVES, bool PAIRS, int RANK, int OCC, bool SPLIT = false>
__global__ __launch_bounds__(WAVES * 64, OCC) void scatter_kernel(const K *__restrict__ keys_in,
                                                             K *__restrict__ keys_out,
                                                             const uint32_t *__restrict__ values_in,
                                                             uint32_t *__restrict__ values_out,
                                                             const uint32_t *__restrict__ offsets, uint32_t n,
                                                             uint32_t shift, uint32_t W, uint32_t B, int xcd_remap,
                                                             const uint32_t *__restrict__ tile_order,
                                                             uint32_t offset_row_stride,
                                                             const K *__restrict__ splitters, uint32_t num_splitters) {
    using DG = typename std::conditional<SPLIT, SplitDigit<K>, RadixDigit<K>>::type;
    __shared__ ChunkSmem<K, ITEMS, WAVES, PAIRS> sm;
    __shared__ K s_split[SPLIT ? 256 : 1];
    DG dg;
    if constexpr (SPLIT) {
        stage_splitters(s_split, splitters, num_splitters);
        dg.splitters = s_split;
        dg.count = num_splitters;
    } else {
        dg.shift = shift;
    }
    const uint32_t w = tile_order ? tile_order[blockIdx.x]
                                  : (xcd_remap ? xcd_contiguous_tile(blockIdx.x, W) : blockIdx.x);
    const uint64_t tile_begin = static_cast<uint64_t>(w) * B * kThreads;
    if (tile_begin >= n) return;  // uniform per workgroup
    const uint64_t tile_keys = static_cast<uint64_t>(B) * kThreads;
    const uint32_t tile_len = static_cast<uint32_t>(tile_begin + tile_keys <= n ? tile_keys : n - tile_begin);
    // consecutive contract tiles are adjacent in every digit's output range (offset[t+1][d] = offset[t][d] +
    // hist[t][d]), so a launch tile made of `offset_row_stride` contract tiles needs only the first one's row
    uint32_t run_off =
        threadIdx.x < kBins ? offsets[static_cast<size_t>(w) * offset_row_stride * kBins + threadIdx.x] : 0u;
    constexpr uint32_t kChunk = ITEMS * WAVES * 64;
    const bool stream_in = static_cast<size_t>(n) * sizeof(K) >= kStreamInBytes;
    for (uint32_t c0 = 0; c0 < tile_len; c0 += kChunk) {
        const uint32_t valid = min(kChunk, tile_len - c0);
        const K *kin = keys_in + tile_begin + c0;
        const uint32_t *vin = PAIRS ? values_in + tile_begin + c0 : nullptr;
        if (valid == kChunk)  // workgroup-uniform
            scatter_chunk<K, ITEMS, WAVES, PAIRS, RANK, true>(sm, kin, vin, keys_out, values_out, valid, dg, run_off, NoLookback{}, NoPieces{}, stream_in);
        else
            scatter_chunk<K, ITEMS, WAVES, PAIRS, RANK, false>(sm, kin, vin, keys_out, values_out, valid, dg, run_off);
    }
    VRS_MARK_FLUSH();
}


// ---------------------------------------------------------------------------------------------
// Self-test of the property RANK_ATOMIC relies on: for one ds_add_rtn_u32 wave-instruction, lanes
// hitting the same address receive their pre-values in ascending lane order.  Each wave draws
// pseudo-random digits of varying skew, ranks them both ways and counts disagreements.
__global__ __launch_bounds__(kThreads) void atomic_rank_selftest_kernel(uint32_t rounds, uint32_t seed,
                                                                        unsigned long long *mismatches) {
    __shared__ uint32_t s_a[kWaves][kBins];
    __shared__ uint32_t s_b[kWaves][kBins];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (int v = 0; v < kWaves; ++v) {
        s_a[v][tid] = 0;
        s_b[v][tid] = 0;
    }
    __syncthreads();
    uint32_t x = seed ^ (blockIdx.x * 0x9E3779B9u) ^ (tid * 0x85EBCA6Bu);
    unsigned long long bad = 0;
    for (uint32_t r = 0; r < rounds; ++r) {
        x ^= x << 13;
        x ^= x >> 17;
        x ^= x << 5;
        const uint32_t bits = (r + blockIdx.x) % 9u;  // 0..8 significant digit bits: heavy to no skew
        uint32_t d = (x >> 7) & ((1u << bits) - 1u);
        if ((r & 3u) == 3u) d = (d * 32u) & 255u;  // same-bank different-address collisions too
        const uint32_t ra = __hip_atomic_fetch_add(&s_a[wave][d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint64_t peers = match_any_digit(d);
        const uint32_t below = count_below(peers);
        const uint32_t prev = s_b[wave][d];
        const uint32_t rb = prev + below;
        if (below + 1u == static_cast<uint32_t>(__popcll(peers))) s_b[wave][d] = prev + below + 1u;
        __builtin_amdgcn_wave_barrier();
        bad += (ra != rb) ? 1u : 0u;
    }
    if (bad) atomicAdd(mismatches, bad);
}

// ---------------------------------------------------------------------------------------------
// K4: single_radixsort -- one workgroup, four passes in one launch
// (single_radixsort.comp:42-140).  Even passes buffer0 -> buffer1, odd passes back; result in
// buffer0.  Same ranking machinery as K3 with a small chunk.
constexpr int kSingleItems = 4;

__global__ __launch_bounds__(kThreads) void single_kernel(uint32_t *buffer0, uint32_t *buffer1, uint32_t n) {
    __shared__ ChunkSmem<uint32_t, kSingleItems, kWaves> sm;
    __shared__ uint32_t s_hist[kBins];
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t kChunk = kSingleItems * kThreads;
    for (uint32_t iteration = 0; iteration < 4u; ++iteration) {
        const uint32_t shift = 8u * iteration;
        const uint32_t *in = (iteration & 1u) ? buffer1 : buffer0;
        uint32_t *out = (iteration & 1u) ? buffer0 : buffer1;
        s_hist[tid] = 0;
        __syncthreads();
        for (uint32_t i0 = 0; i0 < n; i0 += kThreads) {
            const uint32_t i = i0 + tid;
            const bool ok = i < n;
            histogram_count(s_hist, ok ? in[i] : 0u, shift, ok);
        }
        __syncthreads();
        uint32_t run_off = block_exclusive_scan(s_hist[tid], sm.scan_tmp, tid & 63u, tid >> 6);
        __syncthreads();
        const RadixDigit<uint32_t> dg{shift};
        for (uint32_t c0 = 0; c0 < n; c0 += kChunk) {
            const uint32_t valid = min(kChunk, n - c0);
            if (valid == kChunk)
                scatter_chunk<uint32_t, kSingleItems, kWaves, false, RANK_BALLOT, true>(sm, in + c0, nullptr, out, nullptr, valid,
                                                                              dg, run_off);
            else
                scatter_chunk<uint32_t, kSingleItems, kWaves, false, RANK_BALLOT, false>(sm, in + c0, nullptr, out, nullptr, valid,
                                                                               dg, run_off);
        }
        // the next pass reads what this pass wrote: same CU, so a workgroup barrier (with its
        // workgroup-scope fence) orders the global stores before the loads
        __threadfence_block();
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Contract tiles larger than the 8192-key launch tile (NUM_BLOCKS_PER_WORKGROUP = 64 ... 4096, what the
// reference's own sweeps favour) are histogrammed and scattered as 8192-key sub-tiles; the caller-visible
// [W][256] table is the fold of the sub-tile table:  hist[w][d] = sum_s sub[w*S + s][d].
__global__ __launch_bounds__(kThreads) void fold_histograms_kernel(const uint32_t *__restrict__ sub,
                                                                   uint32_t *__restrict__ hist, uint32_t sub_rows,
                                                                   uint32_t S) {
    const uint32_t d = threadIdx.x;
    const uint32_t r0 = blockIdx.x * S;
    const uint32_t r1 = min(r0 + S, sub_rows);
    hist[static_cast<size_t>(blockIdx.x) * kBins + d] = column_sum(sub + static_cast<size_t>(r0) * kBins + d, 0, r1 - r0, 1);
}

// ---------------------------------------------------------------------------------------------
// Key preprocessing the reference leaves to the integrator ("you have to preprocess negative numbers",
// README.md:154-155): order-preserving bijections between int32 / float32 bit patterns and the uint32
// keys the sort orders.  In place, 16 bytes per lane, grid-stride.
//   mode 0  int32   <-> sortable : flip the sign bit (self-inverse)
//   mode 1  float32  -> sortable : negative: flip all bits, else flip the sign bit (IEEE total order)
//   mode 2  sortable -> float32  : inverse of mode 1
__device__ __forceinline__ uint32_t transform_key(uint32_t x, int mode) {
    if (mode == 0) return x ^ 0x80000000u;
    if (mode == 1) return x ^ ((x & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u);
    return x ^ ((x & 0x80000000u) ? 0x80000000u : 0xFFFFFFFFu);
}

__global__ __launch_bounds__(kThreads) void transform_keys_kernel(uint32_t *keys, uint32_t n, int mode) {
    if (reinterpret_cast<uintptr_t>(keys) & 15u) {  // sub-range of a larger allocation: plain 4-byte accesses
        for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads)
            keys[i] = transform_key(keys[i], mode);
        return;
    }
    uint4 *v = reinterpret_cast<uint4 *>(keys);
    const uint32_t nvec = n >> 2;
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < nvec; i += gridDim.x * kThreads) {
        uint4 q = v[i];
        q.x = transform_key(q.x, mode);
        q.y = transform_key(q.y, mode);
        q.z = transform_key(q.z, mode);
        q.w = transform_key(q.w, mode);
        v[i] = q;
    }
    const uint32_t tail = (nvec << 2) + blockIdx.x * kThreads + threadIdx.x;
    if (blockIdx.x == 0 && tail < n) keys[tail] = transform_key(keys[tail], mode);
}

// ---------------------------------------------------------------------------------------------
// On-device counterpart of MultiRadixSort::verify / testSort (MultiRadixSort.cpp:97-102,148-161) for batches too
// many or too large to download: out[0] = number of positions i with keys[i] > keys[i+1] (0 == ascending),
// out[1] = sum of the keys, out[2] = sum of a 64-bit mix of every key (both order-independent: equal before and after a
// sort iff -- up to hash collisions -- the output is a permutation of the input).
__device__ __forceinline__ unsigned long long mix_key(uint32_t k) {
    unsigned long long x = (static_cast<unsigned long long>(k) + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
    x ^= x >> 29;
    return x * 0x94D049BB133111EBull;
}
__global__ __launch_bounds__(kThreads) void verify_keys_kernel(const uint32_t *__restrict__ keys, uint32_t n,
                                                               unsigned long long *__restrict__ out) {
    unsigned long long inv = 0, sum = 0, mix = 0;
    const size_t stride = static_cast<size_t>(gridDim.x) * kThreads, t = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x;
    // 16-byte loads behind a scalar head (the buffer may be a 4-byte aligned sub-range); the element after a vector
    // is one extra cached 4-byte load
    const uint32_t mis = static_cast<uint32_t>((reinterpret_cast<uintptr_t>(keys) / 4u) % 4u);
    const uint32_t head = min(mis ? 4u - mis : 0u, n);
    if (t < head) {
        const uint32_t k = keys[t];
        if (t + 1 < n && k > keys[t + 1]) ++inv;
        sum += k;
        mix += mix_key(k);
    }
    const uint4 *v = reinterpret_cast<const uint4 *>(keys + head);
    const size_t nvec = (n - head) / 4u;
    for (size_t i = t; i < nvec; i += stride) {
        const uint4 q = v[i];
        const size_t next = head + 4u * i + 4u;
        inv += (q.x > q.y) + (q.y > q.z) + (q.z > q.w) + (next < n && q.w > keys[next] ? 1u : 0u);
        sum += static_cast<unsigned long long>(q.x) + q.y + q.z + q.w;
        mix += mix_key(q.x) + mix_key(q.y) + mix_key(q.z) + mix_key(q.w);
    }
    const size_t tail = head + 4u * nvec + t;  // at most 3 keys
    if (tail < n) {
        const uint32_t k = keys[tail];
        if (tail + 1 < n && k > keys[tail + 1]) ++inv;
        sum += k;
        mix += mix_key(k);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        inv += __shfl_down(inv, o);
        sum += __shfl_down(sum, o);
        mix += __shfl_down(mix, o);
    }
    if ((threadIdx.x & 63u) == 0u) {
        if (inv) atomicAdd(&out[0], inv);
        atomicAdd(&out[1], sum);
        atomicAdd(&out[2], mix);
    }
}

hipError_t launch_verify_keys(hipStream_t stream, const uint32_t *keys, uint32_t n, unsigned long long *out3) {
    if (n == 0) return hipSuccess;
    const uint32_t blocks = min((n / 4u + kThreads - 1) / kThreads + 1u, 4096u);
    hipLaunchKernelGGL(verify_keys_kernel, dim3(blocks), dim3(kThreads), 0, stream, keys, n, out3);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// host-side launch wrappers

hipError_t launch_fold_histograms(hipStream_t stream, const uint32_t *sub, uint32_t *hist, uint32_t sub_rows,
                                  uint32_t W, uint32_t S, LaunchEvents ev) {
    if (W == 0) return hipSuccess;
    VRS_LAUNCH(fold_histograms_kernel, dim3(W), dim3(kThreads), stream, ev, sub, hist, sub_rows, S);
    return hipGetLastError();
}

hipError_t launch_transform_keys(hipStream_t stream, uint32_t *keys, uint32_t n, int mode) {
    if (n == 0) return hipSuccess;
    const uint32_t blocks = min((n / 4 + kThreads - 1) / kThreads + 1, 4096u);
    hipLaunchKernelGGL(transform_keys_kernel, dim3(blocks), dim3(kThreads), 0, stream, keys, n, mode);
    return hipGetLastError();
}

uint32_t prefix_chunk_tiles(uint32_t W) {
    uint32_t c = 1;
    while (static_cast<uint64_t>(c) * c < W) c <<= 1;
    return c;
}

hipError_t launch_histograms(hipStream_t stream, const void *keys_in, uint32_t *hist, uint32_t n, uint32_t shift,
                             uint32_t W, uint32_t B, LaunchEvents ev, const uint32_t *tile_order, int key_bytes,
                             const void *splitters, uint32_t num_splitters) {
    if (W == 0) return hipSuccess;
    if (splitters != nullptr) {  // range partition (uint32 keys): bucket = number of splitters <= key
        if (key_bytes != 4 || num_splitters > 255) return hipErrorInvalidValue;
        VRS_LAUNCH((histogram_kernel<uint32_t, 8, true>), dim3(W), dim3(kThreads), stream, ev,
                   static_cast<const uint32_t *>(keys_in), hist, n, shift, W, B, tile_order,
                   static_cast<const uint32_t *>(splitters), num_splitters);
    } else if (key_bytes == 8) {
        VRS_LAUNCH((histogram_kernel<uint64_t, 8, false>), dim3(W), dim3(kThreads), stream, ev,
                   static_cast<const uint64_t *>(keys_in), hist, n, shift, W, B, tile_order,
                   static_cast<const uint64_t *>(nullptr), 0u);
    } else {
        VRS_LAUNCH((histogram_kernel<uint32_t, 8, false>), dim3(W), dim3(kThreads), stream, ev,
                   static_cast<const uint32_t *>(keys_in), hist, n, shift, W, B, tile_order,
                   static_cast<const uint32_t *>(nullptr), 0u);
    }
    return hipGetLastError();
}

hipError_t launch_prefix(hipStream_t stream, const uint32_t *hist, const PrefixScratch &scratch, uint32_t W,
                         LaunchEvents ev) {
    if (W == 0) return hipSuccess;
    const uint32_t C = prefix_chunk_tiles(W);
    const uint32_t G = (W + C - 1) / C;
    if (scratch.granules != nullptr && scratch.fused_max_chunks >= G && G <= kFusedMaxChunks) {
        // one launch; every chunk workgroup is resident at once (G <= compute units), none waits on an unscheduled one
        VRS_LAUNCH(prefix_fused_kernel, dim3(G), dim3(kPrefixThreads), stream, ev, hist, scratch.granules,
                   scratch.offsets, W, C, G, scratch.epoch);
        return hipGetLastError();
    }
    const LaunchEvents first{ev.start, nullptr}, second{nullptr, ev.stop};
    VRS_LAUNCH(chunk_sum_kernel, dim3(G), dim3(kPrefixThreads), stream, first, hist, scratch.chunk_sums, W, C);
    VRS_LAUNCH(offsets_kernel, dim3(G), dim3(kPrefixThreads), stream, second, hist, scratch.chunk_sums,
               scratch.offsets, W, C, G);
    return hipGetLastError();
}

template <typename K, int ITEMS, int WAVES, int RANK, int OCC>
static hipError_t launch_scatter_variant(hipStream_t stream, const void *keys_in, void *keys_out,
                                         const uint32_t *values_in, uint32_t *values_out, const uint32_t *offsets,
                                         uint32_t n, uint32_t shift, uint32_t W, uint32_t B, bool xcd_remap,
                                         LaunchEvents ev, const uint32_t *tile_order, uint32_t offset_row_stride) {
    const int remap = xcd_remap ? 1 : 0;
    const K *kin = static_cast<const K *>(keys_in);
    K *kout = static_cast<K *>(keys_out);
    const K *no_split = nullptr;
    if (values_in != nullptr)
        VRS_LAUNCH((scatter_kernel<K, ITEMS, WAVES, true, RANK, OCC>), dim3(W), dim3(WAVES * 64), stream, ev, kin, kout,
                   values_in, values_out, offsets, n, shift, W, B, remap, tile_order, offset_row_stride, no_split, 0u);
    else
        VRS_LAUNCH((scatter_kernel<K, ITEMS, WAVES, false, RANK, OCC>), dim3(W), dim3(WAVES * 64), stream, ev, kin, kout,
                   values_in, values_out, offsets, n, shift, W, B, remap, tile_order, offset_row_stride, no_split, 0u);
    return hipGetLastError();
}

hipError_t launch_range_partition(hipStream_t stream, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *offsets,
                                  uint32_t n, uint32_t W, bool xcd_remap, bool atomic_rank, const uint32_t *splitters,
                                  uint32_t num_splitters, LaunchEvents ev) {
    if (W == 0) return hipSuccess;
    if (num_splitters > 255) return hipErrorInvalidValue;
    const int remap = xcd_remap ? 1 : 0;
    const uint32_t *no_values = nullptr;
    uint32_t *no_values_out = nullptr;
    const uint32_t *no_order = nullptr;
    if (atomic_rank)
        VRS_LAUNCH((scatter_kernel<uint32_t, 16, 8, false, RANK_ATOMIC, 4, true>), dim3(W), dim3(512), stream, ev, keys_in,
                   keys_out, no_values, no_values_out, offsets, n, 0u, W, 32u, remap, no_order, 1u, splitters, num_splitters);
    else
        VRS_LAUNCH((scatter_kernel<uint32_t, 16, 8, false, RANK_BALLOT, 4, true>), dim3(W), dim3(512), stream, ev, keys_in,
                   keys_out, no_values, no_values_out, offsets, n, 0u, W, 32u, remap, no_order, 1u, splitters, num_splitters);
    return hipGetLastError();
}

#define VRS_SCATTER_ARGS \
    stream, keys_in, keys_out, values_in, values_out, offsets, n, shift, W, B, xcd_remap, ev, tile_order, offset_row_stride

hipError_t launch_scatter(hipStream_t stream, const void *keys_in, void *keys_out, const uint32_t *values_in,
                          uint32_t *values_out, const uint32_t *offsets, uint32_t n, uint32_t shift, uint32_t W,
                          uint32_t B, bool xcd_remap, const ScatterLaunch &cfg, LaunchEvents ev,
                          const uint32_t *tile_order, uint32_t offset_row_stride, int key_bytes) {
    if (W == 0) return hipSuccess;
    const int rank = cfg.atomic_rank ? RANK_ATOMIC : RANK_BALLOT;
    if (key_bytes == 8) {
        // uint64 keys: 4096-key chunks keep the LDS footprint of the uint32 path (32 KiB of keys)
        if (B >= 16)
            return rank == RANK_ATOMIC ? launch_scatter_variant<uint64_t, 8, 8, RANK_ATOMIC, 4>(VRS_SCATTER_ARGS)
                                       : launch_scatter_variant<uint64_t, 8, 8, RANK_BALLOT, 4>(VRS_SCATTER_ARGS);
        return launch_scatter_variant<uint64_t, 4, 4, RANK_BALLOT, 4>(VRS_SCATTER_ARGS);
    }
    // chunk = ITEMS*WAVES*64 keys held in registers + LDS at once; a tile of B blocks is walked in
    // ceil(B*256/chunk) chunks.  cfg.variant (tuning only) = OCC*100000 + ITEMS*1000 + WAVES*10 + RANK
    // (OCC = waves per SIMD the register allocation is held to); 0 = default for this B and rank mode.
    int variant = cfg.variant;
    if (variant == 0) {
        if (B >= 32) variant = 416080 + rank;       // 8192-key chunks, 512 threads
        else if (B >= 16) variant = 416040 + rank;  // 4096-key chunks
        else if (B >= 8) variant = 408040;
        else variant = 404040;
    }
    switch (variant) {
        case 416080: return launch_scatter_variant<uint32_t, 16, 8, RANK_BALLOT, 4>(VRS_SCATTER_ARGS);
        case 416081: return launch_scatter_variant<uint32_t, 16, 8, RANK_ATOMIC, 4>(VRS_SCATTER_ARGS);
        case 416040: return launch_scatter_variant<uint32_t, 16, 4, RANK_BALLOT, 4>(VRS_SCATTER_ARGS);
        case 416041: return launch_scatter_variant<uint32_t, 16, 4, RANK_ATOMIC, 4>(VRS_SCATTER_ARGS);
        case 408040: return launch_scatter_variant<uint32_t, 8, 4, RANK_BALLOT, 4>(VRS_SCATTER_ARGS);
        case 404040: return launch_scatter_variant<uint32_t, 4, 4, RANK_BALLOT, 4>(VRS_SCATTER_ARGS);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_atomic_rank_selftest(hipStream_t stream, uint32_t rounds, uint32_t seed,
                                       unsigned long long *mismatches) {
    hipLaunchKernelGGL(atomic_rank_selftest_kernel, dim3(1024), dim3(kThreads), 0, stream, rounds, seed, mismatches);
    return hipGetLastError();
}

__global__ void xcc_probe_kernel(uint32_t *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

hipError_t launch_xcc_probe(hipStream_t stream, uint32_t *out, uint32_t blocks) {
    hipLaunchKernelGGL(xcc_probe_kernel, dim3(blocks), dim3(512), 0, stream, out);
    return hipGetLastError();
}

hipError_t launch_single(hipStream_t stream, uint32_t *buffer0, uint32_t *buffer1, uint32_t n, LaunchEvents ev) {
    if (n == 0) return hipSuccess;
    VRS_LAUNCH(single_kernel, dim3(1), dim3(kThreads), stream, ev, buffer0, buffer1, n);
    return hipGetLastError();
}


}  // namespace vrs
