// vrs_kernels.hip -- hand-written CDNA4 (gfx950, wave64) kernels of the multi-block LSD radix sort.
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
#include "vrs_local_sort.hpp"

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
        for (; i0 + kStep <= nvec; i0 += kStep) {  // every lane of every wave holds valid vectors
            Vec q[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) q[u] = v[i0 + u * kThreads + tid];
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
    for (uint32_t c0 = 0; c0 < tile_len; c0 += kChunk) {
        const uint32_t valid = min(kChunk, tile_len - c0);
        const K *kin = keys_in + tile_begin + c0;
        const uint32_t *vin = PAIRS ? values_in + tile_begin + c0 : nullptr;
        if (valid == kChunk)  // workgroup-uniform
            scatter_chunk<K, ITEMS, WAVES, PAIRS, RANK, true>(sm, kin, vin, keys_out, values_out, valid, dg, run_off);
        else
            scatter_chunk<K, ITEMS, WAVES, PAIRS, RANK, false>(sm, kin, vin, keys_out, values_out, valid, dg, run_off);
    }
    VRS_MARK_FLUSH();
}

// ---------------------------------------------------------------------------------------------
// K5: the one-call sort (vrs_sort_keys_u32 / _u64 / vrs_sort_pairs_u32) for large N: 36 instead of 48 bytes per key.
//
// The contract path reads the keys once per pass just to count them, because the [W][256] table is part of the
// reference's interface.  When the library owns all four passes it can count ONCE, before the first pass, and let
// every scatter pass find its offsets by decoupled look-back.  A single look-back chain over all tiles does not
// fit this chip (tiles must stay in XCD-contiguous order for the L2s to merge their partial lines, and 500-750
// resident tiles finish 12 ns apart while a hand-off between workgroups takes 1-3 us under load), so the tiles of a
// pass are cut into kStreams independent STREAMS -- one per XCD -- whose starting offsets are known before the pass
// starts.  A stream is a run of neighbouring GROUPS; a key's group is a function of the key alone:
//   pass 0   group g = the g-th slice of the input (whole tiles of 8192 uint32 / 4096 uint64 keys);
//   pass p>0 group g = the keys whose digit p-1 lies in [g * 256/G, (g+1) * 256/G): after pass p-1 they are the
//            contiguous range [P_{p-1}[g * 256/G], P_{p-1}[(g+1) * 256/G]) of its output (P = exclusive digit prefix),
//            whatever their order inside.
// digit_tables_kernel counts, in one read of the keys, H[p][g][d] = #keys of group g of pass p with digit p == d;
// plan_kernel merges the G groups of each pass into kStreams streams of nearly equal length and turns H into their
// ranges and seeds (P_p[d] + the keys with digit d in the groups before the stream's first group);
// onesweep_scatter_kernel walks stream s in tile order on XCD s % 8 (tiles of one stream are neighbours in that L2)
// and looks back only along its own stream.
// Streams follow the data: a pass whose streams cannot be balanced (one group holds far more than 1/kStreams of the
// keys: keys that are all multiples of 256, say) is marked in the plan and run through the contract path instead.

// ---- K5b, hybrid form of the one-call sort for uint32 keys (28 instead of 36 bytes per key): an MSD partition by the top
// kMsdBits bits in two look-back scatter passes (8 + 6 bits), then every bucket (about N / 16384 keys) is sorted by its
// low 18 bits inside ONE workgroup's LDS and written back once.
// (kMsdBits, kMsdSub, ... : vrs_device.hpp)
constexpr int kLocalThreads = 256, kLocalItems = 26;                          // local sort: capacity 6656 keys per bucket
constexpr uint32_t kLocalCap = kLocalThreads * kLocalItems;

// fused form of the counting read: the last workgroup to finish also makes the plan (plan == nullptr: separate kernel)
struct FusedPlanArgs {
    OnesweepPlan *plan;
    OnesweepPlanHead *host_head;
    uint32_t *done;  // ticket counter, zero between launches
    uint32_t stamp, tile, tile_cap, blind_cap;
    StreamCuts cuts0;
};

// LDS row of one group's 256 counters, padded by one word: keys that share the counted digit but not the group
// (sorted input) would otherwise hit one LDS bank from every lane
constexpr int kTableRow = kBins + 1;
// The pass-0 table has only 256 counters, hit by every key of the workgroup: it is kept in COPIES copies, lane l
// adding to copy l % COPIES (word d0 * COPIES + copy, so the lanes of a half wave spread over COPIES banks whatever
// their digits; 32 copies = conflict-free).
template <int GROUPS, int COPIES>
struct TableIndex {
    static constexpr int kShift = GROUPS == 32 ? 3 : GROUPS == 16 ? 4 : 5;  // log2(256 / GROUPS)
    // word of pass-0 digit / of the joint (group of digit p-1, digit p) counter
    static __device__ __forceinline__ uint32_t t0(uint32_t w, uint32_t lane) { return (w & 255u) * COPIES + (lane % COPIES); }
    static __device__ __forceinline__ uint32_t t1(uint32_t w) { return ((w & 255u) >> kShift) * kTableRow + ((w >> 8) & 255u); }
    static __device__ __forceinline__ uint32_t t2(uint32_t w) { return (((w >> 8) & 255u) >> kShift) * kTableRow + ((w >> 16) & 255u); }
    static __device__ __forceinline__ uint32_t t3(uint32_t w) { return (((w >> 16) & 255u) >> kShift) * kTableRow + (w >> 24); }
};

// V counters per lane (one 16-byte vector of keys).  Same-address lanes of one LDS atomic are served one after the
// other, so input with few distinct counters per wave (constant bytes, sorted or clustered keys) would crawl.  When
// (nearly) every lane's keys share a counter -- the signature of such input -- the wave adds once per RUN of equal
// counters across its lanes instead of once per key; uniform-random keys fail the vote at once and take the plain path.
// clustered: the vote, taken by the caller on the first vector of a step (it is a speed heuristic only: both forms
// count every key exactly once).
template <int V>
__device__ __forceinline__ bool table_vote(const uint32_t (&idx)[V]) {
    bool same = true;
#pragma unroll
    for (int j = 1; j < V; ++j) same = same && idx[j] == idx[0];
    return __popcll(__ballot(same)) >= 48;  // wave-uniform
}
template <int V>
__device__ __forceinline__ void table_add(uint32_t *t, const uint32_t (&idx)[V], uint32_t lane, bool clustered) {
    if (clustered) {  // wave-uniform
        // run-length aggregation across the lanes: the first lane of every run of equal counters adds the whole run.
        // (Equal counters in different runs just add twice: always correct, best on sorted / clustered input.)
        bool same = true;
#pragma unroll
        for (int j = 1; j < V; ++j) same = same && idx[j] == idx[0];
        const uint64_t uniform_lanes = __ballot(same);
        const uint32_t mine = same ? idx[0] : 0xFFFFFFFFu;  // lanes that straddle two counters break the runs
        const uint32_t prev = __shfl_up(mine, 1);
        const bool head = same && (lane == 0u || prev != mine);
        const uint64_t breaks = __ballot(head) | ~uniform_lanes;
        const uint64_t after = lane == 63u ? 0ull : breaks >> (lane + 1u);
        const uint32_t run = after ? static_cast<uint32_t>(__ffsll(static_cast<long long>(after))) : 64u - lane;
        if (head) atomicAdd(&t[idx[0]], static_cast<uint32_t>(V) * run);
        if (!same) {
#pragma unroll
            for (int j = 0; j < V; ++j) atomicAdd(&t[idx[j]], 1u);
        }
    } else {
#pragma unroll
        for (int j = 0; j < V; ++j) atomicAdd(&t[idx[j]], 1u);
    }
}

// the 32-bit word of a key that holds the four digits of this group of passes (bits [base_shift, base_shift + 32))
__device__ __forceinline__ uint32_t digit_word(uint32_t key, uint32_t) { return key; }
__device__ __forceinline__ uint32_t digit_word(uint64_t key, uint32_t base_shift) {
    return static_cast<uint32_t>(key >> base_shift);
}

// hm: the 16384-bin histogram of the key's top 14 bits (hybrid form, K5b), or nullptr; t0: nullptr = no LSD tables
template <typename TI>
__device__ __forceinline__ void digit_tables_count(uint32_t *t0, uint32_t *t1, uint32_t *t2, uint32_t *t3, uint32_t *hm,
                                                   uint32_t msd_shift, uint32_t msd_base, uint32_t &msd_over, uint32_t w) {
    if (t0) {  // workgroup-uniform: nullptr when only the bucket histogram is counted (hybrid form, fast count)
        atomicAdd(&t0[TI::t0(w, lane_id())], 1u);
        atomicAdd(&t1[TI::t1(w)], 1u);
        atomicAdd(&t2[TI::t2(w)], 1u);
        atomicAdd(&t3[TI::t3(w)], 1u);
    }
    if (hm) {
        const uint32_t b = (w - msd_base) >> msd_shift;
        msd_over |= (b >> kMsdBits) | (w < msd_base ? 1u : 0u);  // a key above the probed range (or below the promised floor): the plan will refuse the hybrid form
        atomicAdd(&hm[min(b, kMsdBuckets - 1u)], 1u);
    }
}

// one 16-byte vector of keys per lane: 4 uint32 or 2 uint64.  vote: bit t = table t takes the run-length form
template <typename K, typename TI, bool VOTE, bool MSD>
__device__ __forceinline__ void digit_tables_count_vec(uint32_t *t0, uint32_t *t1, uint32_t *t2, uint32_t *t3, uint32_t *hm,
                                                       uint32_t msd_shift, uint32_t msd_base, uint32_t &msd_over,
                                                       const typename KeyVec<K>::type &q, uint32_t base_shift,
                                                       uint32_t lane, uint32_t &vote) {
    constexpr int V = KeyVec<K>::kKeys;
    if constexpr (MSD) {
        if (t0 == nullptr) {  // only the bucket histogram (workgroup-uniform; hm != nullptr then)
            uint32_t im[V];
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const uint32_t w = digit_word(KeyVec<K>::get(q, j), base_shift);
                const uint32_t b = (w - msd_base) >> msd_shift;
                msd_over |= (b >> kMsdBits) | (w < msd_base ? 1u : 0u);
                im[j] = min(b, kMsdBuckets - 1u);
            }
            if constexpr (VOTE) vote = table_vote<V>(im) ? 16u : 0u;
            table_add<V>(hm, im, lane, (vote & 16u) != 0u);
            return;
        }
    }
    uint32_t i0[V], i1[V], i2[V], i3[V], im[MSD ? V : 1];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const uint32_t w = digit_word(KeyVec<K>::get(q, j), base_shift);
        i0[j] = TI::t0(w, lane);
        i1[j] = TI::t1(w);
        i2[j] = TI::t2(w);
        i3[j] = TI::t3(w);
        if constexpr (MSD) {
            const uint32_t b = (w - msd_base) >> msd_shift;
            msd_over |= (b >> kMsdBits) | (w < msd_base ? 1u : 0u);
            im[j] = min(b, kMsdBuckets - 1u);
        }
    }
    if constexpr (VOTE) {
        vote = (table_vote<V>(i0) ? 1u : 0u) | (table_vote<V>(i1) ? 2u : 0u) | (table_vote<V>(i2) ? 4u : 0u) |
               (table_vote<V>(i3) ? 8u : 0u);
        if constexpr (MSD) {
            if (hm) vote |= table_vote<V>(im) ? 16u : 0u;
        }
    }
    table_add<V>(t0, i0, lane, (vote & 1u) != 0u);
    table_add<V>(t1, i1, lane, (vote & 2u) != 0u);
    table_add<V>(t2, i2, lane, (vote & 4u) != 0u);
    table_add<V>(t3, i3, lane, (vote & 8u) != 0u);
    if constexpr (MSD) {
        if (hm) table_add<V>(hm, im, lane, (vote & 16u) != 0u);  // workgroup-uniform: nullptr when the key range is too narrow
    }
}

// one workgroup; thread (p, d).  Merges the GROUPS groups of every pass into kStreams streams of (nearly) equal
// length -- cuts only between groups, so a stream is still a contiguous range of the pass's input and its seed is a
// prefix over whole groups -- and leaves `tables` zeroed for the next sort.  The head goes to device memory (the
// scatter workgroups read their stream from it) and, with system-scope stores, to the pinned host copy (stamp last).
// A device function of one 1024-thread workgroup: the standalone plan kernel, or the tail of the counting read's LAST
// workgroup (fused form: no kernel of its own).  The tables are read with agent-scope loads -- in the fused form they
// were written by other workgroups' atomics in the same launch.
template <int GROUPS>
__device__ __forceinline__ void plan_body(uint32_t *__restrict__ tables, OnesweepPlan *__restrict__ plan,
                                          OnesweepPlanHead *host_head, uint32_t stamp, uint32_t n, uint32_t group_len,
                                          uint32_t tile, uint32_t tile_cap, uint32_t blind_cap, const StreamCuts &cuts0) {
    constexpr uint32_t kGroupDigits = kBins / GROUPS;  // digit values of pass p-1 per group of pass p
    __shared__ uint32_t s_prefix[4][kBins + 1];
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_max[4], s_const[4];
    __shared__ uint32_t s_cut[4][kStreams + 1];  // first group of every stream
    __shared__ OnesweepPlanHead s_head;
    const uint32_t tid = threadIdx.x, p = tid >> 8, d = tid & 255u, lane = tid & 63u, wave = tid >> 6;
    uint32_t before[GROUPS];
    uint32_t total = 0;
#pragma unroll
    for (int g = 0; g < GROUPS; ++g)
        before[g] = __hip_atomic_load(&tables[(static_cast<size_t>(p) * GROUPS + g) * kBins + d], __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
        const uint32_t c = before[g];
        before[g] = total;
        total += c;
        tables[(static_cast<size_t>(p) * GROUPS + g) * kBins + d] = 0;
    }
    uint32_t incl = total;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t x = __shfl_up(incl, o);
        if (lane >= static_cast<uint32_t>(o)) incl += x;
    }
    if (lane == 63u) s_wave[wave] = incl;
    if (tid < 4) {
        s_max[tid] = 0;
        s_const[tid] = 0;
    }
    __syncthreads();
    if (total == n) s_const[p] = 1;  // one digit value holds every key
    uint32_t base = 0;
    for (uint32_t j = p * 4u; j < wave; ++j) base += s_wave[j];
    const uint32_t digit_start = base + incl - total;
    s_prefix[p][d] = digit_start;
    if (d == 255u) s_prefix[p][kBins] = n;
    __syncthreads();
    // where group g of pass q starts in the pass's input
    const auto group_start = [&](uint32_t q, uint32_t g) -> uint32_t {
        if (q == 0) {
            const uint64_t a = static_cast<uint64_t>(g) * group_len;
            return static_cast<uint32_t>(a < n ? a : n);
        }
        return s_prefix[q - 1][g * kGroupDigits];  // g == GROUPS -> n
    };
    if (tid < 4u * kStreams) {  // thread (q, k): the cut between streams k-1 and k of pass q
        const uint32_t q = tid / kStreams, k = tid % kStreams;
        uint32_t cut = 0;
        if (q == 0)  // slices of the input: the host made these cuts (it sizes pass 0's grid from them)
            cut = cuts0.first_group[k];
        else if (k > 0)
            cut = balanced_cut([&](uint32_t g) { return s_prefix[q - 1][g * kGroupDigits]; }, n, k, GROUPS);  // [GROUPS] -> n
        s_cut[q][k] = cut;
        if (k == 0) s_cut[q][kStreams] = GROUPS;
    }
    // where digit d of every group starts in the pass's output; a stream's seed is the row of its first group
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) plan->group_seed[p][g][d] = digit_start + before[g];
    plan->group_seed[p][GROUPS][d] = digit_start + total;
    __syncthreads();
    if (tid < 4u * kStreams) {
        const uint32_t q = tid / kStreams, s = tid % kStreams;
        const uint32_t start = group_start(q, s_cut[q][s]), end = group_start(q, s_cut[q][s + 1]);
        const uint32_t tiles = (end - start + tile - 1u) / tile;
        s_head.stream[q][s] = StreamDesc{start, end - start, s_cut[q][s], tiles};
        atomicMax(&s_max[q], tiles);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t first = 4;
        for (int q = 3; q >= 0; --q) {
            // pass 0's grid is sized by the host from its own cuts; passes 1-3 were enqueued with blind_cap rows
            const uint32_t mode = s_const[q]                       ? kPassIdentity
                                  : s_max[q] > tile_cap            ? kPassUnbalanced
                                  : (q > 0 && s_max[q] > blind_cap) ? kPassLookbackWide
                                                                    : kPassLookback;
            s_head.max_tiles[q] = s_max[q];
            s_head.mode[q] = mode;
            if (mode != kPassLookback) first = static_cast<uint32_t>(q);
        }
        s_head.first_abnormal = first;
        s_head.msd_ok = 0;  // the hybrid form's fields: msd_plan_kernel fills them in when it runs
        s_head.msd_tiles_b = 0;
        s_head.msd_max_bucket = 0;
        s_head.msd_shift_a = 0;
        s_head.lsd_missing = 0;
        s_head.ready = 0;
    }
    __syncthreads();
    if (tid < 4u * kStreams) {  // what a speculatively enqueued pass sees: no tiles from the first abnormal pass on
        const uint32_t q = tid / kStreams, k = tid % kStreams;
        StreamDesc d = s_head.stream[q][k];
        if (q >= s_head.first_abnormal) d.tiles = 0;
        s_head.blind[q][k] = d;
    }
    __syncthreads();
    constexpr uint32_t kHeadWords = sizeof(OnesweepPlanHead) / sizeof(uint32_t);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&s_head);
    for (uint32_t i = tid; i < kHeadWords - 1u; i += 4 * kBins)  // every word but `ready` (the last one)
        reinterpret_cast<uint32_t *>(&plan->head)[i] = src[i];
    if (host_head && tid < 64u) {  // ONE wave writes the host copy, so one wave's fence orders it before the stamp
        for (uint32_t i = tid; i < kHeadWords - 1u; i += 64u)
            __hip_atomic_store(reinterpret_cast<uint32_t *>(host_head) + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        // stamp == 0: another kernel (msd_plan_kernel) completes the head and stamps it
        if (tid == 0 && stamp != 0u) __hip_atomic_store(&host_head->ready, stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int GROUPS>
__global__ __launch_bounds__(4 * kBins) void plan_kernel(uint32_t *__restrict__ tables, OnesweepPlan *__restrict__ plan,
                                                        OnesweepPlanHead *host_head, uint32_t stamp, uint32_t n,
                                                        uint32_t group_len, uint32_t tile, uint32_t tile_cap,
                                                        uint32_t blind_cap, StreamCuts cuts0) {
    plan_body<GROUPS>(tables, plan, host_head, stamp, n, group_len, tile, tile_cap, blind_cap, cuts0);
}


// grid = GROUPS * slices workgroups; workgroup (s, g) counts the g-th part of pass-0 group s and zeroes its share
// of the look-back status words.  group_len (the length of a pass-0 group) is a multiple of 4 * slices.
// The loads run one step ahead of the counting, vector by vector (a vector's register is refilled for the next step
// as soon as it has been consumed), so UNROLL 16-byte loads per lane are in flight all the time.  64-bit keys are sorted
// in two groups of four passes, each with its own counting read: base_shift = 0, then 32.
// MSD (hybrid form, K5b; uint32 keys, GROUPS == 8): the same read also fills a 16384-bin histogram of the top 14 bits
// (msd_hist) and, per pass-0 group, the 256 top-byte counts the MSD pass needs as its streams' seeds (msd_slices).
template <typename K, int GROUPS, int THREADS, int COPIES, int UNROLL, int OCC, bool MSD = false>
__global__ __launch_bounds__(THREADS, OCC) void digit_tables_kernel(const K *__restrict__ keys, uint32_t n,
                                                                    uint32_t base_shift, uint32_t group_len,
                                                                    uint32_t slices, uint32_t *__restrict__ tables,
                                                                    uint4 *__restrict__ status, uint32_t status_vecs,
                                                                    FusedPlanArgs fp, uint32_t *__restrict__ msd_hist,
                                                                    uint32_t *__restrict__ msd_slices, uint32_t msd_only,
                                                                    uint32_t msd_base, uint32_t msd_force_shift) {
    using Vec = typename KeyVec<K>::type;
    using TI = TableIndex<GROUPS, COPIES>;
    constexpr uint32_t V = KeyVec<K>::kKeys;
    __shared__ uint32_t t0_[kBins * COPIES];
    __shared__ uint32_t t[3][GROUPS * kTableRow];
    uint32_t *t0 = t0_;
    __shared__ uint32_t s_msd[MSD ? kMsdBuckets : 1];
    uint32_t *hm = MSD ? s_msd : nullptr;
    // the hybrid form's buckets are the top 14 bits of the key RANGE: msd_hist[kMsdProbeWord] holds the shift a probe of
    // the input suggested (range_probe_kernel); a key above that range sets msd_hist[kMsdOverWord]
    uint32_t msd_shift = 0, msd_over = 0;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    if constexpr (MSD) {
        // The hybrid form buckets the keys by the top 14 bits of their RANGE (32-bit keys: bits 18-31; the reference's
        // 28-bit keys: bits 14-27; ...).  Every workgroup ORs the SAME strided sample of 4096 keys (16 KB: served by L2
        // after the first few) and derives the same bucket shift; every key above the sampled range is flagged below, so
        // a wrong guess costs the hybrid form, never the result.
        __shared__ uint32_t s_or;
        if (tid == 0) s_or = msd_force_shift ? 0xFFFFFFFFu >> (18u - min(msd_force_shift, 18u)) : 0u;  // forced: as if keys < 2^(shift + 14) had been seen
        __syncthreads();
        const uint32_t samples = msd_force_shift ? 0u : min(n, 4096u);
        const uint64_t stride = n / max(samples, 1u);  // >= 1
        uint32_t acc = 0;
        for (uint32_t i = tid; i < samples; i += THREADS) acc |= digit_word(keys[static_cast<uint64_t>(i) * stride], base_shift) - msd_base;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc |= __shfl_down(acc, o);
        if (lane == 0u && acc) atomicOr(&s_or, acc);
        __syncthreads();
        const uint32_t bits = s_or ? 32u - static_cast<uint32_t>(__clz(static_cast<int>(s_or))) : 0u;  // sampled keys < 2^bits
        msd_shift = bits > kMsdBits ? bits - kMsdBits : 0u;
        if (blockIdx.x == 0 && tid == 0) msd_hist[kMsdProbeWord] = msd_shift;  // for the plan
        // a key range below 27 bits is left to the LSD passes (the plan will say so): do not pay for the histogram
        if (msd_shift < kMsdMinShift) hm = nullptr;
        // fast count (msd_only): a range the hybrid form takes gets ONLY the bucket histogram -- 1 LDS add per key instead
        // of 5; should the plan then refuse (a bucket too large), the host counts again for the LSD passes
        else if (msd_only) t0 = nullptr;
    }
    if constexpr (MSD) {
        for (uint32_t c = tid; c < kMsdBuckets; c += THREADS) s_msd[c] = 0;
    }
    for (uint32_t c = tid; c < 3u * GROUPS * kTableRow; c += THREADS) (&t[0][0])[c] = 0;
    for (uint32_t c = tid; c < static_cast<uint32_t>(kBins * COPIES); c += THREADS) t0_[c] = 0;
    {
        const uint4 zero = make_uint4(0, 0, 0, 0);
        const uint32_t per = (status_vecs + gridDim.x - 1) / gridDim.x;
        const uint32_t z0 = blockIdx.x * per, z1 = min(z0 + per, status_vecs);
        for (uint32_t c = z0 + tid; c < z1; c += THREADS) status[c] = zero;
    }
    __syncthreads();
    const uint32_t s = blockIdx.x / slices, g = blockIdx.x % slices;
    const uint32_t part = group_len / slices;
    const uint64_t begin64 = static_cast<uint64_t>(s) * group_len + static_cast<uint64_t>(g) * part;
    if (begin64 < n) {
        const uint32_t begin = static_cast<uint32_t>(begin64);
        const uint32_t len = min(part, n - begin);
        // 16-byte loads need a 16-byte aligned address: peel `head` keys (the buffer may start anywhere in a larger
        // allocation; every slice starts a multiple of V keys after it)
        const uint32_t mis = static_cast<uint32_t>((reinterpret_cast<uintptr_t>(keys + begin) / sizeof(K)) % V);
        const uint32_t head = min((V - mis) % V, len);
        if (tid < head) digit_tables_count<TI>(t0, t[0], t[1], t[2], hm, msd_shift, msd_base, msd_over, digit_word(keys[begin + tid], base_shift));
        const Vec *v = reinterpret_cast<const Vec *>(keys + begin + head);
        const uint32_t nvec = (len - head) / V;
        constexpr uint32_t kStep = THREADS * UNROLL;
        uint32_t i0 = 0;
        Vec cur[UNROLL];
        // The loads of a lane return in issue order, and the compiler's wait before vector r is consumed must hold for
        // every way into the loop: the scheduling barriers keep the issue order r = 0, 1, ... in the prologue and in the
        // loop alike, so that wait is "all but the UNROLL - 1 youngest loads" (s_waitcnt vmcnt(UNROLL - 1) before every
        // vector) instead of one "all but one" at the top of the step.  It measures the same (the kernel runs at the
        // HBM rate of its 400 MB read plus the write-back of the previous kernel's dirty lines),
        // but the loads are what the comment above says they are.
        if (kStep <= nvec) {
#pragma unroll
            for (int r = 0; r < UNROLL; ++r) {
                cur[r] = v[r * THREADS + tid];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (; i0 + kStep <= nvec; i0 += kStep) {
            // the refill is unconditional (the last step re-reads its own vectors, which nobody consumes): a
            // conditional load would force the waits to cover the path on which it was not issued
            const uint32_t refill = i0 + 2u * kStep <= nvec ? i0 + kStep : i0;  // workgroup-uniform
            uint32_t vote = 0;
#pragma unroll
            for (int r = 0; r < UNROLL; ++r) {
                const Vec x = cur[r];
                cur[r] = v[refill + r * THREADS + tid];
                __builtin_amdgcn_sched_barrier(0);
                if (r == 0)
                    digit_tables_count_vec<K, TI, true, MSD>(t0, t[0], t[1], t[2], hm, msd_shift, msd_base, msd_over, x, base_shift, lane, vote);
                else
                    digit_tables_count_vec<K, TI, false, MSD>(t0, t[0], t[1], t[2], hm, msd_shift, msd_base, msd_over, x, base_shift, lane, vote);
            }
        }
        for (uint32_t i = i0 + tid; i < nvec; i += THREADS) {
            const Vec q = v[i];
#pragma unroll
            for (int j = 0; j < static_cast<int>(V); ++j)
                digit_tables_count<TI>(t0, t[0], t[1], t[2], hm, msd_shift, msd_base, msd_over, digit_word(KeyVec<K>::get(q, j), base_shift));
        }
        const uint32_t tail = head + nvec * V + tid;  // at most V - 1 keys
        if (tail < len) digit_tables_count<TI>(t0, t[0], t[1], t[2], hm, msd_shift, msd_base, msd_over, digit_word(keys[begin + tail], base_shift));
    }
    __syncthreads();
    if (t0 != nullptr) {
    for (uint32_t d = tid; d < static_cast<uint32_t>(kBins); d += THREADS) {
        uint32_t sum = 0;
#pragma unroll
        for (int r = 0; r < COPIES; ++r) sum += t0_[d * COPIES + ((r + d) % COPIES)];  // skewed: no bank conflicts
        if (sum)
            __hip_atomic_fetch_add(&tables[static_cast<size_t>(s) * kBins + d], sum, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
    }
    for (uint32_t c = tid; c < 3u * GROUPS * kBins; c += THREADS) {  // c = (pass - 1, group, digit)
        const uint32_t x = (&t[0][0])[(c >> 8) * kTableRow + (c & 255u)];
        if (x) __hip_atomic_fetch_add(&tables[GROUPS * kBins + c], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    }
    if constexpr (MSD) {
      if (hm != nullptr) {
        if (__ballot(msd_over != 0u) != 0ull && lane == 0u)
            __hip_atomic_fetch_or(&msd_hist[kMsdOverWord], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // top-14-bit histogram; the top-byte counts of this workgroup's pass-0 group are the sums of 64 sub-bins each
        for (uint32_t c = tid; c < kMsdBuckets; c += THREADS) {
            const uint32_t x = s_msd[c];
            if (x) __hip_atomic_fetch_add(&msd_hist[c], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < kBins) {
            uint32_t sum = 0;
            for (uint32_t j = 0; j < kMsdBuckets / kBins; ++j) sum += s_msd[tid * (kMsdBuckets / kBins) + ((j + tid) % (kMsdBuckets / kBins))];
            if (sum)
                __hip_atomic_fetch_add(&msd_slices[static_cast<size_t>(s) * kBins + tid], sum, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if constexpr (THREADS == 4 * kBins) {
        if (fp.plan != nullptr) {  // fused form: the workgroup that finishes LAST turns the tables into the plan
            __shared__ uint32_t s_last;
            // every lane's atomics above must have been performed before this workgroup's ticket is drawn
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const uint32_t ticket = __hip_atomic_fetch_add(fp.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last = ticket == gridDim.x - 1u ? 1u : 0u;
                if (s_last) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    __hip_atomic_store(fp.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch
                }
            }
            __syncthreads();
            if (s_last)
                plan_body<GROUPS>(tables, fp.plan, fp.host_head, fp.stamp, n, group_len, fp.tile, fp.tile_cap, fp.blind_cap, fp.cuts0);
        }
    }
}

// grid = kStreams * grid_tiles workgroups; block b -> XCD b % 8 -> stream b%8 + 8*((b/8) % (kStreams/8)),
// tile (b/8) / (kStreams/8): every tile's predecessors in its stream sit in lower-numbered blocks of the same XCD.
// The stream's range comes from the plan in device memory (three scalar loads): the host launches the pass before it
// has seen the plan, with room for the longest stream the plan may accept (tile_cap tiles); surplus workgroups leave.
template <typename K, int ITEMS, int WAVES, bool PAIRS, int RANK, int OCC, bool RESERVE = false>
__global__ __launch_bounds__(WAVES * 64, OCC) void onesweep_scatter_kernel(const K *__restrict__ keys_in,
                                                                       K *__restrict__ keys_out,
                                                                       const uint32_t *__restrict__ values_in,
                                                                       uint32_t *__restrict__ values_out,
                                                                       const OnesweepPlan *__restrict__ plan,
                                                                       uint32_t pass, int forced, uint32_t shift,
                                                                       uint32_t *__restrict__ status,
                                                                       unsigned long long xcc_map, int misplace,
                                                                       uint32_t spin_budget, int hold_tile, uint32_t key_base,
                                                                       MsdPlan *__restrict__ reserve) {
    constexpr uint32_t kTile = ITEMS * WAVES * 64;  // the tile the plan counted with (onesweep_tile_keys)
    __shared__ ChunkSmem<K, ITEMS, WAVES, PAIRS> sm;
    const uint32_t k = blockIdx.x >> 3, i = k / (kStreams / 8);
    // misplace (test hook): odd tiles of every stream run on the neighbouring XCD, so the look-back has to work
    // through the write-through copies instead of one L2
    const uint32_t s = ((blockIdx.x + (misplace ? (i & 1u) : 0u)) & 7u) + 8u * (k % (kStreams / 8));
    // the host enqueued this pass before it knew the plan: a pass at or after the first one that needs another form
    // (identity, unbalanced streams) leaves at once and the host enqueues it again, `forced`, in the right order
    // (the plan keeps a second copy of the streams in which such a pass has no tiles: ONE scalar load decides)
    const StreamDesc sd = forced ? plan->head.stream[pass][s] : plan->head.blind[pass][s];
    if (i >= sd.tiles) return;  // uniform per workgroup
    // forced == 2 (vrs_msd_partition_*: the first MSD pass whatever the plan thinks of THIS shard's buckets): still not without
    // counts -- a key range below 27 bits or a key outside the probed range left the bucket histogram empty or wrong, every seed
    // would be void and a reservation could run out of its range
    if (forced == 2 && plan->head.msd_counted == 0u) return;
    const uint32_t done = i * kTile;
    const uint32_t begin = sd.start + done;
    const uint32_t valid = min(kTile, sd.len - done);
    RadixDigit<K> dg;
    dg.shift = shift == kShiftFromPlan ? plan->head.msd_shift_a : shift;  // first MSD pass of the hybrid form: set by msd_plan_kernel
    dg.base = shift == kShiftFromPlan ? static_cast<K>(key_base) : static_cast<K>(0);
    // byte x of xcc_map = XCC of the blocks with blockIdx % 8 == x (probed); my stream's tiles sit in blocks = s (mod 8)
    const bool foreign = xcc_id() != static_cast<uint32_t>((xcc_map >> (8u * (s & 7u))) & 0xFFu);
    uint32_t unused = 0;
    const uint32_t *vin = PAIRS ? values_in + begin : nullptr;
    if constexpr (RESERVE) {
        // first MSD pass over bare keys: the tile reserves its place in (stream, top byte)'s range instead of looking back
        StreamReserve lb;
        const uint32_t d = threadIdx.x & 255u;
        lb.foreign = foreign;
        lb.cursor = &reserve->cursor_a[s][d];
        lb.back = &reserve->back_a[s][d];
        lb.pad_keys = d == dg(dg.template pad<K>()) ? kTile - valid : 0u;
        lb.seed = plan->group_seed[pass][sd.first_group][d];
        if (foreign) {  // the range of (stream, digit) ends where the next stream's begins
            const uint32_t next_group = s + 1u < static_cast<uint32_t>(kStreams) ? plan->head.stream[pass][s + 1u].first_group : 8u;
            lb.region_len = plan->group_seed[pass][next_group][d] - lb.seed;
        }
        if (valid == kTile)
            scatter_chunk<K, ITEMS, WAVES, PAIRS, RANK, true>(sm, keys_in + begin, vin, keys_out, values_out, valid, dg, unused, lb);
        else
            scatter_chunk<K, ITEMS, WAVES, PAIRS, RANK, false>(sm, keys_in + begin, vin, keys_out, values_out, valid, dg, unused, lb);
    } else {
        StreamLookback lb;
        lb.foreign = foreign;
        lb.hold = hold_tile >= 0 && i == static_cast<uint32_t>(hold_tile);
        lb.stream_keys = keys_in + sd.start;
        lb.done = done;
        if (lb.foreign) {
            // the earlier tiles of the stream are all full: count their digits from the keys themselves
            uint32_t *cnt = sm.whist[0];
            if (threadIdx.x < kBins) cnt[threadIdx.x] = 0;
            __syncthreads();
            recount_keys(cnt, keys_in + sd.start, done, dg);
            __syncthreads();
            if (threadIdx.x < kBins) lb.recounted = cnt[threadIdx.x];
            __syncthreads();
        }
        lb.col = status + static_cast<size_t>(s) * kBins + (threadIdx.x & 255u);
        lb.stride = static_cast<size_t>(kStreams) * kBins;
        lb.index = static_cast<int>(i);
        lb.tag = (pass + 1u) << kLbTagShift;
        lb.budget = spin_budget;
        lb.seed = threadIdx.x < kBins ? plan->group_seed[pass][sd.first_group][threadIdx.x] : 0u;
        if (valid == kTile)
            scatter_chunk<K, ITEMS, WAVES, PAIRS, RANK, true>(sm, keys_in + begin, vin, keys_out, values_out, valid, dg, unused, lb);
        else
            scatter_chunk<K, ITEMS, WAVES, PAIRS, RANK, false>(sm, keys_in + begin, vin, keys_out, values_out, valid, dg, unused, lb);
    }
    VRS_MARK_FLUSH();
}

// ---------------------------------------------------------------------------------------------
// K5b: the hybrid form's own kernels (the counting read is digit_tables_kernel<..., MSD = true>, the first MSD pass is
// onesweep_scatter_kernel on bits 24-31 with the eight input slices as streams).

// Hybrid form for 64-bit keys: the counting read.  Same workgroup -> slice mapping as digit_tables_kernel with 8 groups; ONLY
// the bucket histogram (the top 14 bits of the probed key range) and the top-byte counts of the 8 input slices are
// counted -- the LSD form of 64-bit keys makes its own tables (two counting reads) if the plan refuses.  Zeroes its share
// of the look-back status words like digit_tables_kernel.
__global__ __launch_bounds__(1024) void msd_count_u64_kernel(const uint64_t *__restrict__ keys, uint32_t n, uint32_t group_len,
                                                            uint32_t slices, uint4 *__restrict__ status, uint32_t status_vecs,
                                                            uint32_t *__restrict__ msd_hist, uint32_t *__restrict__ msd_slices) {
    constexpr uint32_t THREADS = 1024, UNROLL = 4;
    __shared__ uint32_t s_msd[kMsdBuckets];
    __shared__ unsigned long long s_or;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    if (tid == 0) s_or = 0;
    for (uint32_t c = tid; c < kMsdBuckets; c += THREADS) s_msd[c] = 0;
    {
        const uint4 zero = make_uint4(0, 0, 0, 0);
        const uint32_t per = (status_vecs + gridDim.x - 1) / gridDim.x;
        const uint32_t z0 = blockIdx.x * per, z1 = min(z0 + per, status_vecs);
        for (uint32_t c = z0 + tid; c < z1; c += THREADS) status[c] = zero;
    }
    __syncthreads();
    {   // every workgroup ORs the same strided sample of 4096 keys and derives the same bucket shift
        const uint32_t samples = min(n, 4096u);
        const uint64_t stride = n / samples;
        unsigned long long acc = 0;
        for (uint32_t i = tid; i < samples; i += THREADS) acc |= keys[static_cast<uint64_t>(i) * stride];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc |= __shfl_down(acc, o);
        if (lane == 0u && acc) atomicOr(&s_or, acc);
    }
    __syncthreads();
    const uint32_t bits = s_or ? 64u - static_cast<uint32_t>(__clzll(static_cast<long long>(s_or))) : 0u;
    const uint32_t shift = bits > kMsdBits ? bits - kMsdBits : 0u;
    if (blockIdx.x == 0 && tid == 0) msd_hist[kMsdProbeWord] = shift;
    uint32_t over = 0;
    const auto count = [&](uint64_t key) {
        const uint64_t b = key >> shift;
        over |= (b >> kMsdBits) != 0ull ? 1u : 0u;
        atomicAdd(&s_msd[static_cast<uint32_t>(b < kMsdBuckets ? b : kMsdBuckets - 1u)], 1u);
    };
    const uint32_t s = blockIdx.x / slices, g = blockIdx.x % slices;
    const uint32_t part = group_len / slices;
    const uint64_t begin64 = static_cast<uint64_t>(s) * group_len + static_cast<uint64_t>(g) * part;
    if (begin64 < n) {
        const uint32_t begin = static_cast<uint32_t>(begin64);
        const uint32_t len = min(part, n - begin);
        // 16-byte loads need a 16-byte aligned address: peel one key if the slice starts on an odd one
        const uint32_t head = min(static_cast<uint32_t>((reinterpret_cast<uintptr_t>(keys + begin) / sizeof(uint64_t)) & 1u), len);
        if (tid < head) count(keys[begin + tid]);
        const ulonglong2 *v = reinterpret_cast<const ulonglong2 *>(keys + begin + head);
        const uint32_t nvec = (len - head) / 2u;
        uint32_t i0 = 0;
        for (; i0 + THREADS * UNROLL <= nvec; i0 += THREADS * UNROLL) {
            ulonglong2 q[UNROLL];
#pragma unroll
            for (uint32_t r = 0; r < UNROLL; ++r) q[r] = v[i0 + r * THREADS + tid];
#pragma unroll
            for (uint32_t r = 0; r < UNROLL; ++r) {
                count(q[r].x);
                count(q[r].y);
            }
        }
        for (uint32_t i = i0 + tid; i < nvec; i += THREADS) {
            const ulonglong2 q = v[i];
            count(q.x);
            count(q.y);
        }
        const uint32_t tail = head + nvec * 2u + tid;  // at most one key
        if (tail < len) count(keys[begin + tail]);
    }
    __syncthreads();
    if (__ballot(over != 0u) != 0ull && lane == 0u)
        __hip_atomic_fetch_or(&msd_hist[kMsdOverWord], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (uint32_t c = tid; c < kMsdBuckets; c += THREADS) {
        const uint32_t x = s_msd[c];
        if (x) __hip_atomic_fetch_add(&msd_hist[c], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < kBins) {
        uint32_t sum = 0;
        for (uint32_t j = 0; j < kMsdBuckets / kBins; ++j) sum += s_msd[tid * (kMsdBuckets / kBins) + ((j + tid) % (kMsdBuckets / kBins))];
        if (sum)
            __hip_atomic_fetch_add(&msd_slices[static_cast<size_t>(s) * kBins + tid], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// One 1024-thread workgroup, after plan_kernel.  counts = [16384] top-14-bit histogram, then [8][256] top-byte counts per
// pass-0 group (both left zeroed for the next sort).
template <uint32_t sub_bits>  // the low bits of the bucket index the second MSD pass sorts by: 6 (a whole sort), 7 or 8
__global__ __launch_bounds__(1024) void msd_plan_kernel(uint32_t *__restrict__ counts, MsdPlan *__restrict__ msd,
                                                       OnesweepPlan *__restrict__ plan_a, OnesweepPlan *__restrict__ plan_lsd,
                                                       OnesweepPlanHead *host_head, uint32_t stamp, uint32_t n, uint32_t tile,
                                                       uint32_t tiles_b_cap, uint32_t local_cap, uint32_t *__restrict__ tables,
                                                       uint32_t group_len, uint32_t tile_cap, uint32_t blind_cap,
                                                       StreamCuts cuts0, uint32_t msd_only, uint32_t max_shift,
                                                       uint32_t *host_log) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_start[kBins + 1];  // where top byte a starts
    __shared__ uint32_t s_tiles[kBins];
    __shared__ uint32_t s_max, s_tiles_b, s_ok;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    constexpr uint32_t kPer = kMsdBuckets / 1024u;  // 16 buckets per thread
    // Everything this workgroup reads from memory is asked for at once, up front (the kernel sits between the counting read
    // and the first scatter pass: every dependent round trip here is a microsecond of the sort): the probed shift and the
    // out-of-range flag, 16 bucket counts per thread, the slices' top-byte counts.
    const uint32_t shift = counts[kMsdProbeWord], over = counts[kMsdOverWord];
    uint32_t c[kPer], sl[8];
    {
        const uint4 *cv = reinterpret_cast<const uint4 *>(counts + tid * kPer);
#pragma unroll
        for (uint32_t j = 0; j < kPer / 4u; ++j) {
            const uint4 q = cv[j];
            c[4 * j] = q.x;
            c[4 * j + 1] = q.y;
            c[4 * j + 2] = q.z;
            c[4 * j + 3] = q.w;
        }
    }
    uint32_t *slices = counts + kMsdBuckets;
#pragma unroll
    for (int g = 0; g < 8; ++g) sl[g] = tid < kBins ? slices[g * kBins + tid] : 0u;
    // Fast count (msd_only): the counting read left the LSD tables out unless the probed key range was too narrow for
    // the hybrid form anyway -- then there is no LSD plan to make (and none is needed if the hybrid form is taken).
    // (msd_only == 2: 64-bit keys -- their LSD form makes its own tables, two counting reads, if it has to run)
    const bool have_tables = msd_only == 0u || (msd_only == 1u && shift < kMsdMinShift);  // workgroup-uniform
    StreamDesc mine{};  // pass 0's stream tid (tid < kStreams)
    // first the plan of the four LSD passes (the same workgroup, no launch of its own; the head is stamped at the end)
    if (have_tables) {
        plan_body<8>(tables, plan_lsd, host_head, 0u, n, group_len, tile, tile_cap, blind_cap, cuts0);
        __syncthreads();
        if (tid < static_cast<uint32_t>(kStreams)) mine = plan_lsd->head.stream[0][tid];
    } else if (tid < static_cast<uint32_t>(kStreams)) {
        // pass 0's streams are slices of the input (the first MSD pass uses them): the same arithmetic as plan_body's
        const uint32_t k = tid;
        const uint64_t a64 = static_cast<uint64_t>(cuts0.first_group[k]) * group_len, b64 = static_cast<uint64_t>(cuts0.first_group[k + 1]) * group_len;
        const uint32_t a = static_cast<uint32_t>(a64 < n ? a64 : n), b = static_cast<uint32_t>(b64 < n ? b64 : n);
        mine = StreamDesc{a, b - a, cuts0.first_group[k], (b - a + tile - 1u) / tile};
        plan_lsd->head.stream[0][k] = mine;
        StreamDesc none = mine;
        none.tiles = 0;
        plan_lsd->head.blind[0][k] = none;  // the speculatively enqueued LSD pass 0 has no plan: it leaves at once
    }
    if (tid == 0) {
        s_max = 0;
        s_tiles_b = 0;
    }
    // (1) exclusive prefix over the 16384 buckets (the counters are left zeroed for the next sort)
    uint32_t sum = 0, mx = 0;
    {
        uint4 *cv = reinterpret_cast<uint4 *>(counts + tid * kPer);
#pragma unroll
        for (uint32_t j = 0; j < kPer / 4u; ++j) cv[j] = make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (uint32_t j = 0; j < kPer; ++j) {
        sum += c[j];
        mx = max(mx, c[j]);
    }
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if (lane >= static_cast<uint32_t>(o)) incl += t;
    }
    if (lane == 63u) s_wave[wave] = incl;
    __syncthreads();
    atomicMax(&s_max, mx);
    uint32_t run = incl - sum;
    for (uint32_t j = 0; j < wave; ++j) run += s_wave[j];
    {
        uint32_t start[kPer];
#pragma unroll
        for (uint32_t j = 0; j < kPer; ++j) {
            const uint32_t b = tid * kPer + j;
            start[j] = run;
            if ((b & ((1u << sub_bits) - 1u)) == 0u) s_start[b >> sub_bits] = run;
            run += c[j];
        }
        uint4 *vb = reinterpret_cast<uint4 *>(msd->base + tid * kPer);
#pragma unroll
        for (uint32_t j = 0; j < kPer / 4u; ++j) vb[j] = make_uint4(start[4 * j], start[4 * j + 1], start[4 * j + 2], start[4 * j + 3]);
    }
    if (tid == 1023u) {
        msd->base[kMsdBuckets] = run;  // == n
        s_start[kBins] = run;
    }
    __syncthreads();
    // with more than 6 bits for the second pass there are fewer than 256 groups for it to walk: the others are empty
    if (tid < kBins && tid >= (kMsdBuckets >> sub_bits)) s_start[tid] = s_start[kBins];
    __syncthreads();
    // (2) seeds of the first MSD pass: where top byte a of pass-0 group g goes = start of a + its keys in earlier groups
    if (tid < kBins) {
        uint32_t before = s_start[tid];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            plan_a->group_seed[0][g][tid] = before;
            before += sl[g];
            slices[g * kBins + tid] = 0;
        }
        plan_a->group_seed[0][8][tid] = before;
        // (3) tiles of top-byte bucket a in the second pass
        s_tiles[tid] = (s_start[tid + 1] - s_start[tid] + tile - 1u) / tile;
    }
    __syncthreads();
    if (tid < 8u) {  // XCD tid walks buckets tid, tid + 8, ...
        uint32_t acc = 0;
        for (uint32_t k = 0; k < 32u; ++k) {
            msd->xcd_tiles[tid][k] = acc;
            acc += s_tiles[tid + 8u * k];
        }
        msd->xcd_tiles[tid][32] = acc;
        atomicMax(&s_tiles_b, acc);
    }
    __syncthreads();
    // the probed range must hold every key and be 27 to 32 bits wide: a narrower range leaves the four LSD passes an
    // identity pass to drop (they then move 28 bytes per key too, without the local sort's LDS work: 24-bit keys measured
    // 0.80 ms LSD vs 0.89 ms hybrid at 10^8 keys), a wider one cannot occur; at most 18 low bits go to the local sort
    if (tid == 0)
        s_ok = (over == 0u && shift >= kMsdMinShift && shift <= max_shift && s_max <= local_cap && s_tiles_b <= tiles_b_cap) ? 1u : 0u;
    __syncthreads();
    // (4) the first MSD pass's streams are pass 0's (slices of the input); exactly one of the two speculatively enqueued
    //     first passes is armed
    if (tid < static_cast<uint32_t>(kStreams)) {
        StreamDesc d = mine;
        plan_a->head.stream[0][tid] = d;
        if (!s_ok) d.tiles = 0;
        plan_a->head.blind[0][tid] = d;
        if (s_ok) plan_lsd->head.blind[0][tid].tiles = 0;
    }
    if (tid == 0) {
        counts[kMsdOverWord] = 0;  // re-armed for the next sort (the shift word is rewritten by every counting read)
        msd->shift = shift;
        msd->ok = s_ok;
        msd->sub_bits = sub_bits;
        plan_a->head.first_abnormal = 4;
        plan_a->head.msd_shift_a = shift + sub_bits;  // the first MSD pass's digit: the top 8 bits of the range
        plan_a->head.msd_counted = (over == 0u && shift >= kMsdMinShift && shift <= max_shift) ? 1u : 0u;  // the bucket histogram holds every key
        plan_lsd->head.msd_ok = s_ok;
        plan_lsd->head.msd_tiles_b = s_tiles_b;
        plan_lsd->head.msd_max_bucket = s_max;
        plan_lsd->head.lsd_missing = have_tables ? 0u : 1u;
        if (host_head) {
            __hip_atomic_store(&host_head->lsd_missing, have_tables ? 0u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&host_head->msd_ok, s_ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&host_head->msd_tiles_b, s_tiles_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&host_head->msd_max_bucket, s_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            // a caller that enqueues several plans before it looks (the rounds of the multi-GPU step) finds each decision in a
            // log of the last 32, keyed by the stamp (pinned host memory): {stamp's low 31 bits, ok}
            if (host_log)
                __hip_atomic_store(&host_log[stamp & (kMsdLogWords - 1u)], (stamp << 1) | s_ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
            __hip_atomic_store(&host_head->ready, stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Second MSD pass: inside every top-byte bucket (a contiguous range of the first pass's output) a stable scatter by bits
// 18-23 -- the look-back machinery with one chain per bucket.  Block b -> XCD b % 8, which walks its 32 buckets in order;
// status row of (XCD x, its j-th tile) = j * 8 + x, so a bucket's tiles are 8 rows apart like a stream's.
template <typename K, int ITEMS, int RANK, bool PAIRS, bool RESERVE = false>
__global__ __launch_bounds__(512, 4) void msd_pass_b_kernel(const K *__restrict__ keys_in, K *__restrict__ keys_out,
                                                            const uint32_t *__restrict__ values_in, uint32_t *__restrict__ values_out,
                                                            MsdPlan *__restrict__ msd, uint32_t *__restrict__ status,
                                                            unsigned long long xcc_map, uint32_t spin_budget, uint32_t key_base,
                                                            uint32_t sub_bits) {
    constexpr uint32_t kTile = ITEMS * 8 * 64;  // the tile the plan counted with (onesweep_tile_keys)
    __shared__ ChunkSmem<K, ITEMS, 8, PAIRS> sm;
    const uint32_t x = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const uint32_t *pt = msd->xcd_tiles[x];
    if (msd->ok == 0u || j >= pt[32]) return;  // uniform per workgroup (enqueued before the plan was known: it may have said no)
    uint32_t k = 0;           // the bucket whose tiles contain j: largest k with pt[k] <= j
#pragma unroll
    for (uint32_t step = 16; step >= 1; step >>= 1)
        if (pt[k + step] <= j) k += step;
    const uint32_t a = x + 8u * k, i = j - pt[k];
    // sub_bits: 6 in a whole sort; up to 8 when the caller grouped the keys by fewer bits (a kernel argument, what the plan was
    // made with: a word of the plan would be one more dependent load in front of the bucket's bounds)
    if (((a + 1u) << sub_bits) > kMsdBuckets) return;  // (no such group: the plan gave it no tiles)
    const uint32_t first = msd->base[a << sub_bits], last = msd->base[(a + 1u) << sub_bits];
    const uint32_t done = i * kTile;
    const uint32_t begin = first + done;
    const uint32_t valid = min(kTile, last - begin);
    BitsDigit dg{msd->shift, (1u << sub_bits) - 1u, key_base};
    const bool foreign = xcc_id() != static_cast<uint32_t>((xcc_map >> (8u * x)) & 0xFFu);
    uint32_t unused = 0;
    const uint32_t *vin = PAIRS ? values_in + begin : nullptr;
    if constexpr (RESERVE) {
        // bare keys take their place in the bucket's range by reservation (StreamReserve)
        StreamReserve lb;
        const uint32_t b = (a << sub_bits) + min(threadIdx.x & 255u, (1u << sub_bits) - 1u);
        lb.foreign = foreign;
        lb.cursor = &msd->cursor_b[b];
        lb.back = &msd->back_b[b];
        lb.pad_keys = (threadIdx.x & 255u) == dg(dg.template pad<K>()) ? kTile - valid : 0u;
        lb.seed = msd->base[b];
        if (foreign) lb.region_len = msd->base[b + 1u] - lb.seed;
        if (valid == kTile)
            scatter_chunk<K, ITEMS, 8, PAIRS, RANK, true>(sm, keys_in + begin, vin, keys_out, values_out, valid, dg, unused, lb);
        else
            scatter_chunk<K, ITEMS, 8, PAIRS, RANK, false>(sm, keys_in + begin, vin, keys_out, values_out, valid, dg, unused, lb);
    } else {
        StreamLookback lb;
        lb.foreign = foreign;
        lb.stream_keys = keys_in + first;
        lb.done = done;
        if (lb.foreign) {
            uint32_t *cnt = sm.whist[0];
            if (threadIdx.x < kBins) cnt[threadIdx.x] = 0;
            __syncthreads();
            recount_keys(cnt, keys_in + first, done, dg);
            __syncthreads();
            if (threadIdx.x < kBins) lb.recounted = cnt[threadIdx.x];
            __syncthreads();
        }
        lb.col = status + (static_cast<size_t>(pt[k]) * 8u + x) * kBins + (threadIdx.x & 255u);
        lb.stride = static_cast<size_t>(8) * kBins;
        lb.index = static_cast<int>(i);
        lb.tag = 6u << kLbTagShift;
        lb.budget = spin_budget;
        lb.seed = threadIdx.x < (1u << sub_bits) ? msd->base[(a << sub_bits) + threadIdx.x] : 0u;
        if (valid == kTile)
            scatter_chunk<K, ITEMS, 8, PAIRS, RANK, true>(sm, keys_in + begin, vin, keys_out, values_out, valid, dg, unused, lb);
        else
            scatter_chunk<K, ITEMS, 8, PAIRS, RANK, false>(sm, keys_in + begin, vin, keys_out, values_out, valid, dg, unused, lb);
    }
}

// The local sort is the last kernel of a hybrid sort and LDS-bound: it has HBM time to spare, so it also clears the look-back
// status words for the NEXT sort (the counting read, which is HBM-bound, then skips its 15.6 MB of zero stores): workgroup b
// of `blocks` clears the b-th share of status[0, vecs).
struct StatusClear {
    uint4 *status;   // nullptr: nothing to clear
    uint32_t vecs;
};
// ... and it re-arms the reservation counters of the MSD passes (MsdPlan::cursor_* / back_*): workgroup b = bucket b clears the
// second pass's counters of its bucket, the first 2 * kStreams workgroups one row each of the first pass's.
// (`cursors` = &MsdPlan::cursor_a of the same plan the kernel reads through a const pointer: a pointer of its own, so that the
// plan's fields stay scalar loads)
__device__ __forceinline__ void rearm_reservation(uint32_t *__restrict__ cursors, uint32_t threads) {
    constexpr uint32_t kRowsA = 2u * kStreams;                   // cursor_a rows, then back_a rows
    uint32_t *cursor_b = cursors + kRowsA * 256u, *back_b = cursor_b + kMsdBuckets;
    const uint32_t b = blockIdx.x;
    if (threadIdx.x == 0) {
        cursor_b[b] = 0;
        back_b[b] = 0;
    }
    if (b < kRowsA)
        for (uint32_t c = threadIdx.x; c < 256u; c += threads) cursors[b * 256u + c] = 0;
}
__device__ __forceinline__ void clear_status_share(const StatusClear &sc, uint32_t threads) {
    if (sc.status == nullptr) return;
    const uint32_t per = (sc.vecs + gridDim.x - 1u) / gridDim.x;
    const uint32_t z0 = blockIdx.x * per, z1 = min(z0 + per, sc.vecs);
    for (uint32_t c = z0 + threadIdx.x; c < z1; c += threads) sc.status[c] = make_uint4(0, 0, 0, 0);
}

// One LSD pass over the keys a workgroup holds in registers (wave-striped: wave v owns ITEMS * 64 consecutive positions,
// item i of lane l is position v * ITEMS * 64 + i * 64 + l; positions >= n hold nothing and stay where they are), through
// LDS: counters fed by returning LDS atomics, a scan over the bins, re-bucketing, striped read-back.
// STABLE: one counter table per wave (lane order inside an instruction is the RANK_ATOMIC property, item order and wave order
// come from the tables' prefix) -- equal digits keep their order.  Not STABLE: ONE table for the workgroup, a quarter of the
// zeroing and scanning; equal digits come out in any order -- enough for the FIRST pass over bare keys (keys that tie in
// this digit are told apart by the later pass or are equal), never for payloads.
template <int THREADS, int ITEMS, int BITS, bool PAIRS, bool STABLE, typename K = uint32_t>
__device__ __forceinline__ void local_pass(K (&key)[ITEMS], uint32_t (&val)[PAIRS ? ITEMS : 1], K *s_keys,
                                           uint32_t *s_vals, uint32_t *s_hist, uint32_t *s_tmp, uint32_t shift, uint32_t n) {
    // thread t scans bins [t * PER, (t + 1) * PER); a workgroup of more threads than bins (1024 threads, 512 bins: the large
    // buckets of pairs and 64-bit keys) leaves its upper waves out of the scan
    constexpr int WAVES = THREADS / 64, BINS = 1 << BITS, TABLES = STABLE ? WAVES : 1, PER = BINS >= THREADS ? BINS / THREADS : 1;
    static_assert(PER * THREADS == BINS || (PER == 1 && THREADS % BINS == 0), "every scanning thread owns PER whole bins");
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const bool scans = THREADS <= BINS || tid < static_cast<uint32_t>(BINS);  // wave-uniform
    for (uint32_t c = tid; c < TABLES * BINS; c += THREADS) s_hist[c] = 0;
    __syncthreads();
    uint32_t *my = s_hist + (STABLE ? wave * BINS : 0u);
    const uint32_t seg = wave * (ITEMS * 64) + lane;
    uint32_t rank[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        rank[i] = seg + i * 64;
        if (rank[i] < n) {
            const uint32_t d = static_cast<uint32_t>(key[i] >> shift) & (BINS - 1);
            const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
            const uint64_t active = __ballot(1);
            if (__ballot(d == d0) == active) {  // one digit value for the whole instruction: one add instead of up to 64 on one counter
                uint32_t old = 0;
                if (__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(active >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(active), 0u)) == 0u)
                    old = __hip_atomic_fetch_add(&my[d0], static_cast<uint32_t>(__popcll(active)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                rank[i] = __builtin_amdgcn_readfirstlane(old) +
                          __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(active >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(active), 0u));
            } else {
                rank[i] = __hip_atomic_fetch_add(&my[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();
    {   // exclusive prefix over (bin, table): thread t owns bins [t * PER, (t + 1) * PER)
        uint32_t c[TABLES][PER], total = 0;
#pragma unroll
        for (int v = 0; v < TABLES; ++v) {
            if constexpr (PER == 2) {
                const uint2 q = reinterpret_cast<const uint2 *>(s_hist + v * BINS)[tid];
                c[v][0] = q.x;
                c[v][1] = q.y;
                total += q.x + q.y;
            } else {
#pragma unroll
                for (int p_ = 0; p_ < PER; ++p_) {
                    c[v][p_] = scans ? s_hist[v * BINS + tid * PER + p_] : 0u;
                    total += c[v][p_];
                }
            }
        }
        uint32_t incl = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o);
            if (lane >= static_cast<uint32_t>(o)) incl += t;
        }
        if (lane == 63u) s_tmp[wave] = incl;
        __syncthreads();
        uint32_t acc = incl - total;
#pragma unroll
        for (int v = 0; v < WAVES; ++v) acc += (static_cast<uint32_t>(v) < wave) ? s_tmp[v] : 0u;
        uint32_t out[TABLES][PER];
#pragma unroll
        for (int p_ = 0; p_ < PER; ++p_) {
#pragma unroll
            for (int v = 0; v < TABLES; ++v) {
                out[v][p_] = acc;
                acc += c[v][p_];
            }
        }
#pragma unroll
        for (int v = 0; v < TABLES; ++v) {
            if constexpr (PER == 2) {
                reinterpret_cast<uint2 *>(s_hist + v * BINS)[tid] = make_uint2(out[v][0], out[v][1]);
            } else {
#pragma unroll
                for (int p_ = 0; p_ < PER; ++p_)
                    if (scans) s_hist[v * BINS + tid * PER + p_] = out[v][p_];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
        if (seg + i * 64 < n) rank[i] += my[static_cast<uint32_t>(key[i] >> shift) & (BINS - 1)];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
        if (seg + i * 64 < n) s_keys[rank[i]] = key[i];
    if constexpr (PAIRS) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
            if (seg + i * 64 < n) s_vals[rank[i]] = val[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) key[i] = s_keys[seg + i * 64];
    if constexpr (PAIRS) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) val[i] = s_vals[seg + i * 64];
    }
    __syncthreads();
}

// the bucket with ITEMS keys per thread (n <= ITEMS * THREADS): read once, two stable 9-bit passes, written back
template <int THREADS, int ITEMS, bool PAIRS>
__device__ __forceinline__ void local_sort_bucket(uint32_t *bucket, uint32_t *bucket_vals, uint32_t n, uint32_t *s_keys,
                                                  uint32_t *s_vals, uint32_t *s_hist, uint32_t *s_tmp) {
    constexpr int BITS = 9;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t key[ITEMS], val[PAIRS ? ITEMS : 1];
    const uint32_t seg = wave * (ITEMS * 64) + lane;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t idx = seg + i * 64;
        const uint32_t k = bucket[idx < n ? idx : n - 1u];
        key[i] = k;  // positions >= n hold nothing: the passes leave them alone and they are not written
    }
    if constexpr (PAIRS) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t idx = seg + i * 64;
            val[i] = bucket_vals[idx < n ? idx : n - 1u];
        }
    }
    local_pass<THREADS, ITEMS, BITS, PAIRS, PAIRS>(key, val, s_keys, s_vals, s_hist, s_tmp, 0, n);  // bare keys: any order of ties
    local_pass<THREADS, ITEMS, BITS, PAIRS, true>(key, val, s_keys, s_vals, s_hist, s_tmp, BITS, n);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t idx = seg + i * 64;
        if (idx < n) bucket[idx] = key[i];
    }
    if constexpr (PAIRS) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t idx = seg + i * 64;
            if (idx < n) bucket_vals[idx] = val[i];
        }
    }
}

// ---- the local sort of bare uint32 keys (round 3 form).  One workgroup per bucket of the MSD partition, the bucket sorted by its
// low 18 bits inside LDS in two 9-bit passes and written back in place -- the algorithm of local_pass above (returning LDS
// atomics rank the keys; pass 1 over bare keys in any order of ties with ONE counter table, pass 2 stable with one table per
// wave), laid out for the LDS pipe, which is what bounds this kernel (rocprofv3: LDS array busy 80 % of the kernel, 60 % of
// that bank conflicts of the three random accesses per key and pass; profiles/labs/r03_local_sort_variants.txt):
//  * the bucket is moved in 16-byte vectors: global_load_dwordx4 from the bucket's first 16-byte boundary (slot q = key index
//    minus that boundary; the first `mis` slots belong to the bucket before), ds_read_b128, global_store_dwordx4;
//  * pass 1 writes position L (the order pass 2 must see) to LDS word (L & ~255) | ((L & 63) << 2) | ((L >> 6) & 3), so that ONE
//    ds_read_b128 per lane returns the lane's four wave-striped items of pass 2 (item 4g + c of lane t is L = seg + (4g + c) 64 + t);
//    pass 2 writes slot mis + position, so the final read is a ds_read_b128 of whole 16-byte global vectors;
//  * no item is predicated: a slot that holds no key (before the bucket's first key, behind its last) takes part with a dummy
//    counter of its own -- one per LANE: 64 lanes returning from ONE counter are served one after the other, 115 instead of 10
//    cycles per instruction -- and a position fixed by arithmetic (it keeps its place behind the keys); the selects are compiled
//    into the first and last vector row of pass 1 and the last 17 items of a wave in pass 2 only;
//  * counters count BYTES (add 4): every rank is an LDS byte offset, pass 2's prefix starts at 4 mis;
//  * the same-counter guard of local_pass (a whole instruction on one counter: constant digits) costs 16 us at 10^8 uniform
//    keys when compiled into every item, so it is switched per bucket and pass: every wave looks at its first vector row, and
//    only a bucket in which some instruction has half its lanes on one counter runs the guarded form.
// the value unchanged, but opaque to the optimiser: used to make it RECOMPUTE a counter address (two VALU instructions)
// instead of keeping 28 of them alive from the returning adds to the base reads -- the registers that decide between 119 and
// "128 + spills to scratch" (scratch traffic is HBM traffic: 230 MB per launch at 10^8 keys, profiles/labs/r03_local_sort_spills.txt)
// (opaque(): vrs_device.hpp)
// (kLeanRow, kLeanMaxVec, lean_sort_body: vrs_local_sort.hpp)
template <int THREADS, int VEC>
__device__ __attribute__((noinline)) void lean_sort_guarded(uint32_t *abase, uint32_t mis, uint32_t n, uint32_t *s_keys, uint32_t *s_hist2,
                                                           uint32_t *s_tmp, uint32_t guards);

template <int THREADS, int VEC>
__device__ __forceinline__ void lean_sort_bucket(uint32_t *abase, uint32_t mis, uint32_t n, uint32_t *s_keys, uint32_t *s_hist2,
                                                 uint32_t *s_tmp) {
    constexpr int WAVES = THREADS / 64, ITEMS = 4 * VEC;
    static_assert(THREADS == 256 || THREADS == 512, "the scans give every thread 2 or 1 bins");
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t end = mis + n;  // slots [mis, end) hold keys
    const uint32_t nvec = (end + 3u) / 4u;
    uint32_t k[ITEMS];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        uint32_t v = j * THREADS + tid;
        if (j == VEC - 1) v = v < nvec ? v : nvec - 1u;  // only the last row can reach behind the bucket
        const uint4 t = reinterpret_cast<const uint4 *>(abase)[v];
        k[4 * j] = t.x;
        k[4 * j + 1] = t.y;
        k[4 * j + 2] = t.z;
        k[4 * j + 3] = t.w;
    }
    // every table zeroed here: WAVES tables of pass 2, then pass 1's
    uint32_t *s_hist = s_hist2 + WAVES * kLeanRow;
    {
        constexpr uint32_t kVecs = (WAVES + 1) * kLeanRow / 4;
        for (uint32_t c = tid; c < kVecs; c += THREADS) reinterpret_cast<uint4 *>(s_hist2)[c] = make_uint4(0, 0, 0, 0);
    }
    {   // does some instruction of this wave's first row put half its lanes on one counter?  bit 0: pass 1, bit 1: pass 2
        uint32_t skew = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t a1 = k[c] & 511u, a2 = (k[c] >> 9) & 511u;
            skew |= __popcll(__ballot(a1 == __builtin_amdgcn_readfirstlane(a1))) >= 32 ? 1u : 0u;
            skew |= __popcll(__ballot(a2 == __builtin_amdgcn_readfirstlane(a2))) >= 32 ? 2u : 0u;
        }
        if (lane == 0u) s_tmp[16 + wave] = skew;
    }
    __syncthreads();
    uint32_t guards = 0;
#pragma unroll
    for (int v = 0; v < WAVES; ++v) guards |= s_tmp[16 + v];
    guards = __builtin_amdgcn_readfirstlane(guards);
#ifdef VRS_LAB_NO_LOCAL_GUARD  // lab builds: the guarded copy compiled away
    guards = 0;
#endif
    // Two copies of the rest.  The common one is inlined and carries no trace of the guard; the guarded one is a CALL that loads
    // the bucket again -- kept out of line so that its register demand cannot push the common path into scratch spills (spills
    // are HBM traffic: with both inlined the kernel moved 1032 instead of 800 MB per launch at 10^8 keys).
    if (guards == 0u) lean_sort_body<THREADS, VEC, false>(k, abase, mis, n, s_keys, s_hist2, s_tmp, false, false);
    else lean_sort_guarded<THREADS, VEC>(abase, mis, n, s_keys, s_hist2, s_tmp, guards);
}

template <int THREADS, int VEC>
__device__ __attribute__((noinline)) void lean_sort_guarded(uint32_t *abase, uint32_t mis, uint32_t n, uint32_t *s_keys, uint32_t *s_hist2,
                                                           uint32_t *s_tmp, uint32_t guards) {
    constexpr int ITEMS = 4 * VEC;
    const uint32_t nvec = (mis + n + 3u) / 4u;
    uint32_t k[ITEMS];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        uint32_t v = j * THREADS + threadIdx.x;
        if (j == VEC - 1) v = v < nvec ? v : nvec - 1u;
        const uint4 t = reinterpret_cast<const uint4 *>(abase)[v];
        k[4 * j] = t.x;
        k[4 * j + 1] = t.y;
        k[4 * j + 2] = t.z;
        k[4 * j + 3] = t.w;
    }
    lean_sort_body<THREADS, VEC, true>(k, abase, mis, n, s_keys, s_hist2, s_tmp, (guards & 1u) != 0u, (guards & 2u) != 0u);
}

// THREADS = 256: up to 7165 keys per bucket (uniform keys: N <= 1.05e8), 38 KB of LDS, four workgroups per CU;
// THREADS = 512: up to 14333 keys (N <= 2.1e8), 78 KB, two per CU -- the same 16 waves
template <int THREADS>
__global__ __launch_bounds__(THREADS, 4) void msd_local_sort_keys_kernel(uint32_t *__restrict__ keys, const MsdPlan *__restrict__ msd, StatusClear sc,
                                                                         uint32_t *__restrict__ cursors) {
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[THREADS * 4 * kLeanMaxVec + 4];
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[(THREADS / 64 + 1) * kLeanRow];
    __shared__ uint32_t s_tmp[32];
    if (msd->ok == 0u) return;  // enqueued before the plan was known, and the plan refused the hybrid form
    rearm_reservation(cursors, THREADS);
    clear_status_share(sc, THREADS);
    const uint32_t begin = msd->base[blockIdx.x], n = msd->base[blockIdx.x + 1] - begin;
    const uint32_t mis = static_cast<uint32_t>((reinterpret_cast<uintptr_t>(keys + begin) >> 2) & 3u);
    if (n == 0 || mis + n > THREADS * 4u * kLeanMaxVec) return;  // uniform; above the capacity cannot happen (the plan would have refused)
    uint32_t *abase = keys + begin - mis;
    switch ((mis + n + 4u * THREADS - 1u) / (4u * THREADS)) {  // rows of THREADS vectors the bucket touches
        case 1: lean_sort_bucket<THREADS, 1>(abase, mis, n, s_keys, s_hist, s_tmp); break;
        case 2: lean_sort_bucket<THREADS, 2>(abase, mis, n, s_keys, s_hist, s_tmp); break;
        case 3: lean_sort_bucket<THREADS, 3>(abase, mis, n, s_keys, s_hist, s_tmp); break;
        case 4: lean_sort_bucket<THREADS, 4>(abase, mis, n, s_keys, s_hist, s_tmp); break;
        case 5: lean_sort_bucket<THREADS, 5>(abase, mis, n, s_keys, s_hist, s_tmp); break;
        case 6: lean_sort_bucket<THREADS, 6>(abase, mis, n, s_keys, s_hist, s_tmp); break;
        default: lean_sort_bucket<THREADS, 7>(abase, mis, n, s_keys, s_hist, s_tmp); break;
    }
}
constexpr uint32_t kLeanCap = 256u * 4u * kLeanMaxVec - 3u, kLeanBigCap = 512u * 4u * kLeanMaxVec - 3u;  // whatever the misalignment

// ---- small buckets (up to 1789 keys: uniform keys below about 2.5e7): ONE WAVE per bucket, no workgroup barrier anywhere --
// the LDS executes one wave's operations in order -- so a CU runs 16 independent buckets instead of 4 workgroups that each wait
// on barriers with their lanes mostly empty.  The same two 9-bit passes and the same slot / dummy-counter scheme as
// lean_sort_bucket, one 512-counter table reused by both passes (a single wave ranks in instruction, then lane order: stable).
// What made 10^7 keys worth the hybrid form: 16384 buckets of 610 keys take 16 us here, 60 us with 256 threads per bucket.
__device__ __forceinline__ void wave_phase() {  // orders this wave's LDS traffic for the compiler; the hardware keeps it in order anyway
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// exclusive prefix of the 512 byte-counters of a wave's table, 8 per lane, starting at `start`
__device__ __forceinline__ void wave_scan512(uint32_t *tbl, uint32_t lane, uint32_t start) {
    uint4 a = reinterpret_cast<uint4 *>(tbl)[2 * lane], b = reinterpret_cast<uint4 *>(tbl)[2 * lane + 1];
    const uint32_t s = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
    uint32_t incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if (lane >= static_cast<uint32_t>(o)) incl += t;
    }
    uint32_t acc = incl - s + start;
    uint4 oa, ob;
    oa.x = acc; acc += a.x; oa.y = acc; acc += a.y; oa.z = acc; acc += a.z; oa.w = acc; acc += a.w;
    ob.x = acc; acc += b.x; ob.y = acc; acc += b.y; ob.z = acc; acc += b.z; ob.w = acc;
    reinterpret_cast<uint4 *>(tbl)[2 * lane] = oa;
    reinterpret_cast<uint4 *>(tbl)[2 * lane + 1] = ob;
}
template <int VEC, bool GUARD>
__device__ __forceinline__ void wave_sort_body(uint32_t (&k)[4 * VEC], uint32_t *abase, uint32_t mis, uint32_t n, uint32_t *s_keys,
                                               uint32_t *tbl, bool guard1, bool guard2);
template <int VEC>
__device__ __attribute__((noinline)) void wave_sort_guarded(uint32_t *abase, uint32_t mis, uint32_t n, uint32_t *s_keys, uint32_t *tbl,
                                                           uint32_t skew);

template <int VEC>
__device__ __forceinline__ void wave_sort_bucket(uint32_t *abase, uint32_t mis, uint32_t n, uint32_t *s_keys, uint32_t *tbl) {
    constexpr int ITEMS = 4 * VEC;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t end = mis + n, nvec = (end + 3u) / 4u;
    uint32_t k[ITEMS];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        uint32_t v = j * 64 + lane;
        if (j == VEC - 1) v = v < nvec ? v : nvec - 1u;
        const uint4 t = reinterpret_cast<const uint4 *>(abase)[v];
        k[4 * j] = t.x;
        k[4 * j + 1] = t.y;
        k[4 * j + 2] = t.z;
        k[4 * j + 3] = t.w;
    }
    char *tb = reinterpret_cast<char *>(tbl);
    const auto zero_table = [&] {  // 576 words: two 16-byte stores per lane + one more from the first 16 lanes
        reinterpret_cast<uint4 *>(tbl)[2 * lane] = make_uint4(0, 0, 0, 0);
        reinterpret_cast<uint4 *>(tbl)[2 * lane + 1] = make_uint4(0, 0, 0, 0);
        if (lane < 16u) reinterpret_cast<uint4 *>(tbl)[128 + lane] = make_uint4(0, 0, 0, 0);
    };
    zero_table();
    uint32_t skew = 0;  // does an instruction of the first row put half its lanes on one counter?  bit 0: pass 1, bit 1: pass 2
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t a1 = k[c] & 511u, a2 = (k[c] >> 9) & 511u;
        skew |= __popcll(__ballot(a1 == __builtin_amdgcn_readfirstlane(a1))) >= 32 ? 1u : 0u;
        skew |= __popcll(__ballot(a2 == __builtin_amdgcn_readfirstlane(a2))) >= 32 ? 2u : 0u;
    }
    skew = __builtin_amdgcn_readfirstlane(skew);
    wave_phase();
    if (skew == 0u) wave_sort_body<VEC, false>(k, abase, mis, n, s_keys, tbl, false, false);
    else wave_sort_guarded<VEC>(abase, mis, n, s_keys, tbl, skew);  // out of line, loads the bucket again: see lean_sort_bucket
}

template <int VEC>
__device__ __attribute__((noinline)) void wave_sort_guarded(uint32_t *abase, uint32_t mis, uint32_t n, uint32_t *s_keys, uint32_t *tbl,
                                                           uint32_t skew) {
    constexpr int ITEMS = 4 * VEC;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nvec = (mis + n + 3u) / 4u;
    uint32_t k[ITEMS];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        uint32_t v = j * 64 + lane;
        if (j == VEC - 1) v = v < nvec ? v : nvec - 1u;
        const uint4 t = reinterpret_cast<const uint4 *>(abase)[v];
        k[4 * j] = t.x;
        k[4 * j + 1] = t.y;
        k[4 * j + 2] = t.z;
        k[4 * j + 3] = t.w;
    }
    wave_sort_body<VEC, true>(k, abase, mis, n, s_keys, tbl, (skew & 1u) != 0u, (skew & 2u) != 0u);
}

template <int VEC, bool GUARD>
__device__ __forceinline__ void wave_sort_body(uint32_t (&k)[4 * VEC], uint32_t *abase, uint32_t mis, uint32_t n, uint32_t *s_keys,
                                               uint32_t *tbl, bool guard1, bool guard2) {
    constexpr int ITEMS = 4 * VEC;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t end = mis + n, nvec = (end + 3u) / 4u;
    uint32_t rank[ITEMS];
    char *tb = reinterpret_cast<char *>(tbl);
    const auto zero_table = [&] {
        reinterpret_cast<uint4 *>(tbl)[2 * lane] = make_uint4(0, 0, 0, 0);
        reinterpret_cast<uint4 *>(tbl)[2 * lane + 1] = make_uint4(0, 0, 0, 0);
        if (lane < 16u) reinterpret_cast<uint4 *>(tbl)[128 + lane] = make_uint4(0, 0, 0, 0);
    };
    const auto ranked_add = [&](uint32_t a, bool guard) -> uint32_t {
        uint32_t *counter = reinterpret_cast<uint32_t *>(tb + a);
        if (GUARD && guard) {
            const uint32_t a0 = __builtin_amdgcn_readfirstlane(a);
            if (__ballot(a == a0) == ~0ull) {
                uint32_t old = 0;
                if (lane == 0u) old = __hip_atomic_fetch_add(counter, 256u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return __builtin_amdgcn_readfirstlane(old) + 4u * lane;
            }
        }
        return __hip_atomic_fetch_add(counter, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // ---- pass 1: low 9 bits (any order of ties)
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        uint32_t a = (k[i] << 2) & 0x7FCu;
        if (i < 4 || i >= ITEMS - 4) {
            const uint32_t q = 4u * ((i >> 2) * 64 + lane) + (i & 3);
            a = (q - mis < n) ? a : 2048u + 4u * lane;
        }
        rank[i] = ranked_add(a, guard1);
    }
    wave_phase();
    wave_scan512(tbl, lane, 0u);
    wave_phase();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        uint32_t a = (opaque(k[i]) << 2) & 0x7FCu;
        if (i < 4 || i >= ITEMS - 4) {
            const uint32_t q = 4u * ((i >> 2) * 64 + lane) + (i & 3);
            const bool valid = q - mis < n;
            a = valid ? a : 2048u + 4u * lane;
            const uint32_t r = rank[i] + *reinterpret_cast<const uint32_t *>(tb + a);
            rank[i] = valid ? r : 4u * (q < mis ? n + q : q);
            continue;
        }
        rank[i] += *reinterpret_cast<const uint32_t *>(tb + a);
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t Lb = rank[i];
        const uint32_t ph = (Lb & ~1023u) | ((Lb & 252u) << 2) | ((Lb >> 6) & 12u);
        *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(s_keys) + ph) = k[i];
    }
    wave_phase();
    zero_table();  // behind pass 1's base reads in the LDS queue
#pragma unroll
    for (int g = 0; g < VEC; ++g) {
        const uint4 t = reinterpret_cast<const uint4 *>(s_keys + g * 256)[lane];
        k[4 * g] = t.x;
        k[4 * g + 1] = t.y;
        k[4 * g + 2] = t.z;
        k[4 * g + 3] = t.w;
    }
    wave_phase();
    // ---- pass 2: high 9 bits, stable (instruction order, then lane order); any slot may be empty here: a bucket of a few rows
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        uint32_t a = (k[i] >> 7) & 0x7FCu;
        a = (i * 64 + lane < n) ? a : 2048u + 4u * lane;
        rank[i] = ranked_add(a, guard2);
    }
    wave_phase();
    wave_scan512(tbl, lane, 4u * mis);
    wave_phase();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t L = i * 64 + lane;
        uint32_t a = (opaque(k[i]) >> 7) & 0x7FCu;
        a = L < n ? a : 2048u + 4u * lane;
        const uint32_t r = rank[i] + *reinterpret_cast<const uint32_t *>(tb + a);
        rank[i] = L < n ? r : 4u * (mis + L);
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(s_keys) + rank[i]) = k[i];
    wave_phase();
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const uint32_t v = j * 64 + lane;
        if (v < nvec) {
            const uint4 q4 = reinterpret_cast<const uint4 *>(s_keys)[v];
            const uint32_t q = 4u * v;
            if (q >= mis && q + 4u <= end) {
                reinterpret_cast<uint4 *>(abase)[v] = q4;
            } else {
                if (q + 0u - mis < n) abase[q + 0u] = q4.x;
                if (q + 1u - mis < n) abase[q + 1u] = q4.y;
                if (q + 2u - mis < n) abase[q + 2u] = q4.z;
                if (q + 3u - mis < n) abase[q + 3u] = q4.w;
            }
        }
    }
}
constexpr uint32_t kWaveCap = 64u * 4u * kLeanMaxVec - 3u;  // 1789 keys
__global__ __launch_bounds__(64, 4) void msd_local_sort_wave_kernel(uint32_t *__restrict__ keys, const MsdPlan *__restrict__ msd, StatusClear sc,
                                                                  uint32_t *__restrict__ cursors) {
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[64 * 4 * kLeanMaxVec + 4];
    __shared__ __attribute__((aligned(16))) uint32_t s_tbl[kLeanRow];
    if (msd->ok == 0u) return;
    rearm_reservation(cursors, 64);
    clear_status_share(sc, 64);
    const uint32_t begin = msd->base[blockIdx.x], n = msd->base[blockIdx.x + 1] - begin;
    const uint32_t mis = static_cast<uint32_t>((reinterpret_cast<uintptr_t>(keys + begin) >> 2) & 3u);
    if (n == 0 || mis + n > 64u * 4u * kLeanMaxVec) return;
    uint32_t *abase = keys + begin - mis;
    switch ((mis + n + 255u) / 256u) {
        case 1: wave_sort_bucket<1>(abase, mis, n, s_keys, s_tbl); break;
        case 2: wave_sort_bucket<2>(abase, mis, n, s_keys, s_tbl); break;
        case 3: wave_sort_bucket<3>(abase, mis, n, s_keys, s_tbl); break;
        case 4: wave_sort_bucket<4>(abase, mis, n, s_keys, s_tbl); break;
        case 5: wave_sort_bucket<5>(abase, mis, n, s_keys, s_tbl); break;
        case 6: wave_sort_bucket<6>(abase, mis, n, s_keys, s_tbl); break;
        default: wave_sort_bucket<7>(abase, mis, n, s_keys, s_tbl); break;
    }
}

// Key + payload pairs: the payload doubles a bucket's LDS footprint (53 + 16 KB), so two workgroups of 512 threads x up
// to 13 pairs share a CU.  Buckets of up to twice that (inputs of 10^8 to 2 * 10^8 pairs) get ONE workgroup of 1024 threads
// per CU (106 + 32 KB).
constexpr int kLocalPairThreads = 512, kLocalPairItems = kLocalCap / kLocalPairThreads;  // 13
constexpr int kLocalPairThreadsBig = 1024;
constexpr uint32_t kLocalCapBig = kLocalPairThreadsBig * kLocalPairItems;  // 13312
template <int THREADS>
__global__ __launch_bounds__(THREADS, 4) void msd_local_sort_pairs_kernel(uint32_t *__restrict__ keys,
                                                                                         uint32_t *__restrict__ values,
                                                                                         const MsdPlan *__restrict__ msd, StatusClear sc,
                                                                                         uint32_t *__restrict__ cursors) {
    constexpr int WAVES = THREADS / 64;
    constexpr uint32_t CAP = THREADS * kLocalPairItems;
    __shared__ uint32_t s_keys[CAP];
    __shared__ uint32_t s_vals[CAP];
    __shared__ uint32_t s_hist[WAVES << 9];
    __shared__ uint32_t s_tmp[1 + WAVES];
    if (msd->ok == 0u) return;
    rearm_reservation(cursors, THREADS);
    clear_status_share(sc, THREADS);
    const uint32_t begin = msd->base[blockIdx.x], n = msd->base[blockIdx.x + 1] - begin;
    if (n == 0 || n > CAP) return;
    uint32_t *bucket = keys + begin, *bvals = values + begin;
    const uint32_t used = (n + THREADS - 1u) / THREADS;
    if (used <= 2) local_sort_bucket<THREADS, 2, true>(bucket, bvals, n, s_keys, s_vals, s_hist, s_tmp);
    else if (used <= 4) local_sort_bucket<THREADS, 4, true>(bucket, bvals, n, s_keys, s_vals, s_hist, s_tmp);
    else if (used <= 6) local_sort_bucket<THREADS, 6, true>(bucket, bvals, n, s_keys, s_vals, s_hist, s_tmp);
    else if (used <= 8) local_sort_bucket<THREADS, 8, true>(bucket, bvals, n, s_keys, s_vals, s_hist, s_tmp);
    else if (used <= 10) local_sort_bucket<THREADS, 10, true>(bucket, bvals, n, s_keys, s_vals, s_hist, s_tmp);
    else if (used <= 12) local_sort_bucket<THREADS, 12, true>(bucket, bvals, n, s_keys, s_vals, s_hist, s_tmp);
    else local_sort_bucket<THREADS, kLocalPairItems, true>(bucket, bvals, n, s_keys, s_vals, s_hist, s_tmp);
}

// 64-bit keys: the bucket's keys differ only in their low `shift` bits (up to 50): ceil(shift / 9) LDS passes, the first in any
// order of ties, the others stable -- or, when that is more than four, the top four and a check (see below).  512 threads x up to 13 keys (8 bytes each: the footprint of the pairs kernel).
template <int THREADS, int ITEMS>
__device__ __forceinline__ void local_sort_bucket_u64(uint64_t *bucket, uint32_t n, uint32_t passes, uint64_t *s_keys,
                                                      uint32_t *s_hist, uint32_t *s_tmp) {
    constexpr int BITS = 9;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint64_t key[ITEMS];
    uint32_t none[1];
    const uint32_t seg = wave * (ITEMS * 64) + lane;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t idx = seg + i * 64;
        key[i] = bucket[idx < n ? idx : n - 1u];
    }
    if (passes > 4u) {
        // More than 36 low bits: sort by the TOP four digits first -- with a few thousand keys per bucket hardly any two tie in
        // 36 bits (three digits are not enough: 23 bits below the bucket's own, two ties per bucket of 6000 uniform keys) -- and
        // look whether that already is the order of the whole keys (neighbours compared in LDS).  Only a bucket with a pair
        // still out of order runs all the passes, from the bottom.
        local_pass<THREADS, ITEMS, BITS, false, false, uint64_t>(key, none, s_keys, nullptr, s_hist, s_tmp, BITS * (passes - 4u), n);
        for (uint32_t p_ = passes - 3u; p_ < passes; ++p_)
            local_pass<THREADS, ITEMS, BITS, false, true, uint64_t>(key, none, s_keys, nullptr, s_hist, s_tmp, BITS * p_, n);
        int bad = 0;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t idx = seg + i * 64;
            if (idx + 1u < n) bad |= key[i] > s_keys[idx + 1u] ? 1 : 0;  // s_keys still holds what the last pass left
        }
        if (__syncthreads_or(bad) == 0) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const uint32_t idx = seg + i * 64;
                if (idx < n) bucket[idx] = key[i];
            }
            return;
        }
    }
    if (passes > 0u) local_pass<THREADS, ITEMS, BITS, false, false, uint64_t>(key, none, s_keys, nullptr, s_hist, s_tmp, 0, n);
    for (uint32_t p_ = 1; p_ < passes; ++p_)
        local_pass<THREADS, ITEMS, BITS, false, true, uint64_t>(key, none, s_keys, nullptr, s_hist, s_tmp, BITS * p_, n);
    if (passes == 0u) return;  // one distinct key per bucket
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t idx = seg + i * 64;
        if (idx < n) bucket[idx] = key[i];
    }
}

template <int THREADS>
__global__ __launch_bounds__(THREADS, 4) void msd_local_sort_u64_kernel(uint64_t *__restrict__ keys,
                                                                        const MsdPlan *__restrict__ msd, StatusClear sc,
                                                                        uint32_t *__restrict__ cursors) {
    constexpr int WAVES = THREADS / 64;
    constexpr uint32_t CAP = THREADS * kLocalPairItems;
    __shared__ uint64_t s_keys[CAP];
    __shared__ uint32_t s_hist[WAVES << 9];
    __shared__ uint32_t s_tmp[1 + WAVES];
    if (msd->ok == 0u) return;
    rearm_reservation(cursors, THREADS);
    clear_status_share(sc, THREADS);
    const uint32_t begin = msd->base[blockIdx.x], n = msd->base[blockIdx.x + 1] - begin;
    if (n == 0 || n > CAP) return;
    const uint32_t passes = (msd->shift + 8u) / 9u;
    uint64_t *bucket = keys + begin;
    const uint32_t used = (n + THREADS - 1u) / THREADS;
    if (used <= 2) local_sort_bucket_u64<THREADS, 2>(bucket, n, passes, s_keys, s_hist, s_tmp);
    else if (used <= 4) local_sort_bucket_u64<THREADS, 4>(bucket, n, passes, s_keys, s_hist, s_tmp);
    else if (used <= 6) local_sort_bucket_u64<THREADS, 6>(bucket, n, passes, s_keys, s_hist, s_tmp);
    else if (used <= 8) local_sort_bucket_u64<THREADS, 8>(bucket, n, passes, s_keys, s_hist, s_tmp);
    else if (used <= 10) local_sort_bucket_u64<THREADS, 10>(bucket, n, passes, s_keys, s_hist, s_tmp);
    else if (used <= 12) local_sort_bucket_u64<THREADS, 12>(bucket, n, passes, s_keys, s_hist, s_tmp);
    else local_sort_bucket_u64<THREADS, kLocalPairItems>(bucket, n, passes, s_keys, s_hist, s_tmp);
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

uint32_t onesweep_tile_keys(int key_bytes) { return key_bytes == 8 ? 4096u * VRS_LB_WAVES / 8u : VRS_LB_ITEMS * 64u * VRS_LB_WAVES; }

// largest power of two <= x (x >= 1)
static uint32_t floor_pow2(uint32_t x) {
    uint32_t p = 1;
    while (p * 2u <= x) p *= 2u;
    return p;
}

template <typename K, int GROUPS, int THREADS, int COPIES, int UNROLL, int OCC, bool MSD = false>
static void launch_digit_tables_variant(hipStream_t stream, const void *keys, uint32_t n, uint32_t base_shift,
                                        uint32_t group_len, uint32_t *tables, uint32_t *status, size_t status_words,
                                        int compute_units, LaunchEvents ev, const FusedPlanArgs &fp,
                                        uint32_t *msd_counts = nullptr, uint32_t msd_only = 0, uint32_t msd_base = 0,
                                        uint32_t msd_force_shift = 0) {
    // one workgroup per (pass-0 group, slice): a power-of-two number of slices that fills the chip once
    const uint32_t wgs = static_cast<uint32_t>(compute_units) * (OCC * 256 / THREADS);
    uint32_t slices = floor_pow2(wgs / GROUPS > 0 ? wgs / GROUPS : 1u);
    if (MSD) {
        // every workgroup of the hybrid form's counting read zeroes and flushes 16384 counters whatever it counts: below about
        // 3e7 keys fewer, longer-running workgroups are cheaper (10^7 keys: 64 workgroups, 1 M instead of 4 M counter flushes)
        const uint32_t by_size = floor_pow2(std::max<uint32_t>(n / (GROUPS * 131072u), 1u));
        slices = std::min(slices, by_size);
    }
    const dim3 grid(GROUPS * slices), block(THREADS);
    const uint32_t vecs = static_cast<uint32_t>(status_words / 4);
    VRS_LAUNCH((digit_tables_kernel<K, GROUPS, THREADS, COPIES, UNROLL, OCC, MSD>), grid, block, stream, ev,
               static_cast<const K *>(keys), n, base_shift, group_len, slices, tables, reinterpret_cast<uint4 *>(status), vecs,
               fp, msd_counts, msd_counts ? msd_counts + kMsdBuckets : nullptr, msd_only, msd_base, msd_force_shift);
}

hipError_t launch_digit_tables_msd(hipStream_t stream, const void *keys, uint32_t n, uint32_t group_len, uint32_t *tables,
                                   uint32_t *status, size_t status_words, int compute_units, uint32_t *msd_counts,
                                   bool msd_only, LaunchEvents ev, uint32_t key_base, uint32_t force_shift) {
    launch_digit_tables_variant<uint32_t, 8, 1024, 32, VRS_DT_UNROLL, 4, true>(stream, keys, n, 0, group_len, tables, status,
                                                                               status_words, compute_units, ev,
                                                                               FusedPlanArgs{}, msd_counts, msd_only ? 1u : 0u, key_base, force_shift);
    return hipGetLastError();
}

hipError_t launch_msd_plan(hipStream_t stream, uint32_t *msd_counts, MsdPlan *msd, OnesweepPlan *plan_a,
                           OnesweepPlan *plan_lsd, OnesweepPlanHead *host_head, uint32_t stamp, uint32_t n, uint32_t tile,
                           uint32_t tiles_b_cap, uint32_t local_cap, uint32_t *tables, uint32_t group_len, uint32_t tile_cap,
                           uint32_t blind_cap, const StreamCuts &cuts0, uint32_t msd_only, uint32_t max_shift, uint32_t *host_log,
                           uint32_t sub_bits) {
#define VRS_MSD_PLAN(SUB)                                                                                                     \
    hipLaunchKernelGGL(msd_plan_kernel<SUB>, dim3(1), dim3(1024), 0, stream, msd_counts, msd, plan_a, plan_lsd, host_head, stamp, \
                       n, tile, tiles_b_cap, local_cap, tables, group_len, tile_cap, blind_cap, cuts0, msd_only, max_shift, host_log)
    if (sub_bits == 6u) VRS_MSD_PLAN(6u);
    else if (sub_bits == 7u) VRS_MSD_PLAN(7u);
    else if (sub_bits == 8u) VRS_MSD_PLAN(8u);
    else return hipErrorInvalidValue;
#undef VRS_MSD_PLAN
    return hipGetLastError();
}

hipError_t launch_msd_pass_b(hipStream_t stream, const void *keys_in, void *keys_out, const uint32_t *values_in,
                             uint32_t *values_out, MsdPlan *msd, uint32_t *status, uint32_t tiles_b, bool atomic_rank,
                             unsigned long long xcc_map, int key_bytes, uint32_t spin_budget, LaunchEvents ev, uint32_t key_base,
                             uint32_t sub_bits, bool reserve) {
    if (tiles_b == 0) return hipSuccess;
    if (sub_bits < 6u || sub_bits > 8u) return hipErrorInvalidValue;
    if (key_bytes == 8 && values_in != nullptr) return hipErrorInvalidValue;
    const dim3 grid(8 * tiles_b), block(512);
#define VRS_PASS_B(K, ITEMS, RANK, PAIRS, RESERVE)                                                                         \
    VRS_LAUNCH((msd_pass_b_kernel<K, ITEMS, RANK, PAIRS, RESERVE>), grid, block, stream, ev, static_cast<const K *>(keys_in), \
               static_cast<K *>(keys_out), values_in, values_out, msd, status, xcc_map, spin_budget, key_base, sub_bits)
    // (the hybrid form runs only with the LDS-atomic ranking; bare keys may take their places by reservation)
    if (key_bytes == 8) {
        if (!atomic_rank) VRS_PASS_B(uint64_t, 8, RANK_BALLOT, false, false);
        else if (reserve) VRS_PASS_B(uint64_t, 8, RANK_ATOMIC, false, true);
        else VRS_PASS_B(uint64_t, 8, RANK_ATOMIC, false, false);
    } else if (values_in != nullptr) {
        if (atomic_rank) VRS_PASS_B(uint32_t, 16, RANK_ATOMIC, true, false); else VRS_PASS_B(uint32_t, 16, RANK_BALLOT, true, false);
    } else {
        if (!atomic_rank) VRS_PASS_B(uint32_t, 16, RANK_BALLOT, false, false);
        else if (reserve) VRS_PASS_B(uint32_t, 16, RANK_ATOMIC, false, true);
        else VRS_PASS_B(uint32_t, 16, RANK_ATOMIC, false, false);
    }
#undef VRS_PASS_B
    return hipGetLastError();
}

hipError_t launch_msd_count_u64(hipStream_t stream, const void *keys, uint32_t n, uint32_t group_len, uint32_t *status,
                                size_t status_words, int compute_units, uint32_t *msd_counts, LaunchEvents ev) {
    const uint32_t wgs = static_cast<uint32_t>(compute_units);
    const uint32_t slices = floor_pow2(wgs / 8u > 0 ? wgs / 8u : 1u);
    VRS_LAUNCH(msd_count_u64_kernel, dim3(8 * slices), dim3(1024), stream, ev, static_cast<const uint64_t *>(keys), n, group_len,
               slices, reinterpret_cast<uint4 *>(status), static_cast<uint32_t>(status_words / 4), msd_counts,
               msd_counts + kMsdBuckets);
    return hipGetLastError();
}

hipError_t launch_msd_local_sort_u64(hipStream_t stream, void *keys, MsdPlan *msd, uint32_t max_bucket, LaunchEvents ev,
                                     uint32_t *clear_status, size_t clear_words) {
    if (max_bucket > kLocalCapBig) return hipErrorInvalidValue;  // the plan would have refused
    const StatusClear sc{reinterpret_cast<uint4 *>(clear_status), static_cast<uint32_t>(clear_words / 4)};
    if (max_bucket > kLocalCap)
        VRS_LAUNCH(msd_local_sort_u64_kernel<kLocalPairThreadsBig>, dim3(kMsdBuckets), dim3(kLocalPairThreadsBig), stream, ev, static_cast<uint64_t *>(keys), msd, sc, &msd->cursor_a[0][0]);
    else
        VRS_LAUNCH(msd_local_sort_u64_kernel<kLocalPairThreads>, dim3(kMsdBuckets), dim3(kLocalPairThreads), stream, ev, static_cast<uint64_t *>(keys), msd, sc, &msd->cursor_a[0][0]);
    return hipGetLastError();
}

hipError_t launch_msd_local_sort(hipStream_t stream, uint32_t *keys, uint32_t *values, MsdPlan *msd, uint32_t max_bucket,
                                 LaunchEvents ev, uint32_t *clear_status, size_t clear_words) {
    if (max_bucket > msd_local_capacity(values != nullptr)) return hipErrorInvalidValue;  // the plan would have refused
    const StatusClear sc{reinterpret_cast<uint4 *>(clear_status), static_cast<uint32_t>(clear_words / 4)};
    if (values != nullptr && max_bucket > kLocalCap)
        VRS_LAUNCH(msd_local_sort_pairs_kernel<kLocalPairThreadsBig>, dim3(kMsdBuckets), dim3(kLocalPairThreadsBig), stream, ev, keys, values, msd, sc, &msd->cursor_a[0][0]);
    else if (values != nullptr)
        VRS_LAUNCH(msd_local_sort_pairs_kernel<kLocalPairThreads>, dim3(kMsdBuckets), dim3(kLocalPairThreads), stream, ev, keys, values, msd, sc, &msd->cursor_a[0][0]);
    else if (max_bucket <= kWaveCap)
        VRS_LAUNCH(msd_local_sort_wave_kernel, dim3(kMsdBuckets), dim3(64), stream, ev, keys, msd, sc, &msd->cursor_a[0][0]);
    else if (max_bucket > kLeanCap)
        VRS_LAUNCH(msd_local_sort_keys_kernel<512>, dim3(kMsdBuckets), dim3(512), stream, ev, keys, msd, sc, &msd->cursor_a[0][0]);
    else
        VRS_LAUNCH(msd_local_sort_keys_kernel<256>, dim3(kMsdBuckets), dim3(256), stream, ev, keys, msd, sc, &msd->cursor_a[0][0]);
    return hipGetLastError();
}

uint32_t msd_local_capacity_small() { return kLeanCap; }
uint32_t msd_local_capacity_wave() { return kWaveCap; }
uint32_t msd_local_capacity(bool pairs_or_wide) { return pairs_or_wide ? kLocalCapBig : kLeanBigCap; }
uint32_t msd_local_capacity_pairs_small() { return kLocalCap; }

hipError_t launch_digit_tables(hipStream_t stream, const void *keys, uint32_t n, int key_bytes, uint32_t base_shift,
                               uint32_t group_len, uint32_t groups, uint32_t *tables, uint32_t *status,
                               size_t status_words, int compute_units, LaunchEvents ev, const FusedPlan *fused) {
    FusedPlanArgs fp{};
    if (fused) {
        fp.plan = fused->plan;
        fp.host_head = fused->host_head;
        fp.done = fused->done;
        fp.stamp = fused->stamp;
        fp.tile = fused->tile;
        fp.tile_cap = fused->tile_cap;
        fp.blind_cap = fused->blind_cap;
        fp.cuts0 = fused->cuts0;
    }
#define VRS_DT(K, G, T, C, U, O) \
    launch_digit_tables_variant<K, G, T, C, U, O>(stream, keys, n, base_shift, group_len, tables, status, status_words, compute_units, ev, fp)
    // (THREADS, COPIES, UNROLL, OCC): one 1024-thread workgroup per CU (32 groups: 131 KiB of LDS counters, 8 groups: 57)
    if (key_bytes == 8) {
        if (groups == 32) VRS_DT(uint64_t, 32, 1024, 32, VRS_DT_UNROLL, 4);
        else if (groups == 16) VRS_DT(uint64_t, 16, 1024, 32, VRS_DT_UNROLL, 4);
        else if (groups == 8) VRS_DT(uint64_t, 8, 1024, 32, VRS_DT_UNROLL, 4);
        else return hipErrorInvalidValue;
    } else {
        if (groups == 32) VRS_DT(uint32_t, 32, 1024, 32, VRS_DT_UNROLL, 4);
        else if (groups == 16) VRS_DT(uint32_t, 16, 1024, 32, VRS_DT_UNROLL, 4);
        else if (groups == 8) VRS_DT(uint32_t, 8, 1024, 32, VRS_DT_UNROLL, 4);
        else return hipErrorInvalidValue;
    }
#undef VRS_DT
    return hipGetLastError();
}

StreamCuts pass0_stream_cuts(uint32_t n, uint32_t group_len, uint32_t groups) {
    const auto start_of = [&](uint32_t g) -> uint32_t {
        const uint64_t a = static_cast<uint64_t>(g) * group_len;
        return static_cast<uint32_t>(a < n ? a : n);
    };
    StreamCuts c;
    c.first_group[0] = 0;
    for (uint32_t k = 1; k < static_cast<uint32_t>(kStreams); ++k) c.first_group[k] = balanced_cut(start_of, n, k, groups);
    c.first_group[kStreams] = groups;
    return c;
}

hipError_t launch_plan(hipStream_t stream, uint32_t *tables, OnesweepPlan *plan, OnesweepPlanHead *host_head,
                       uint32_t stamp, uint32_t n, uint32_t group_len, uint32_t groups, uint32_t tile, uint32_t tile_cap,
                       uint32_t blind_cap, const StreamCuts &cuts0) {
    const dim3 grid(1), block(4 * kBins);
    if (groups == 32)
        hipLaunchKernelGGL(plan_kernel<32>, grid, block, 0, stream, tables, plan, host_head, stamp, n, group_len, tile, tile_cap, blind_cap, cuts0);
    else if (groups == 16)
        hipLaunchKernelGGL(plan_kernel<16>, grid, block, 0, stream, tables, plan, host_head, stamp, n, group_len, tile, tile_cap, blind_cap, cuts0);
    else if (groups == 8)
        hipLaunchKernelGGL(plan_kernel<8>, grid, block, 0, stream, tables, plan, host_head, stamp, n, group_len, tile, tile_cap, blind_cap, cuts0);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_onesweep_scatter(hipStream_t stream, const void *keys_in, void *keys_out, const uint32_t *values_in,
                                   uint32_t *values_out, const OnesweepPlan *plan, uint32_t pass, uint32_t shift,
                                   uint32_t *status, uint32_t grid_tiles, int forced, bool atomic_rank,
                                   unsigned long long xcc_map, int key_bytes, uint32_t spin_budget, int hold_tile,
                                   LaunchEvents ev, bool misplace, uint32_t key_base, MsdPlan *reserve) {
    const int mis = misplace ? 1 : 0, force = forced;
    if (grid_tiles == 0) return hipSuccess;
    const dim3 grid(kStreams * grid_tiles), block(64 * VRS_LB_WAVES);
    const bool pairs = values_in != nullptr;
#define VRS_ONESWEEP_R(K, ITEMS, PAIRS, RANK, RESERVE)                                                                \
    VRS_LAUNCH((onesweep_scatter_kernel<K, ITEMS, VRS_LB_WAVES, PAIRS, RANK, 4, RESERVE>), grid, block, stream, ev,   \
               static_cast<const K *>(keys_in), static_cast<K *>(keys_out), values_in, values_out, plan, pass, force,  \
               shift, status, xcc_map, mis, spin_budget, hold_tile, key_base, reserve)
#define VRS_ONESWEEP(K, ITEMS, PAIRS, RANK) VRS_ONESWEEP_R(K, ITEMS, PAIRS, RANK, false)
    // the first MSD pass of the hybrid form over bare keys (LDS-atomic ranking) may take its places by reservation
    if (reserve != nullptr && !pairs && atomic_rank && shift == kShiftFromPlan) {
        if (key_bytes == 8) VRS_ONESWEEP_R(uint64_t, 8, false, RANK_ATOMIC, true);
        else VRS_ONESWEEP_R(uint32_t, VRS_LB_ITEMS, false, RANK_ATOMIC, true);
    } else if (key_bytes == 8 && pairs) {  // uint64 keys + uint32 payloads: 4096-pair tiles (32 KB of keys + 16 KB of payloads in LDS)
        if (atomic_rank) VRS_ONESWEEP(uint64_t, 8, true, RANK_ATOMIC); else VRS_ONESWEEP(uint64_t, 8, true, RANK_BALLOT);
    } else if (key_bytes == 8) {
        if (atomic_rank) VRS_ONESWEEP(uint64_t, 8, false, RANK_ATOMIC); else VRS_ONESWEEP(uint64_t, 8, false, RANK_BALLOT);
    } else if (pairs) {
        if (atomic_rank) VRS_ONESWEEP(uint32_t, VRS_LB_ITEMS, true, RANK_ATOMIC); else VRS_ONESWEEP(uint32_t, VRS_LB_ITEMS, true, RANK_BALLOT);
    } else {
        if (atomic_rank) VRS_ONESWEEP(uint32_t, VRS_LB_ITEMS, false, RANK_ATOMIC); else VRS_ONESWEEP(uint32_t, VRS_LB_ITEMS, false, RANK_BALLOT);
    }
#undef VRS_ONESWEEP
#undef VRS_ONESWEEP_R
    return hipGetLastError();
}

hipError_t launch_single(hipStream_t stream, uint32_t *buffer0, uint32_t *buffer1, uint32_t n, LaunchEvents ev) {
    if (n == 0) return hipSuccess;
    VRS_LAUNCH(single_kernel, dim3(1), dim3(kThreads), stream, ev, buffer0, buffer1, n);
    return hipGetLastError();
}

}  // namespace vrs
