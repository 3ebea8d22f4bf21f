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
#include "vrs_kernels.h"

namespace vrs {

constexpr int kBins = 256;     // RADIX_SORT_BINS
constexpr int kThreads = 256;  // 4 wave64 per workgroup
constexpr int kWaves = kThreads / 64;

__device__ __forceinline__ uint32_t lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// number of set bits of `mask` strictly below the calling lane
__device__ __forceinline__ uint32_t count_below(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

__device__ __forceinline__ uint32_t digit_of(uint32_t key, uint32_t shift) {
    return (key >> shift) & (kBins - 1);
}

// Observed dispatch places workgroup b on XCD b % 8 (speed only, never correctness).  Remap so
// that XCD x walks a CONTIGUOUS range of tiles: the partial cache lines at the two ends of every
// digit run are then completed by the neighbouring tile inside the SAME L2.
__device__ __forceinline__ uint32_t xcd_contiguous_tile(uint32_t b, uint32_t W) {
    const uint32_t q = W >> 3, r = W & 7u;
    const uint32_t xcd = b & 7u, idx = b >> 3;
    const uint32_t base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + idx;
}

// ---------------------------------------------------------------------------------------------
// 256-thread exclusive scan (one value per thread).  s_tmp: kWaves words of LDS.
// Contains one __syncthreads(); callers must separate consecutive uses by another barrier.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *s_tmp, uint32_t lane,
                                                         uint32_t wave) {
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if (lane >= static_cast<uint32_t>(o)) incl += t;
    }
    if (lane == 63u) s_tmp[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int j = 0; j < kWaves; ++j) base += (static_cast<uint32_t>(j) < wave) ? s_tmp[j] : 0u;
    return base + incl - v;
}

// ---------------------------------------------------------------------------------------------
// K1: per-tile digit histogram.
// One LDS counter per digit; a wave whose 64 keys all carry the same digit (sorted / constant /
// zero upper bytes -- the reference's own 28-bit keys make pass 3 mostly that) is detected with one
// __ballot and collapsed into a single ds_add of __popcll(mask) instead of a 64-way same-address
// atomic.
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

template <int UNROLL>
__global__ __launch_bounds__(kThreads) void histogram_kernel(const uint32_t *__restrict__ keys,
                                                             uint32_t *__restrict__ hist, uint32_t n,
                                                             uint32_t shift, uint32_t W, uint32_t B) {
    __shared__ uint32_t s_hist[kBins];
    const uint32_t tid = threadIdx.x;
    const uint32_t w = blockIdx.x;
    s_hist[tid] = 0;
    __syncthreads();

    const uint64_t tile_begin = static_cast<uint64_t>(w) * B * kThreads;
    if (tile_begin < n) {
        const uint64_t tile_keys = static_cast<uint64_t>(B) * kThreads;
        const uint32_t len = static_cast<uint32_t>(tile_begin + tile_keys <= n ? tile_keys : n - tile_begin);
        // tile_begin is a multiple of 256 keys = 1 KiB, so 16-byte loads are aligned whenever the
        // buffer base is (checked by the host).
        const uint4 *v = reinterpret_cast<const uint4 *>(keys + tile_begin);
        const uint32_t nvec = len >> 2;
        for (uint32_t i0 = 0; i0 < nvec; i0 += kThreads * UNROLL) {
            uint4 q[UNROLL];
            bool ok[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const uint32_t i = i0 + u * kThreads + tid;
                ok[u] = i < nvec;
                q[u] = ok[u] ? v[i] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                histogram_count(s_hist, q[u].x, shift, ok[u]);
                histogram_count(s_hist, q[u].y, shift, ok[u]);
                histogram_count(s_hist, q[u].z, shift, ok[u]);
                histogram_count(s_hist, q[u].w, shift, ok[u]);
            }
        }
        const uint32_t tail = (nvec << 2) + tid;  // at most 3 keys
        if (tail < len) atomicAdd(&s_hist[digit_of(keys[tile_begin + tail], shift)], 1u);
    }
    __syncthreads();
    hist[static_cast<size_t>(w) * kBins + tid] = s_hist[tid];
}

// ---------------------------------------------------------------------------------------------
// K2: offsets from the [W][256] table in O(W*256) (the reference re-sums the whole table in every
// workgroup: O(W^2*256), multi_radixsort.comp:58-62).  Tiles are grouped in chunks of C.
//   chunk_sum_kernel : chunk_sums[g][d] = sum of hist rows of chunk g
//   offsets_kernel   : base_d = excl_scan_d(sum_g chunk_sums[g][d]); walks chunk g's rows writing
//                      offsets[w][d] = base_d + (rows before w)
__global__ __launch_bounds__(kThreads) void chunk_sum_kernel(const uint32_t *__restrict__ hist,
                                                             uint32_t *__restrict__ chunk_sums, uint32_t W,
                                                             uint32_t C) {
    const uint32_t d = threadIdx.x;
    const uint32_t row0 = blockIdx.x * C;
    const uint32_t rows = min(C, W - row0);
    const uint32_t *p = hist + static_cast<size_t>(row0) * kBins + d;
    uint32_t s = 0;
    uint32_t j = 0;
    for (; j + 8 <= rows; j += 8) {
        uint32_t t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = p[static_cast<size_t>(j + u) * kBins];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; j < rows; ++j) s += p[static_cast<size_t>(j) * kBins];
    chunk_sums[static_cast<size_t>(blockIdx.x) * kBins + d] = s;
}

__global__ __launch_bounds__(kThreads) void offsets_kernel(const uint32_t *__restrict__ hist,
                                                           const uint32_t *__restrict__ chunk_sums,
                                                           uint32_t *__restrict__ offsets, uint32_t W, uint32_t C,
                                                           uint32_t G) {
    __shared__ uint32_t s_tmp[kWaves];
    const uint32_t d = threadIdx.x;
    const uint32_t g = blockIdx.x;
    uint32_t before = 0, total = 0;
    for (uint32_t j = 0; j < G; ++j) {
        const uint32_t v = chunk_sums[static_cast<size_t>(j) * kBins + d];
        before += (j < g) ? v : 0u;
        total += v;
    }
    const uint32_t base = block_exclusive_scan(total, s_tmp, d & 63u, d >> 6);
    uint32_t run = base + before;
    const uint32_t row0 = g * C;
    const uint32_t rows = min(C, W - row0);
    const uint32_t *p = hist + static_cast<size_t>(row0) * kBins + d;
    uint32_t *o = offsets + static_cast<size_t>(row0) * kBins + d;
    uint32_t j = 0;
    for (; j + 8 <= rows; j += 8) {
        uint32_t t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = p[static_cast<size_t>(j + u) * kBins];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            o[static_cast<size_t>(j + u) * kBins] = run;
            run += t[u];
        }
    }
    for (; j < rows; ++j) {
        o[static_cast<size_t>(j) * kBins] = run;
        run += p[static_cast<size_t>(j) * kBins];
    }
}

// ---------------------------------------------------------------------------------------------
// K3 building block: stable scatter of one chunk of <= ITEMS*256 keys.
//
// Layout in the chunk ("wave-striped"): wave v owns the contiguous segment
// [v*ITEMS*64, (v+1)*ITEMS*64); its item i, lane l is key index v*ITEMS*64 + i*64 + l, so every
// load instruction of a wave covers 256 contiguous bytes and (wave, item, lane) order == input
// order, which is what stability needs.
//
// Ranking (per wave, per item): eight __ballot votes -- one per digit bit -- give each lane the
// 64-bit mask of lanes holding the same digit ("match-any").  rank-in-wave = per-wave LDS counter
// of that digit + number of matching lanes below me (mbcnt); the highest matching lane bumps the
// counter by __popcll(mask).  LDS operations of one wave execute in order, so item i+1 sees item
// i's update without a barrier.
template <int ITEMS>
struct ChunkSmem {
    uint32_t keys[ITEMS * kThreads];  // re-bucketed keys (then payloads), chunk order by digit
    uint32_t whist[kWaves][kBins];    // per-wave digit counters -> per-wave digit start positions
    uint32_t gbase[kBins];            // global offset of digit d minus its start inside the chunk
    uint32_t scan_tmp[kWaves];
};

__device__ __forceinline__ uint64_t match_any_digit(uint32_t d) {
    uint64_t peers = ~0ull;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

// `run_off`: thread t's running global offset of digit t (advanced by this chunk's count of t).
// `valid`: number of real keys in the chunk (the rest is padding that sorts behind everything).
template <int ITEMS, bool PAIRS>
__device__ __forceinline__ void scatter_chunk(ChunkSmem<ITEMS> &sm, const uint32_t *kin, const uint32_t *vin,
                                              uint32_t *kout, uint32_t *vout, uint32_t valid, uint32_t shift,
                                              uint32_t &run_off) {
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;

#pragma unroll
    for (int v = 0; v < kWaves; ++v) sm.whist[v][tid] = 0;

    uint32_t key[ITEMS];
    const uint32_t seg = wave * (ITEMS * 64) + lane;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t idx = seg + i * 64;
        // padding key 0xFFFFFFFF has digit 255 under every shift and the highest chunk indices,
        // so it ranks behind every real key and is never stored
        key[i] = idx < valid ? kin[idx] : 0xFFFFFFFFu;
    }
    __syncthreads();

    uint32_t rank[ITEMS];
    uint32_t *my_hist = sm.whist[wave];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t d = digit_of(key[i], shift);
        const uint64_t peers = match_any_digit(d);
        const uint32_t below = count_below(peers);
        const uint32_t prev = my_hist[d];
        rank[i] = prev + below;
        if (below + 1u == static_cast<uint32_t>(__popcll(peers))) my_hist[d] = prev + below + 1u;
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();

    // thread t == digit t: digit starts inside the chunk, per-wave starts, global base
    {
        uint32_t c[kWaves];
        uint32_t total = 0;
#pragma unroll
        for (int v = 0; v < kWaves; ++v) {
            c[v] = sm.whist[v][tid];
            total += c[v];
        }
        const uint32_t excl = block_exclusive_scan(total, sm.scan_tmp, lane, wave);
        uint32_t acc = excl;
#pragma unroll
        for (int v = 0; v < kWaves; ++v) {
            sm.whist[v][tid] = acc;
            acc += c[v];
        }
        sm.gbase[tid] = run_off - excl;
        run_off += total;
    }
    __syncthreads();

    uint32_t pos[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        pos[i] = my_hist[digit_of(key[i], shift)] + rank[i];
        sm.keys[pos[i]] = key[i];
    }
    __syncthreads();

    uint32_t dst[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t p = i * kThreads + tid;
        const uint32_t k = sm.keys[p];
        dst[i] = sm.gbase[digit_of(k, shift)] + p;
        if (p < valid) kout[dst[i]] = k;
    }

    if constexpr (PAIRS) {
        uint32_t val[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t idx = seg + i * 64;
            val[i] = idx < valid ? vin[idx] : 0u;
        }
        __syncthreads();  // everyone has read its keys back
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) sm.keys[pos[i]] = val[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t p = i * kThreads + tid;
            if (p < valid) vout[dst[i]] = sm.keys[p];
        }
        __syncthreads();  // sm.keys is reused by the next chunk's keys
    }
}

template <int ITEMS, bool PAIRS>
__global__ __launch_bounds__(kThreads) void scatter_kernel(const uint32_t *__restrict__ keys_in,
                                                           uint32_t *__restrict__ keys_out,
                                                           const uint32_t *__restrict__ values_in,
                                                           uint32_t *__restrict__ values_out,
                                                           const uint32_t *__restrict__ offsets, uint32_t n,
                                                           uint32_t shift, uint32_t W, uint32_t B, int xcd_remap) {
    __shared__ ChunkSmem<ITEMS> sm;
    const uint32_t w = xcd_remap ? xcd_contiguous_tile(blockIdx.x, W) : blockIdx.x;
    const uint64_t tile_begin = static_cast<uint64_t>(w) * B * kThreads;
    if (tile_begin >= n) return;  // uniform per workgroup
    const uint64_t tile_keys = static_cast<uint64_t>(B) * kThreads;
    const uint32_t tile_len = static_cast<uint32_t>(tile_begin + tile_keys <= n ? tile_keys : n - tile_begin);
    uint32_t run_off = offsets[static_cast<size_t>(w) * kBins + threadIdx.x];
    constexpr uint32_t kChunk = ITEMS * kThreads;
    for (uint32_t c0 = 0; c0 < tile_len; c0 += kChunk) {
        const uint32_t valid = min(kChunk, tile_len - c0);
        scatter_chunk<ITEMS, PAIRS>(sm, keys_in + tile_begin + c0, PAIRS ? values_in + tile_begin + c0 : nullptr,
                                    keys_out, values_out, valid, shift, run_off);
    }
}

// ---------------------------------------------------------------------------------------------
// K4: single_radixsort -- one workgroup, four passes in one launch
// (single_radixsort.comp:42-140).  Even passes buffer0 -> buffer1, odd passes back; result in
// buffer0.  Same ranking machinery as K3 with a small chunk.
constexpr int kSingleItems = 4;

__global__ __launch_bounds__(kThreads) void single_kernel(uint32_t *buffer0, uint32_t *buffer1, uint32_t n) {
    __shared__ ChunkSmem<kSingleItems> sm;
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
        for (uint32_t c0 = 0; c0 < n; c0 += kChunk) {
            const uint32_t valid = min(kChunk, n - c0);
            scatter_chunk<kSingleItems, false>(sm, in + c0, nullptr, out, nullptr, valid, shift, run_off);
        }
        // the next pass reads what this pass wrote: same CU, so a workgroup barrier (with its
        // workgroup-scope fence) orders the global stores before the loads
        __threadfence_block();
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// host-side launch wrappers

uint32_t prefix_chunk_tiles(uint32_t W) {
    uint32_t c = 1;
    while (static_cast<uint64_t>(c) * c < W) c <<= 1;
    return c;
}

hipError_t launch_histograms(hipStream_t stream, const uint32_t *keys_in, uint32_t *hist, uint32_t n,
                             uint32_t shift, uint32_t W, uint32_t B) {
    if (W == 0) return hipSuccess;
    hipLaunchKernelGGL(histogram_kernel<8>, dim3(W), dim3(kThreads), 0, stream, keys_in, hist, n, shift, W, B);
    return hipGetLastError();
}

hipError_t launch_prefix(hipStream_t stream, const uint32_t *hist, const PrefixScratch &scratch, uint32_t W) {
    if (W == 0) return hipSuccess;
    const uint32_t C = prefix_chunk_tiles(W);
    const uint32_t G = (W + C - 1) / C;
    hipLaunchKernelGGL(chunk_sum_kernel, dim3(G), dim3(kThreads), 0, stream, hist, scratch.chunk_sums, W, C);
    hipLaunchKernelGGL(offsets_kernel, dim3(G), dim3(kThreads), 0, stream, hist, scratch.chunk_sums,
                       scratch.offsets, W, C, G);
    return hipGetLastError();
}

template <int ITEMS>
static void launch_scatter_items(hipStream_t stream, const uint32_t *keys_in, uint32_t *keys_out,
                                 const uint32_t *values_in, uint32_t *values_out, const uint32_t *offsets,
                                 uint32_t n, uint32_t shift, uint32_t W, uint32_t B, bool xcd_remap) {
    if (values_in != nullptr) {
        hipLaunchKernelGGL((scatter_kernel<ITEMS, true>), dim3(W), dim3(kThreads), 0, stream, keys_in, keys_out,
                           values_in, values_out, offsets, n, shift, W, B, xcd_remap ? 1 : 0);
    } else {
        hipLaunchKernelGGL((scatter_kernel<ITEMS, false>), dim3(W), dim3(kThreads), 0, stream, keys_in, keys_out,
                           values_in, values_out, offsets, n, shift, W, B, xcd_remap ? 1 : 0);
    }
}

hipError_t launch_scatter(hipStream_t stream, const uint32_t *keys_in, uint32_t *keys_out,
                          const uint32_t *values_in, uint32_t *values_out, const uint32_t *offsets, uint32_t n,
                          uint32_t shift, uint32_t W, uint32_t B, bool xcd_remap) {
    if (W == 0) return hipSuccess;
    // chunk = ITEMS*256 keys held in registers + LDS at once; a tile of B blocks is walked in
    // ceil(B/ITEMS) chunks
    if (B >= 32)
        launch_scatter_items<32>(stream, keys_in, keys_out, values_in, values_out, offsets, n, shift, W, B, xcd_remap);
    else if (B >= 16)
        launch_scatter_items<16>(stream, keys_in, keys_out, values_in, values_out, offsets, n, shift, W, B, xcd_remap);
    else if (B >= 8)
        launch_scatter_items<8>(stream, keys_in, keys_out, values_in, values_out, offsets, n, shift, W, B, xcd_remap);
    else
        launch_scatter_items<4>(stream, keys_in, keys_out, values_in, values_out, offsets, n, shift, W, B, xcd_remap);
    return hipGetLastError();
}

hipError_t launch_single(hipStream_t stream, uint32_t *buffer0, uint32_t *buffer1, uint32_t n) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(single_kernel, dim3(1), dim3(kThreads), 0, stream, buffer0, buffer1, n);
    return hipGetLastError();
}

}  // namespace vrs
