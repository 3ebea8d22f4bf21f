// Lab (round 3): the hybrid form's local sort (one workgroup per bucket of ~6100 keys, 18 low bits, two 9-bit LDS passes).
// Round 2 left it at 207 us for 800 MB (floor: load + store alone 126 us).  Variants timed here on the same input:
//   base      the product kernel's structure (4-byte striped loads / LDS reads, one table in pass 1, per-wave tables in pass 2)
//   vec       16-byte global loads / stores and 16-byte LDS reads: pass 1 writes through a permuted LDS layout so that a lane's
//             ds_read_b128 returns its four wave-striped items of pass 2; pass 2 writes the natural layout (shifted by the bucket's
//             misalignment) so that the final read is a ds_read_b128 and the store a global_store_dwordx4
//   pf        vec + persistent workgroups that load the NEXT bucket into registers before they sort the current one
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/local_sort_lab2.hip -o tools/lab/local_sort_lab2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int BITS = 9, BINS = 1 << BITS;

// ---------------------------------------------------------------------------------------------- base (product structure)
template <int THREADS, int ITEMS, bool STABLE>
__device__ __forceinline__ void base_pass(uint32_t (&key)[ITEMS], uint32_t *s_keys, uint32_t *s_hist, uint32_t *s_tmp, uint32_t shift, uint32_t n) {
    constexpr int WAVES = THREADS / 64, TABLES = STABLE ? WAVES : 1, PER = BINS / THREADS;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t c = tid; c < TABLES * BINS; c += THREADS) s_hist[c] = 0;
    __syncthreads();
    uint32_t *my = s_hist + (STABLE ? wave * BINS : 0u);
    const uint32_t seg = wave * (ITEMS * 64) + lane;
    uint32_t rank[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        rank[i] = seg + i * 64;
        if (rank[i] < n) rank[i] = __hip_atomic_fetch_add(&my[(key[i] >> shift) & (BINS - 1)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    {
        uint32_t c[TABLES][PER], total = 0;
#pragma unroll
        for (int v = 0; v < TABLES; ++v) {
            const uint2 q = reinterpret_cast<const uint2 *>(s_hist + v * BINS)[tid];
            c[v][0] = q.x; c[v][1] = q.y; total += q.x + q.y;
        }
        uint32_t incl = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (lane >= (uint32_t)o) incl += t; }
        if (lane == 63u) s_tmp[wave] = incl;
        __syncthreads();
        uint32_t acc = incl - total;
#pragma unroll
        for (int v = 0; v < WAVES; ++v) acc += ((uint32_t)v < wave) ? s_tmp[v] : 0u;
        uint32_t out[TABLES][PER];
#pragma unroll
        for (int p_ = 0; p_ < PER; ++p_)
#pragma unroll
            for (int v = 0; v < TABLES; ++v) { out[v][p_] = acc; acc += c[v][p_]; }
#pragma unroll
        for (int v = 0; v < TABLES; ++v) reinterpret_cast<uint2 *>(s_hist + v * BINS)[tid] = make_uint2(out[v][0], out[v][1]);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) if (seg + i * 64 < n) rank[i] += my[(key[i] >> shift) & (BINS - 1)];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) if (seg + i * 64 < n) s_keys[rank[i]] = key[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) key[i] = s_keys[seg + i * 64];
    __syncthreads();
}
template <int THREADS, int ITEMS>
__device__ __forceinline__ void base_bucket(uint32_t *bucket, uint32_t n, uint32_t *s_keys, uint32_t *s_hist, uint32_t *s_tmp) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t key[ITEMS];
    const uint32_t seg = wave * (ITEMS * 64) + lane;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) { const uint32_t idx = seg + i * 64; key[i] = bucket[idx < n ? idx : n - 1u]; }
    base_pass<THREADS, ITEMS, false>(key, s_keys, s_hist, s_tmp, 0, n);
    base_pass<THREADS, ITEMS, true>(key, s_keys, s_hist, s_tmp, BITS, n);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) { const uint32_t idx = seg + i * 64; if (idx < n) bucket[idx] = key[i]; }
}
__global__ __launch_bounds__(256, 4) void base_kernel(uint32_t *__restrict__ keys, const uint32_t *__restrict__ off) {
    __shared__ uint32_t s_keys[256 * 26];
    __shared__ uint32_t s_hist[4 << 9];
    __shared__ uint32_t s_tmp[8];
    const uint32_t begin = off[blockIdx.x], n = off[blockIdx.x + 1] - begin;
    if (n == 0 || n > 256 * 26) return;
    uint32_t *b = keys + begin;
    const uint32_t used = (n + 255u) / 256u;
    if (used <= 4) base_bucket<256, 4>(b, n, s_keys, s_hist, s_tmp);
    else if (used <= 8) base_bucket<256, 8>(b, n, s_keys, s_hist, s_tmp);
    else if (used <= 12) base_bucket<256, 12>(b, n, s_keys, s_hist, s_tmp);
    else if (used <= 16) base_bucket<256, 16>(b, n, s_keys, s_hist, s_tmp);
    else if (used <= 20) base_bucket<256, 20>(b, n, s_keys, s_hist, s_tmp);
    else if (used <= 24) base_bucket<256, 24>(b, n, s_keys, s_hist, s_tmp);
    else base_bucket<256, 26>(b, n, s_keys, s_hist, s_tmp);
}

// ---------------------------------------------------------------------------------------------- vec
// LDS index space q = (global key index) - abase, abase = the bucket's begin rounded down to 4 keys: q in [mis, mis + n).
// Thread t holds VEC vectors; vector j covers q = 4 * (j * THREADS + t) + c, c = 0..3.
template <int MAXVEC>
struct VecKeysT { uint32_t w[4 * MAXVEC]; };  // plain words: vector types in a struct ended up in scratch

template <int THREADS, int VEC, typename VK>
__device__ __forceinline__ void vec_load(VK &k, const uint32_t *abase, uint32_t nvec) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const uint32_t v = j * THREADS + threadIdx.x;
        const uint4 t = reinterpret_cast<const uint4 *>(abase)[v < nvec ? v : nvec - 1u];  // unpredicated, clamped
        k.w[4 * j] = t.x; k.w[4 * j + 1] = t.y; k.w[4 * j + 2] = t.z; k.w[4 * j + 3] = t.w;
    }
}

// scan of TABLES tables of 512 bins by THREADS = 256 threads (2 bins each): table-major inside a bin (bin b of table 0, 1, ...)
template <int THREADS, int TABLES>
__device__ __forceinline__ void scan_tables(uint32_t *s_hist, uint32_t *s_tmp) {
    constexpr int WAVES = THREADS / 64, PER = BINS / THREADS;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t c[TABLES][PER], total = 0;
#pragma unroll
    for (int v = 0; v < TABLES; ++v) {
        if constexpr (PER == 2) {
            const uint2 q = reinterpret_cast<const uint2 *>(s_hist + v * BINS)[tid];
            c[v][0] = q.x; c[v][1] = q.y; total += q.x + q.y;
        } else {
            c[v][0] = s_hist[v * BINS + tid]; total += c[v][0];
        }
    }
    uint32_t incl = total;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (lane >= (uint32_t)o) incl += t; }
    if (lane == 63u) s_tmp[wave] = incl;
    __syncthreads();
    uint32_t acc = incl - total;
#pragma unroll
    for (int v = 0; v < WAVES; ++v) acc += ((uint32_t)v < wave) ? s_tmp[v] : 0u;
    uint32_t out[TABLES][PER];
#pragma unroll
    for (int p_ = 0; p_ < PER; ++p_)
#pragma unroll
        for (int v = 0; v < TABLES; ++v) { out[v][p_] = acc; acc += c[v][p_]; }
#pragma unroll
    for (int v = 0; v < TABLES; ++v) {
        if constexpr (PER == 2) reinterpret_cast<uint2 *>(s_hist + v * BINS)[tid] = make_uint2(out[v][0], out[v][1]);
        else s_hist[v * BINS + tid] = out[v][0];
    }
}

// sorts the bucket held in k (as loaded by vec_load) and stores it; s_keys: THREADS * VEC * 4 words
template <int THREADS, int VEC, int MAXVEC, bool PREFETCH>
__device__ __forceinline__ void vec_sort_store(VecKeysT<MAXVEC> &k, const VecKeysT<MAXVEC> &nxt, uint32_t *abase, uint32_t mis, uint32_t n, uint32_t *s_keys, uint32_t *s_hist2, uint32_t *s_tmp) {
    constexpr int WAVES = THREADS / 64, ITEMS = 4 * VEC;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // ---- pass 1: low 9 bits, ONE table (behind pass 2's per-wave tables: all zeroed here), ties in any order
    uint32_t *s_hist = s_hist2 + WAVES * BINS;
    for (uint32_t c = tid; c < (WAVES + 1) * BINS; c += THREADS) s_hist2[c] = 0;
    __syncthreads();
    uint32_t rank[ITEMS];
#pragma unroll
    for (int j = 0; j < VEC; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t q = 4u * (j * THREADS + tid) + c;
            if (q - mis < n) rank[4 * j + c] = __hip_atomic_fetch_add(&s_hist[k.w[4 * j + c] & (BINS - 1)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    __syncthreads();
    scan_tables<THREADS, 1>(s_hist, s_tmp);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < VEC; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t q = 4u * (j * THREADS + tid) + c;
            if (q - mis < n) rank[4 * j + c] += s_hist[k.w[4 * j + c] & (BINS - 1)];
        }
    // logical position L of pass 2 (wave w owns [w * ITEMS * 64, ...), item i = 4g + c of lane t is L = w*ITEMS*64 + i*64 + t)
    // lives at word (L & ~255) | ((L & 63) << 2) | ((L >> 6) & 3): ITEMS * 64 is a multiple of 256
#pragma unroll
    for (int j = 0; j < VEC; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t q = 4u * (j * THREADS + tid) + c;
            const uint32_t L = rank[4 * j + c];
            if (q - mis < n) s_keys[(L & ~255u) | ((L & 63u) << 2) | ((L >> 6) & 3u)] = k.w[4 * j + c];
        }
    __syncthreads();
    // ---- pass 2: high 9 bits, one table per wave, stable
    const uint32_t seg = wave * (ITEMS * 64);
#pragma unroll
    for (int g = 0; g < VEC; ++g) {
        const uint4 t = reinterpret_cast<const uint4 *>(s_keys + seg + g * 256)[lane];
        k.w[4 * g] = t.x; k.w[4 * g + 1] = t.y; k.w[4 * g + 2] = t.z; k.w[4 * g + 3] = t.w;
    }
    uint32_t *my = s_hist2 + wave * BINS;
#pragma unroll
    for (int g = 0; g < VEC; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t L = seg + (4 * g + c) * 64 + lane;
            if (L < n) rank[4 * g + c] = __hip_atomic_fetch_add(&my[(k.w[4 * g + c] >> BITS) & (BINS - 1)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    __syncthreads();
    scan_tables<THREADS, WAVES>(s_hist2, s_tmp);
    __syncthreads();
#pragma unroll
    for (int g = 0; g < VEC; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t L = seg + (4 * g + c) * 64 + lane;
            if (L < n) rank[4 * g + c] += my[(k.w[4 * g + c] >> BITS) & (BINS - 1)];
        }
    // (every wave read its pass-2 keys out of s_keys before its atomics, and two barriers lie behind those: s_keys is free)
#pragma unroll
    for (int g = 0; g < VEC; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t L = seg + (4 * g + c) * 64 + lane;
            if (L < n) s_keys[mis + rank[4 * g + c]] = k.w[4 * g + c];
        }
    // the keys are in LDS now: the registers take the next bucket (its loads have had the whole sort to arrive), BEFORE this
    // bucket's stores are issued -- loads and stores retire on one in-order counter
    if constexpr (PREFETCH) {
#pragma unroll
        for (int j = 0; j < 4 * MAXVEC; ++j) k.w[j] = nxt.w[j];
    }
    __syncthreads();
    // ---- store: q = 4v + c holds sorted position q - mis
    const uint32_t nvec = (mis + n + 3u) / 4u;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const uint32_t v = j * THREADS + tid;
        if (v < nvec) {
            const uint4 q4 = reinterpret_cast<const uint4 *>(s_keys)[v];
            const uint32_t q = 4u * v;
            if (q >= mis && q + 4u <= mis + n) {
                reinterpret_cast<uint4 *>(abase)[v] = q4;
            } else {  // the two ends of the bucket: the neighbours' keys in the same 16 bytes are not ours to write
                if (q + 0u - mis < n) abase[q + 0u] = q4.x;
                if (q + 1u - mis < n) abase[q + 1u] = q4.y;
                if (q + 2u - mis < n) abase[q + 2u] = q4.z;
                if (q + 3u - mis < n) abase[q + 3u] = q4.w;
            }
        }
    }
    __syncthreads();  // s_keys is reused by the next bucket (persistent form)
}

template <int THREADS, int MAXVEC, bool PREFETCH>
__device__ __forceinline__ void vec_dispatch(VecKeysT<MAXVEC> &k, const VecKeysT<MAXVEC> &nxt, uint32_t *abase, uint32_t mis, uint32_t n, uint32_t *s_keys, uint32_t *s_hist, uint32_t *s_tmp) {
    const uint32_t used = (mis + n + 4u * THREADS - 1u) / (4u * THREADS);
#define VRS_CASE(V) case V: if constexpr (V <= MAXVEC) vec_sort_store<THREADS, V, MAXVEC, PREFETCH>(k, nxt, abase, mis, n, s_keys, s_hist, s_tmp); break;
    switch (used) {
        VRS_CASE(1) VRS_CASE(2) VRS_CASE(3) VRS_CASE(4) VRS_CASE(5) VRS_CASE(6)
        default: vec_sort_store<THREADS, MAXVEC, MAXVEC, PREFETCH>(k, nxt, abase, mis, n, s_keys, s_hist, s_tmp); break;
    }
#undef VRS_CASE
}

template <int THREADS, int MAXVEC, int OCC>
__global__ __launch_bounds__(THREADS, OCC) void vec_kernel(uint32_t *__restrict__ keys, const uint32_t *__restrict__ off) {
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[THREADS * 4 * MAXVEC];
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[(THREADS / 64 + 1) << 9];
    __shared__ uint32_t s_tmp[16];
    const uint32_t begin = off[blockIdx.x], n = off[blockIdx.x + 1] - begin;
    const uint32_t mis = static_cast<uint32_t>((reinterpret_cast<uintptr_t>(keys + begin) >> 2) & 3u);
    if (n == 0 || mis + n > THREADS * 4 * MAXVEC) return;
    uint32_t *abase = keys + begin - mis;
    VecKeysT<MAXVEC> k;
    vec_load<THREADS, MAXVEC>(k, abase, (mis + n + 3u) / 4u);
    vec_dispatch<THREADS, MAXVEC, false>(k, k, abase, mis, n, s_keys, s_hist, s_tmp);
}

// persistent: workgroup w sorts buckets w, w + grid, ...; the next bucket's keys are loaded before the current one is sorted
template <int THREADS, int MAXVEC, int OCC>
__global__ __launch_bounds__(THREADS, OCC) void pf_kernel(uint32_t *__restrict__ keys, const uint32_t *__restrict__ off, uint32_t buckets) {
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[THREADS * 4 * MAXVEC];
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[(THREADS / 64 + 1) << 9];
    __shared__ uint32_t s_tmp[16];
    uint32_t b = blockIdx.x;
    if (b >= buckets) return;
    uint32_t begin = off[b], n = off[b + 1] - begin;
    uint32_t mis = static_cast<uint32_t>((reinterpret_cast<uintptr_t>(keys + begin) >> 2) & 3u);
    VecKeysT<MAXVEC> cur, nxt;
    vec_load<THREADS, MAXVEC>(cur, keys + begin - mis, (mis + n + 3u) / 4u);
    for (;;) {
        const uint32_t b2 = b + gridDim.x;
        uint32_t begin2 = 0, n2 = 0, mis2 = 0;
        if (b2 < buckets) {
            begin2 = off[b2];
            n2 = off[b2 + 1] - begin2;
            mis2 = static_cast<uint32_t>((reinterpret_cast<uintptr_t>(keys + begin2) >> 2) & 3u);
            vec_load<THREADS, MAXVEC>(nxt, keys + begin2 - mis2, (mis2 + n2 + 3u) / 4u);
        }
        if (n != 0 && mis + n <= THREADS * 4 * MAXVEC) {
            vec_dispatch<THREADS, MAXVEC, true>(cur, nxt, keys + begin - mis, mis, n, s_keys, s_hist, s_tmp);
        } else {
#pragma unroll
            for (int j = 0; j < 4 * MAXVEC; ++j) cur.w[j] = nxt.w[j];
        }
        if (b2 >= buckets) break;
        b = b2; begin = begin2; n = n2; mis = mis2;
    }
}


// ---------------------------------------------------------------------------------------------- wave
// ONE WAVE per bucket (16 MSD bits: ~1526 keys per bucket at 10^8 keys): no workgroup barrier anywhere -- the LDS executes one
// wave's operations in order -- so the 16 waves of a CU are 16 independent instruction streams.  Two 8-bit passes.
// Per-wave LDS: WCAP key words + 256 counters.
template <int B>
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// exclusive scan of 256 counters held 4 per lane (b128); returns nothing: writes the prefixes back
__device__ __forceinline__ void wave_scan256(uint32_t *tbl, uint32_t lane) {
    uint4 c = reinterpret_cast<uint4 *>(tbl)[lane];
    const uint32_t s = c.x + c.y + c.z + c.w;
    uint32_t incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (lane >= (uint32_t)o) incl += t; }
    uint32_t acc = incl - s;
    uint4 o4;
    o4.x = acc; acc += c.x; o4.y = acc; acc += c.y; o4.z = acc; acc += c.z; o4.w = acc;
    reinterpret_cast<uint4 *>(tbl)[lane] = o4;
}
template <int VEC, int WCAP>
__device__ __forceinline__ void wave_sort(uint32_t *abase, uint32_t mis, uint32_t n, uint32_t *s_keys, uint32_t *tbl) {
    constexpr int ITEMS = 4 * VEC;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nvec = (mis + n + 3u) / 4u;
    uint32_t k[ITEMS], rank[ITEMS];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const uint32_t v = j * 64 + lane;
        const uint4 t = reinterpret_cast<const uint4 *>(abase)[v < nvec ? v : nvec - 1u];
        k[4 * j] = t.x; k[4 * j + 1] = t.y; k[4 * j + 2] = t.z; k[4 * j + 3] = t.w;
    }
    reinterpret_cast<uint4 *>(tbl)[lane] = make_uint4(0, 0, 0, 0);
    wave_fence<0>();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t q = 4u * ((i >> 2) * 64 + lane) + (i & 3);
        if (q - mis < n) rank[i] = __hip_atomic_fetch_add(&tbl[k[i] & 255u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    wave_fence<1>();
    wave_scan256(tbl, lane);
    wave_fence<2>();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t q = 4u * ((i >> 2) * 64 + lane) + (i & 3);
        if (q - mis < n) rank[i] += tbl[k[i] & 255u];
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t q = 4u * ((i >> 2) * 64 + lane) + (i & 3);
        const uint32_t L = rank[i];
        if (q - mis < n) s_keys[(L & ~255u) | ((L & 63u) << 2) | ((L >> 6) & 3u)] = k[i];
    }
    wave_fence<3>();
    reinterpret_cast<uint4 *>(tbl)[lane] = make_uint4(0, 0, 0, 0);  // behind the base reads in the LDS queue
#pragma unroll
    for (int g = 0; g < VEC; ++g) {
        const uint4 t = reinterpret_cast<const uint4 *>(s_keys + g * 256)[lane];
        k[4 * g] = t.x; k[4 * g + 1] = t.y; k[4 * g + 2] = t.z; k[4 * g + 3] = t.w;
    }
    wave_fence<4>();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t L = i * 64 + lane;
        if (L < n) rank[i] = __hip_atomic_fetch_add(&tbl[(k[i] >> 8) & 255u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    wave_fence<5>();
    wave_scan256(tbl, lane);
    wave_fence<6>();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t L = i * 64 + lane;
        if (L < n) rank[i] += tbl[(k[i] >> 8) & 255u];
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t L = i * 64 + lane;
        if (L < n) s_keys[mis + rank[i]] = k[i];
    }
    wave_fence<7>();
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const uint32_t v = j * 64 + lane;
        if (v < nvec) {
            const uint4 q4 = reinterpret_cast<const uint4 *>(s_keys)[v];
            const uint32_t q = 4u * v;
            if (q >= mis && q + 4u <= mis + n) {
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
template <int MAXVEC, int WAVES, int OCC>
__global__ __launch_bounds__(WAVES * 64, OCC) void wave_kernel(uint32_t *__restrict__ keys, const uint32_t *__restrict__ off, uint32_t buckets) {
    constexpr int WCAP = 256 * MAXVEC;
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[WAVES][WCAP];
    __shared__ __attribute__((aligned(16))) uint32_t s_tbl[WAVES][256];
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x * WAVES + wave;
    if (b >= buckets) return;
    const uint32_t begin = off[b], n = off[b + 1] - begin;
    const uint32_t mis = static_cast<uint32_t>((reinterpret_cast<uintptr_t>(keys + begin) >> 2) & 3u);
    if (n == 0 || mis + n > WCAP) return;
    uint32_t *abase = keys + begin - mis;
    const uint32_t used = (mis + n + 255u) / 256u;
#define VRS_CASE(V) case V: if constexpr (V <= MAXVEC) wave_sort<V, WCAP>(abase, mis, n, s_keys[wave], s_tbl[wave]); break;
    switch (used) {
        VRS_CASE(1) VRS_CASE(2) VRS_CASE(3) VRS_CASE(4) VRS_CASE(5) VRS_CASE(6) VRS_CASE(7)
        default: wave_sort<MAXVEC, WCAP>(abase, mis, n, s_keys[wave], s_tbl[wave]); break;
    }
#undef VRS_CASE
}


// ---------------------------------------------------------------------------------------------- lean
// The kernels above spend their time ISSUING instructions (product kernel: ~54 VALU + ~74 SALU per key and lane, most of it exec-mask
// bookkeeping of per-item predicates), not in the LDS or on memory.  Same algorithm as `vec`, written for a minimal instruction
// count: no predicated item anywhere -- slots that hold no key (before the bucket's first key in its first 16 bytes, behind its
// last key) carry digit 512, an extra counter behind the 512 real ones, picked with a compare + select in the FIRST and LAST row
// only (pass 2: in the last 16 items of a wave only) -- and counters that count BYTES (every rank is an LDS byte offset).
__device__ unsigned long long *g_marks;  // [buckets][12]
__shared__ unsigned long long s_marks[12];
#ifdef LEAN_MARKS
#define MARK(i) do { if (threadIdx.x == 0) s_marks[(i)] = __builtin_readcyclecounter(); } while (0)
#define MARK_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define MARK(i)
#define MARK_WAIT_VM()
#endif
constexpr int kLeanRow = 576;  // words per table: 512 bins + 64 dummy counters (one per lane) for the slots that hold no key
template <int THREADS, int VEC, bool GUARD>
__device__ __forceinline__ void lean_sort_store(uint32_t *abase, uint32_t mis, uint32_t n, uint32_t *s_keys, uint32_t *s_hist2, uint32_t *s_tmp) {
    constexpr int WAVES = THREADS / 64, ITEMS = 4 * VEC;
    static_assert(THREADS == 256, "two bins per thread in the scans");
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t end = mis + n;              // slots [mis, end) hold keys
    const uint32_t nvec = (end + 3u) / 4u;
    uint32_t k[ITEMS], rank[ITEMS];
    MARK(0);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        uint32_t v = j * THREADS + tid;
        if (j == VEC - 1) v = v < nvec ? v : nvec - 1u;  // only the last row can reach behind the bucket
        const uint4 t = reinterpret_cast<const uint4 *>(abase)[v];
        k[4 * j] = t.x; k[4 * j + 1] = t.y; k[4 * j + 2] = t.z; k[4 * j + 3] = t.w;
    }
    // all tables zeroed here: [WAVES] tables of pass 2, then pass 1's
    uint32_t *s_hist = s_hist2 + WAVES * kLeanRow;
    {
        constexpr uint32_t kVecs = (WAVES + 1) * kLeanRow / 4;
        for (uint32_t c = tid; c < kVecs; c += THREADS) reinterpret_cast<uint4 *>(s_hist2)[c] = make_uint4(0, 0, 0, 0);
    }
    MARK_WAIT_VM();
    __syncthreads();
    MARK(1);
    // ---- pass 1: low 9 bits, one table, ties in any order; byte address of counter d = 4 d
#pragma unroll
    for (int j = 0; j < VEC; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t a = (k[4 * j + c] << 2) & 0x7FCu;
            if (j == 0 || j == VEC - 1) {
                const uint32_t q = 4u * (j * THREADS + tid) + c;
                a = (q - mis < n) ? a : 2048u + 4u * lane;  // same-address returning adds are served lane by lane: every lane its own dummy
            }
            if constexpr (GUARD) {
                const uint32_t a0 = __builtin_amdgcn_readfirstlane(a);
                if (__ballot(a == a0) == ~0ull) {  // one counter for the whole instruction: one add of 64 keys
                    uint32_t old = 0;
                    if (lane == 0u) old = __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(s_hist) + a0), 256u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    rank[4 * j + c] = __builtin_amdgcn_readfirstlane(old) + 4u * lane;
                    continue;
                }
            }
            rank[4 * j + c] = __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(s_hist) + a), 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    __syncthreads();
    MARK(2);
    {   // exclusive prefix over the 512 bins (two per thread); the empty slots' bin starts behind the last key
        const uint2 q = reinterpret_cast<const uint2 *>(s_hist)[tid];
        const uint32_t total = q.x + q.y;
        uint32_t incl = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (lane >= (uint32_t)o) incl += t; }
        if (lane == 63u) s_tmp[wave] = incl;
        __syncthreads();
        uint32_t acc = incl - total;
#pragma unroll
        for (int v = 0; v < WAVES; ++v) acc += ((uint32_t)v < wave) ? s_tmp[v] : 0u;
        reinterpret_cast<uint2 *>(s_hist)[tid] = make_uint2(acc, acc + q.x);
    }
    __syncthreads();
    MARK(3);
#pragma unroll
    for (int j = 0; j < VEC; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t a = (k[4 * j + c] << 2) & 0x7FCu;
            if (j == 0 || j == VEC - 1) {
                const uint32_t q = 4u * (j * THREADS + tid) + c;
                // an empty slot behind the bucket keeps its place (position q: there are mis + n slots before the first of them and
                // mis empty ones among those), the ones before the bucket follow the keys (position n + q)
                const bool valid = q - mis < n;
                a = valid ? a : 2048u + 4u * lane;
                const uint32_t r = rank[4 * j + c] + *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(s_hist) + a);
                rank[4 * j + c] = valid ? r : 4u * (q < mis ? n + q : q);
                continue;
            }
            rank[4 * j + c] += *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(s_hist) + a);
        }
    // byte offset Lb = 4 L of pass 2's logical position L lives at byte (Lb & ~1023) | ((Lb & 252) << 2) | ((Lb >> 6) & 12)
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t Lb = rank[i];
        const uint32_t ph = (Lb & ~1023u) | ((Lb & 252u) << 2) | ((Lb >> 6) & 12u);
        *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(s_keys) + ph) = k[i];
    }
    __syncthreads();
    MARK(4);
    // ---- pass 2: high 9 bits, one table per wave, stable
    const uint32_t seg = wave * (ITEMS * 64);
#pragma unroll
    for (int g = 0; g < VEC; ++g) {
        const uint4 t = reinterpret_cast<const uint4 *>(s_keys + seg + g * 256)[lane];
        k[4 * g] = t.x; k[4 * g + 1] = t.y; k[4 * g + 2] = t.z; k[4 * g + 3] = t.w;
    }
    char *my = reinterpret_cast<char *>(s_hist2 + wave * kLeanRow);
    // only the last 1024 logical positions (of the whole bucket) can be empty: the last 16 items of a wave when VEC >= 4
    constexpr int kFirstMaybeEmpty = VEC >= 5 ? ITEMS - 17 : 0;  // up to 1024 + 3 empty slots
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        uint32_t a = (k[i] >> 7) & 0x7FCu;
        if (i >= kFirstMaybeEmpty) a = (seg + i * 64 + lane < n) ? a : 2048u + 4u * lane;
        if constexpr (GUARD) {
            const uint32_t a0 = __builtin_amdgcn_readfirstlane(a);
            if (__ballot(a == a0) == ~0ull) {
                uint32_t old = 0;
                if (lane == 0u) old = __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(my + a0), 256u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                rank[i] = __builtin_amdgcn_readfirstlane(old) + 4u * lane;
                continue;
            }
        }
        rank[i] = __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(my + a), 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    MARK(5);
    {   // exclusive prefix over (bin, wave), the empty slots' bin last; starts at the bucket's misalignment: pass 2 writes slot = mis + position
        uint32_t c[WAVES][2], total = 0;
#pragma unroll
        for (int v = 0; v < WAVES; ++v) {
            const uint2 q = reinterpret_cast<const uint2 *>(s_hist2 + v * kLeanRow)[tid];
            c[v][0] = q.x; c[v][1] = q.y; total += q.x + q.y;
        }
        uint32_t incl = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (lane >= (uint32_t)o) incl += t; }
        if (lane == 63u) s_tmp[wave] = incl;
        __syncthreads();
        uint32_t acc = incl - total + 4u * mis;
#pragma unroll
        for (int v = 0; v < WAVES; ++v) acc += ((uint32_t)v < wave) ? s_tmp[v] : 0u;
        uint32_t out[WAVES][2];
#pragma unroll
        for (int p_ = 0; p_ < 2; ++p_)
#pragma unroll
            for (int v = 0; v < WAVES; ++v) { out[v][p_] = acc; acc += c[v][p_]; }
#pragma unroll
        for (int v = 0; v < WAVES; ++v) reinterpret_cast<uint2 *>(s_hist2 + v * kLeanRow)[tid] = make_uint2(out[v][0], out[v][1]);
    }
    __syncthreads();
    MARK(6);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        uint32_t a = (k[i] >> 7) & 0x7FCu;
        if (i >= kFirstMaybeEmpty) {  // an empty slot stays where it is: slot mis + L
            const uint32_t L = seg + i * 64 + lane;
            a = L < n ? a : 2048u + 4u * lane;
            const uint32_t r = rank[i] + *reinterpret_cast<const uint32_t *>(my + a);
            rank[i] = L < n ? r : 4u * (mis + L);
            continue;
        }
        rank[i] += *reinterpret_cast<const uint32_t *>(my + a);
    }
    // (every wave read its pass-2 keys out of s_keys before its atomics, and two barriers lie behind those: s_keys is free)
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(s_keys) + rank[i]) = k[i];
    __syncthreads();
    MARK(7);
    // ---- store: slot q = 4 v + c holds sorted position q - mis
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const uint32_t v = j * THREADS + tid;
        if (j > 0 && j < VEC - 1) {
            reinterpret_cast<uint4 *>(abase)[v] = reinterpret_cast<const uint4 *>(s_keys)[v];
        } else if (v < nvec) {
            const uint4 q4 = reinterpret_cast<const uint4 *>(s_keys)[v];
            const uint32_t q = 4u * v;
            if (q >= mis && q + 4u <= end) {
                reinterpret_cast<uint4 *>(abase)[v] = q4;
            } else {  // the two ends of the bucket: the neighbours' keys in the same 16 bytes are not ours to write
                if (q + 0u - mis < n) abase[q + 0u] = q4.x;
                if (q + 1u - mis < n) abase[q + 1u] = q4.y;
                if (q + 2u - mis < n) abase[q + 2u] = q4.z;
                if (q + 3u - mis < n) abase[q + 3u] = q4.w;
            }
        }
    }
    MARK(8);
    MARK_WAIT_VM();
    MARK(9);
#ifdef LEAN_MARKS
    if (tid == 0) for (int i_ = 0; i_ < 10; ++i_) g_marks[(size_t)blockIdx.x * 12 + i_] = s_marks[i_];
#endif
}

template <int OCC, bool GUARD = false>
__global__ __launch_bounds__(256, OCC) void lean_kernel(uint32_t *__restrict__ keys, const uint32_t *__restrict__ off) {
    constexpr int MAXVEC = 7;
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[256 * 4 * MAXVEC + 4];
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[5 * kLeanRow];
    __shared__ uint32_t s_tmp[16];
    const uint32_t begin = off[blockIdx.x], n = off[blockIdx.x + 1] - begin;
    const uint32_t mis = static_cast<uint32_t>((reinterpret_cast<uintptr_t>(keys + begin) >> 2) & 3u);
    if (n == 0 || mis + n > 256 * 4 * MAXVEC) return;
    uint32_t *abase = keys + begin - mis;
    const uint32_t used = (mis + n + 1023u) / 1024u;
#ifdef LEAN_ONLY
    lean_sort_store<256, LEAN_ONLY, GUARD>(abase, mis, n, s_keys, s_hist, s_tmp);
    return;
#endif
    switch (used) {
        case 1: lean_sort_store<256, 1, GUARD>(abase, mis, n, s_keys, s_hist, s_tmp); break;
        case 2: lean_sort_store<256, 2, GUARD>(abase, mis, n, s_keys, s_hist, s_tmp); break;
        case 3: lean_sort_store<256, 3, GUARD>(abase, mis, n, s_keys, s_hist, s_tmp); break;
        case 4: lean_sort_store<256, 4, GUARD>(abase, mis, n, s_keys, s_hist, s_tmp); break;
        case 5: lean_sort_store<256, 5, GUARD>(abase, mis, n, s_keys, s_hist, s_tmp); break;
        case 6: lean_sort_store<256, 6, GUARD>(abase, mis, n, s_keys, s_hist, s_tmp); break;
        default: lean_sort_store<256, 7, GUARD>(abase, mis, n, s_keys, s_hist, s_tmp); break;
    }
}

// timing floors: load + store only, through the same 16-byte path
__global__ __launch_bounds__(256, 4) void copy_kernel(uint32_t *__restrict__ keys, const uint32_t *__restrict__ off) {
    const uint32_t begin = off[blockIdx.x], n = off[blockIdx.x + 1] - begin;
    const uint32_t mis = static_cast<uint32_t>((reinterpret_cast<uintptr_t>(keys + begin) >> 2) & 3u);
    if (n == 0) return;
    uint32_t *abase = keys + begin - mis;
    const uint32_t nvec = (mis + n + 3u) / 4u;
    VecKeysT<7> k;
    vec_load<256, 7>(k, abase, nvec);
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const uint32_t v = j * 256 + threadIdx.x;
        if (v < nvec && 4u * v >= mis && 4u * v + 4u <= mis + n) reinterpret_cast<uint4 *>(abase)[v] = make_uint4(k.w[4 * j], k.w[4 * j + 1], k.w[4 * j + 2], k.w[4 * j + 3]);
    }
}

struct Input {
    uint32_t n, nb;
    uint32_t *d_part, *d_keys, *d_off;
    std::vector<uint32_t> sorted;  // expected
};

template <typename F>
void run(const char *name, Input &in, hipStream_t st, F launch, bool check = true) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9, sum = 0; const int reps = 6;
    for (int r = 0; r < reps + 1; ++r) {
        CK(hipMemcpyAsync(in.d_keys, in.d_part, (size_t)in.n * 4, hipMemcpyDeviceToDevice, st));  // re-arm; also what precedes the kernel in the sort: a write of its input
        CK(hipEventRecord(a, st));
        launch();
        CK(hipEventRecord(b, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (r) { best = std::min(best, ms); sum += ms; }
    }
    bool ok = true;
    if (check) {
        std::vector<uint32_t> o(in.n); CK(hipMemcpy(o.data(), in.d_keys, (size_t)in.n * 4, hipMemcpyDeviceToHost));
        ok = o == in.sorted;
    }
    printf("%-40s min %.1f us avg %.1f us (%.2f TB/s) %s\n", name, best * 1e3, sum / reps * 1e3, 8.0 * in.n / (best * 1e-3) / 1e12, check ? (ok ? "exact" : "WRONG") : "-");
    fflush(stdout);
}

static Input make_input(const std::vector<uint32_t> &h, const std::vector<uint32_t> &sorted, int msd_bits) {
    const uint32_t n = (uint32_t)h.size();
    const uint32_t nb = 1u << msd_bits;
    std::vector<uint32_t> cnt(nb + 1, 0), part(n);
    for (auto k : h) cnt[(k >> (32 - msd_bits)) + 1]++;
    for (uint32_t b = 0; b < nb; ++b) cnt[b + 1] += cnt[b];
    { std::vector<uint32_t> cur(cnt.begin(), cnt.end() - 1); for (auto k : h) part[cur[k >> (32 - msd_bits)]++] = k; }
    uint32_t maxb = 0; for (uint32_t b = 0; b < nb; ++b) maxb = std::max(maxb, cnt[b + 1] - cnt[b]);
    printf("n=%u buckets=%u max bucket=%u\n", n, nb, maxb);
    Input in; in.n = n; in.nb = nb;
    in.sorted = sorted;
    CK(hipMalloc(&in.d_part, (size_t)n * 4)); CK(hipMalloc(&in.d_keys, (size_t)n * 4)); CK(hipMalloc(&in.d_off, (size_t)(nb + 1) * 4));
    CK(hipMemcpy(in.d_part, part.data(), (size_t)n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(in.d_off, cnt.data(), (size_t)(nb + 1) * 4, hipMemcpyHostToDevice));
    return in;
}

int main(int argc, char **argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atof(argv[1]) : 100000000u;
    std::vector<uint32_t> h(n);
    std::mt19937 gen(1);
    for (auto &x : h) x = gen();
    const char *dist = argc > 3 ? argv[3] : "uniform";
    if (dist[0] == 's') std::sort(h.begin(), h.end());                      // sorted
    if (dist[0] == 'l') for (auto &x : h) x = (x & ~511u) | 7u;              // low 9 bits constant
    if (dist[0] == 'm') for (auto &x : h) x = (x & ~(511u << 9)) | (5u << 9); // middle 9 bits constant
    printf("distribution %s\n", dist);
    std::vector<uint32_t> sorted = h; std::sort(sorted.begin(), sorted.end());
    Input in = make_input(h, sorted, 14);
    Input in16 = make_input(h, sorted, 16);
    const uint32_t nb = in.nb;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned long long *d_marks; CK(hipMalloc(&d_marks, (size_t)nb * 12 * 8)); CK(hipMemset(d_marks, 0, (size_t)nb * 12 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_marks), &d_marks, sizeof(d_marks)));
    const char *only = argc > 2 ? argv[2] : nullptr;  // "lean": just that kernel (for rocprofv3 counter passes)
    if (only) {
        run("lean 256x7 occ4", in, st, [&] { hipLaunchKernelGGL((lean_kernel<4>), dim3(nb), dim3(256), 0, st, in.d_keys, in.d_off); });
        run("lean 256x7 occ4 guard", in, st, [&] { hipLaunchKernelGGL((lean_kernel<4, true>), dim3(nb), dim3(256), 0, st, in.d_keys, in.d_off); });
        run("vec 256x7 occ4", in, st, [&] { hipLaunchKernelGGL((vec_kernel<256, 7, 4>), dim3(nb), dim3(256), 0, st, in.d_keys, in.d_off); });
        run("base", in, st, [&] { hipLaunchKernelGGL(base_kernel, dim3(nb), dim3(256), 0, st, in.d_keys, in.d_off); });
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) {
        run("copy (16-byte load + store)", in, st, [&] { hipLaunchKernelGGL(copy_kernel, dim3(nb), dim3(256), 0, st, in.d_keys, in.d_off); }, false);
        run("base", in, st, [&] { hipLaunchKernelGGL(base_kernel, dim3(nb), dim3(256), 0, st, in.d_keys, in.d_off); });
        run("vec 256x7 occ4", in, st, [&] { hipLaunchKernelGGL((vec_kernel<256, 7, 4>), dim3(nb), dim3(256), 0, st, in.d_keys, in.d_off); });
        run("vec 512x4 3wg/cu", in, st, [&] { hipLaunchKernelGGL((vec_kernel<512, 4, 6>), dim3(nb), dim3(512), 0, st, in.d_keys, in.d_off); });
        run("vec 512x4 2wg/cu", in, st, [&] { hipLaunchKernelGGL((vec_kernel<512, 4, 4>), dim3(nb), dim3(512), 0, st, in.d_keys, in.d_off); });
        run("lean 256x7 occ4", in, st, [&] { hipLaunchKernelGGL((lean_kernel<4>), dim3(nb), dim3(256), 0, st, in.d_keys, in.d_off); });
#ifdef LEAN_MARKS
        {
            std::vector<unsigned long long> m((size_t)nb * 12);
            CK(hipMemcpy(m.data(), d_marks, m.size() * 8, hipMemcpyDeviceToHost));
            double sum[12] = {0}; unsigned long long tmin = ~0ull, tmax = 0;
            for (uint32_t w = 0; w < nb; ++w) {
                for (int i = 1; i < 10; ++i) sum[i] += double(m[(size_t)w * 12 + i] - m[(size_t)w * 12 + i - 1]);
                tmin = std::min(tmin, m[(size_t)w * 12]); tmax = std::max(tmax, m[(size_t)w * 12 + 9]);
            }
            printf("  marks (cycles): span %llu | load %.0f | p1 atomics %.0f | scan1 %.0f | p1 base+write %.0f | p2 read+atomics %.0f | scan2 %.0f | p2 base+write %.0f | store issue %.0f | store drain %.0f | lifetime %.0f\n",
                   tmax - tmin, sum[1] / nb, sum[2] / nb, sum[3] / nb, sum[4] / nb, sum[5] / nb, sum[6] / nb, sum[7] / nb, sum[8] / nb, sum[9] / nb,
                   (sum[1] + sum[2] + sum[3] + sum[4] + sum[5] + sum[6] + sum[7] + sum[8] + sum[9]) / nb);
        }
#endif
        run("lean 256x7 occ5", in, st, [&] { hipLaunchKernelGGL((lean_kernel<5>), dim3(nb), dim3(256), 0, st, in.d_keys, in.d_off); });
        run("wave16 4 waves/wg cap 1792 occ4", in16, st, [&] { hipLaunchKernelGGL((wave_kernel<7, 4, 4>), dim3((in16.nb + 3) / 4), dim3(256), 0, st, in16.d_keys, in16.d_off, in16.nb); });
        run("wave16 4 waves/wg cap 2048 occ4", in16, st, [&] { hipLaunchKernelGGL((wave_kernel<8, 4, 4>), dim3((in16.nb + 3) / 4), dim3(256), 0, st, in16.d_keys, in16.d_off, in16.nb); });
        run("wave16 1 wave/wg cap 1792 occ4", in16, st, [&] { hipLaunchKernelGGL((wave_kernel<7, 1, 4>), dim3(in16.nb), dim3(64), 0, st, in16.d_keys, in16.d_off, in16.nb); });
        run("wave16 2 waves/wg cap 1792 occ5", in16, st, [&] { hipLaunchKernelGGL((wave_kernel<7, 2, 5>), dim3((in16.nb + 1) / 2), dim3(128), 0, st, in16.d_keys, in16.d_off, in16.nb); });
        for (uint32_t grid : {1024u})
            run(("pf 256x7 occ4 grid " + std::to_string(grid)).c_str(), in, st, [&] { hipLaunchKernelGGL((pf_kernel<256, 7, 4>), dim3(grid), dim3(256), 0, st, in.d_keys, in.d_off, nb); });
        for (uint32_t grid : {768u, 1536u})
            run(("pf 256x7 occ3 grid " + std::to_string(grid)).c_str(), in, st, [&] { hipLaunchKernelGGL((pf_kernel<256, 7, 3>), dim3(grid), dim3(256), 0, st, in.d_keys, in.d_off, nb); });
        for (uint32_t grid : {512u, 1024u})
            run(("pf 512x4 2wg/cu grid " + std::to_string(grid)).c_str(), in, st, [&] { hipLaunchKernelGGL((pf_kernel<512, 4, 4>), dim3(grid), dim3(512), 0, st, in.d_keys, in.d_off, nb); });
    }
    return 0;
}
