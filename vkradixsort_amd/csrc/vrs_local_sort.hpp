// vrs_local_sort.hpp -- the LDS-local sort of one bucket of bare uint32 keys by its low 18 bits (two 9-bit passes), shared by
// the counted hybrid form (vrs_msd_hybrid.hip: the bucket lies contiguous in the buffer and is sorted in place) and the pool form
// (vrs_msd_pool.hip: the bucket is gathered from runs and written to its final place).  See vrs_msd_hybrid.hip, "the local sort of
// bare uint32 keys", for how the body is laid out for the LDS pipe.
#pragma once
#include "vrs_device.hpp"

namespace vrs {

constexpr int kLeanRow = 512 + 64;  // words per counter table: 512 digits + one dummy per lane for slots without a key
constexpr int kLeanMaxVec = 7;      // 16-byte vectors per thread: capacity THREADS * 28 slots
// k[4 j + c] = slot q = 4 (j THREADS + tid) + c of the bucket as it was READ (from the first 16-byte boundary at or before its
// first key); slots [mis_in, mis_in + n) hold keys (mis_in < 4).  Sorts them and stores the sorted bucket to abase[mis .. mis + n)
// (16-byte vector stores inside; mis < 4: the misalignment of the place it is WRITTEN to -- the same as mis_in for a bucket sorted in
// place, any other for one that is read from elsewhere: pass 1 leaves the keys in position order, whatever slots they came from).
// the sorted bucket's 16-byte stores.  STREAM: the bucket goes to a buffer nobody reads again soon (the pool form writes the caller's
// buffer, its input came from elsewhere) -- a nontemporal store, which does not take room in the memory-side cache away from the
// second pass's output the other buckets are still to read (pool form, 10^8 keys: 0.546 -> 0.529 ms).  Not for the in-place sorts of
// the counted form: their lines are in the caches already.
template <bool STREAM>
__device__ __forceinline__ void store_sorted(uint4 *p, const uint4 &q) {
    if constexpr (STREAM) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = {q.x, q.y, q.z, q.w};
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
    } else {
        *p = q;
    }
}

// ALIGNED_IN: mis_in == 0 (a bucket read from its own 16-byte aligned region, written elsewhere): the slots without a key are then
// the LAST ones -- which reach up to three slots into the row before the last when the output's misalignment adds a row.
template <int THREADS, int VEC, bool GUARD, bool STREAM = false, bool ALIGNED_IN = false>
__device__ __forceinline__ void lean_sort_body(uint32_t (&k)[4 * VEC], uint32_t *abase, uint32_t mis, uint32_t n, uint32_t *s_keys,
                                               uint32_t *s_hist2, uint32_t *s_tmp, bool guard1, bool guard2, uint32_t mis_in) {
    constexpr int WAVES = THREADS / 64, ITEMS = 4 * VEC, PER = 512 / THREADS;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t end = mis + n;
    const uint32_t nvec = (end + 3u) / 4u;
    uint32_t rank[ITEMS];
    uint32_t *s_hist = s_hist2 + WAVES * kLeanRow;
    // ---- pass 1: low 9 bits, one table, ties in any order; byte address of counter d = 4 d
#pragma unroll
    for (int j = 0; j < VEC; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t a = (k[4 * j + c] << 2) & 0x7FCu;
            if (ALIGNED_IN ? j >= VEC - 2 : (j == 0 || j == VEC - 1)) {
                const uint32_t q = 4u * (j * THREADS + tid) + c;
                a = (q - mis_in < n) ? a : 2048u + 4u * lane;
            }
            uint32_t *counter = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(s_hist) + a);
            if (GUARD && guard1) {  // workgroup-uniform
                const uint32_t a0 = __builtin_amdgcn_readfirstlane(a);
                if (__ballot(a == a0) == ~0ull) {  // one counter for the whole instruction: lane 0 adds the 64 keys
                    uint32_t old = 0;
                    if (lane == 0u) old = __hip_atomic_fetch_add(counter, 256u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    rank[4 * j + c] = __builtin_amdgcn_readfirstlane(old) + 4u * lane;
                    continue;
                }
            }
            rank[4 * j + c] = __hip_atomic_fetch_add(counter, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    __syncthreads();
    {   // exclusive prefix over the 512 bins
        uint32_t c[PER], total = 0;
        if constexpr (PER == 2) {
            const uint2 q = reinterpret_cast<const uint2 *>(s_hist)[tid];
            c[0] = q.x;
            c[1] = q.y;
            total = q.x + q.y;
        } else {
            c[0] = s_hist[tid];
            total = c[0];
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
        if constexpr (PER == 2) reinterpret_cast<uint2 *>(s_hist)[tid] = make_uint2(acc, acc + c[0]);
        else s_hist[tid] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < VEC; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t a = (opaque(k[4 * j + c]) << 2) & 0x7FCu;
            if (ALIGNED_IN ? j >= VEC - 2 : (j == 0 || j == VEC - 1)) {
                // a slot behind the bucket keeps its place (position q: mis_in + n slots lie before the first of them, mis_in of
                // those without a key), the ones before the bucket follow the keys (position n + q)
                const uint32_t q = 4u * (j * THREADS + tid) + c;
                const bool valid = q - mis_in < n;
                a = valid ? a : 2048u + 4u * lane;
                const uint32_t r = rank[4 * j + c] + *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(s_hist) + a);
                rank[4 * j + c] = valid ? r : 4u * (q < mis_in ? n + q : q);
                continue;
            }
            rank[4 * j + c] += *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(s_hist) + a);
        }
    // byte offset Lb = 4 L of position L goes to byte (Lb & ~1023) | ((Lb & 252) << 2) | ((Lb >> 6) & 12)
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t Lb = rank[i];
        const uint32_t ph = (Lb & ~1023u) | ((Lb & 252u) << 2) | ((Lb >> 6) & 12u);
        *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(s_keys) + ph) = k[i];
    }
    __syncthreads();
    // ---- pass 2: high 9 bits, one table per wave, stable
    const uint32_t seg = wave * (ITEMS * 64);
#pragma unroll
    for (int g = 0; g < VEC; ++g) {
        const uint4 t = reinterpret_cast<const uint4 *>(s_keys + seg + g * 256)[lane];
        k[4 * g] = t.x;
        k[4 * g + 1] = t.y;
        k[4 * g + 2] = t.z;
        k[4 * g + 3] = t.w;
    }
    char *my = reinterpret_cast<char *>(s_hist2 + wave * kLeanRow);
    // fewer than a row (4 THREADS slots) + 3 slots hold no key, all at the end of the position space: with 256 threads and five
    // rows or more that is the last 17 items of the last wave, otherwise it may be any item of any wave
    constexpr int kEmptyItems = (4 * THREADS + 3 + 63) / 64;
    constexpr int kFirstMaybeEmpty = ITEMS >= kEmptyItems ? ITEMS - kEmptyItems : 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        uint32_t a = (k[i] >> 7) & 0x7FCu;
        if (i >= kFirstMaybeEmpty) a = (seg + i * 64 + lane < n) ? a : 2048u + 4u * lane;
        uint32_t *counter = reinterpret_cast<uint32_t *>(my + a);
        if (GUARD && guard2) {
            const uint32_t a0 = __builtin_amdgcn_readfirstlane(a);
            if (__ballot(a == a0) == ~0ull) {
                uint32_t old = 0;
                if (lane == 0u) old = __hip_atomic_fetch_add(counter, 256u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                rank[i] = __builtin_amdgcn_readfirstlane(old) + 4u * lane;
                continue;
            }
        }
        rank[i] = __hip_atomic_fetch_add(counter, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    {   // exclusive prefix over (bin, wave); starts at the bucket's misalignment: pass 2 writes slot = mis + position
        uint32_t c[WAVES][PER], total = 0;
#pragma unroll
        for (int v = 0; v < WAVES; ++v) {
            if constexpr (PER == 2) {
                const uint2 q = reinterpret_cast<const uint2 *>(s_hist2 + v * kLeanRow)[tid];
                c[v][0] = q.x;
                c[v][1] = q.y;
                total += q.x + q.y;
            } else {
                c[v][0] = s_hist2[v * kLeanRow + tid];
                total += c[v][0];
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
        uint32_t acc = incl - total + 4u * mis;
#pragma unroll
        for (int v = 0; v < WAVES; ++v) acc += (static_cast<uint32_t>(v) < wave) ? s_tmp[v] : 0u;
        uint32_t out[WAVES][PER];
#pragma unroll
        for (int p_ = 0; p_ < PER; ++p_)
#pragma unroll
            for (int v = 0; v < WAVES; ++v) {
                out[v][p_] = acc;
                acc += c[v][p_];
            }
#pragma unroll
        for (int v = 0; v < WAVES; ++v) {
            if constexpr (PER == 2) reinterpret_cast<uint2 *>(s_hist2 + v * kLeanRow)[tid] = make_uint2(out[v][0], out[v][1]);
            else s_hist2[v * kLeanRow + tid] = out[v][0];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        uint32_t a = (opaque(k[i]) >> 7) & 0x7FCu;
        if (i >= kFirstMaybeEmpty) {  // a slot without a key stays where it is: slot mis + L
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
    // ---- store: slot q = 4 v + c holds sorted position q - mis
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const uint32_t v = j * THREADS + tid;
        if (j > 0 && j < VEC - 1) {
            store_sorted<STREAM>(reinterpret_cast<uint4 *>(abase) + v, reinterpret_cast<const uint4 *>(s_keys)[v]);
        } else if (v < nvec) {
            const uint4 q4 = reinterpret_cast<const uint4 *>(s_keys)[v];
            const uint32_t q = 4u * v;
            if (q >= mis && q + 4u <= end) {
                store_sorted<STREAM>(reinterpret_cast<uint4 *>(abase) + v, q4);
            } else {  // the two ends of the bucket: the neighbours' keys in the same 16 bytes are not ours to write
                if (q + 0u - mis < n) abase[q + 0u] = q4.x;
                if (q + 1u - mis < n) abase[q + 1u] = q4.y;
                if (q + 2u - mis < n) abase[q + 2u] = q4.z;
                if (q + 3u - mis < n) abase[q + 3u] = q4.w;
            }
        }
    }
}

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
// One WAVE sorts a bucket of up to 64 * 28 slots (no workgroup barrier: the LDS executes one wave's operations in order): the same two
// 9-bit passes, slots and dummy counters as lean_sort_body, one 512-counter table reused by both passes.  mis_in / ALIGNED_IN / STREAM:
// see lean_sort_body.
template <int VEC, bool GUARD, bool ALIGNED_IN = false, bool STREAM = false>
__device__ __forceinline__ void wave_sort_body(uint32_t (&k)[4 * VEC], uint32_t *abase, uint32_t mis, uint32_t n, uint32_t *s_keys,
                                               uint32_t *tbl, bool guard1, bool guard2, uint32_t mis_in) {
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
        if (ALIGNED_IN ? i >= ITEMS - 8 : (i < 4 || i >= ITEMS - 4)) {
            const uint32_t q = 4u * ((i >> 2) * 64 + lane) + (i & 3);
            a = (q - mis_in < n) ? a : 2048u + 4u * lane;
        }
        rank[i] = ranked_add(a, guard1);
    }
    wave_phase();
    wave_scan512(tbl, lane, 0u);
    wave_phase();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        uint32_t a = (opaque(k[i]) << 2) & 0x7FCu;
        if (ALIGNED_IN ? i >= ITEMS - 8 : (i < 4 || i >= ITEMS - 4)) {
            const uint32_t q = 4u * ((i >> 2) * 64 + lane) + (i & 3);
            const bool valid = q - mis_in < n;
            a = valid ? a : 2048u + 4u * lane;
            const uint32_t r = rank[i] + *reinterpret_cast<const uint32_t *>(tb + a);
            rank[i] = valid ? r : 4u * (q < mis_in ? n + q : q);
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
                store_sorted<STREAM>(reinterpret_cast<uint4 *>(abase) + v, q4);
            } else {
                if (q + 0u - mis < n) abase[q + 0u] = q4.x;
                if (q + 1u - mis < n) abase[q + 1u] = q4.y;
                if (q + 2u - mis < n) abase[q + 2u] = q4.z;
                if (q + 3u - mis < n) abase[q + 3u] = q4.w;
            }
        }
    }
}

// One LSD pass over the keys a workgroup holds in registers (wave-striped: wave v owns ITEMS * 64 consecutive positions,
// item i of lane l is position v * ITEMS * 64 + i * 64 + l; positions >= n hold nothing and stay where they are), through
// LDS: counters fed by returning LDS atomics, a scan over the bins, re-bucketing, striped read-back.
// STABLE: one counter table per wave (lane order inside an instruction is the RANK_ATOMIC property, item order and wave order
// come from the tables' prefix) -- equal digits keep their order.  Not STABLE: ONE table for the workgroup, a quarter of the
// zeroing and scanning; equal digits come out in any order -- enough for the FIRST pass over bare keys (keys that tie in
// this digit are told apart by the later pass or are equal), never for payloads.
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
// scanned(): called once the pass has scanned its counters, in front of the keys' scatter through LDS -- the scan's registers are free again and a
// scatter, a barrier and a read-back lie ahead: where a caller starts loads it wants in flight behind the pass.
template <int THREADS, int ITEMS, int BITS, bool PAIRS, bool STABLE, typename K = uint32_t, typename Hook = NoHook>
__device__ __forceinline__ void local_pass(K (&key)[ITEMS], uint32_t (&val)[PAIRS ? ITEMS : 1], K *s_keys,
                                           uint32_t *s_vals, uint32_t *s_hist, uint32_t *s_tmp, uint32_t shift, uint32_t n, Hook scanned = Hook()) {
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
    scanned();
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

// the bucket with ITEMS keys per thread (n <= ITEMS * THREADS): read once, two stable 9-bit passes, written back -- in place
// (the counted form) or to another place (the pool form's pairs: from the slack buffers to the caller's)
template <int THREADS, int ITEMS, bool PAIRS, bool STREAM = false>  // STREAM: nobody reads the output soon -- its stores go around the caches
__device__ __forceinline__ void local_sort_bucket_to(const uint32_t *src, const uint32_t *src_vals, uint32_t *bucket, uint32_t *bucket_vals, uint32_t n,
                                                     uint32_t *s_keys, uint32_t *s_vals, uint32_t *s_hist, uint32_t *s_tmp) {
    constexpr int BITS = 9;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t key[ITEMS], val[PAIRS ? ITEMS : 1];
    const uint32_t seg = wave * (ITEMS * 64) + lane;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t idx = seg + i * 64;
        const uint32_t k = src[idx < n ? idx : n - 1u];
        key[i] = k;  // positions >= n hold nothing: the passes leave them alone and they are not written
    }
    if constexpr (PAIRS) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t idx = seg + i * 64;
            val[i] = src_vals[idx < n ? idx : n - 1u];
        }
    }
    local_pass<THREADS, ITEMS, BITS, PAIRS, PAIRS>(key, val, s_keys, s_vals, s_hist, s_tmp, 0, n);  // bare keys: any order of ties
    local_pass<THREADS, ITEMS, BITS, PAIRS, true>(key, val, s_keys, s_vals, s_hist, s_tmp, BITS, n);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t idx = seg + i * 64;
        if (idx < n) {
            if constexpr (STREAM) __builtin_nontemporal_store(key[i], bucket + idx);
            else bucket[idx] = key[i];
        }
    }
    if constexpr (PAIRS) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t idx = seg + i * 64;
            if (idx < n) {
                if constexpr (STREAM) __builtin_nontemporal_store(val[i], bucket_vals + idx);
                else bucket_vals[idx] = val[i];
            }
        }
    }
}

// The pool form's bucket of key + payload PAIRS, sorted as ONE word per pair: inside a bucket every key has the same bits from `lowbits` up
// (the bucket's index; lowbits <= 18), and a pair's place in the bucket as it was read fits IDXB bits -- word = (the key's low bits) << IDXB | place.
// Two STABLE 9-bit passes over the words' key bits leave equal keys in the order of their places, which is the order they came in.  The payloads
// take no part in the passes: they are read into registers while the second pass runs, put into the SAME LDS array the words were sorted in once
// that pass has read it back (payload of place p at slot p), and fetched by the place each sorted word carries.  One array instead of two: 43 KB
// of LDS per workgroup instead of 70, three workgroups per CU instead of two.
template <int THREADS, int ITEMS, int IDXB, bool STREAM>
__device__ __forceinline__ void local_sort_packed_pairs_to(const uint32_t *src, const uint32_t *src_vals, uint32_t *bucket, uint32_t *bucket_vals, uint32_t n,
                                                           uint32_t lowbits, uint32_t *s_keys, uint32_t *s_hist, uint32_t *s_tmp) {
    static_assert(THREADS * ITEMS <= (1 << IDXB) && IDXB + 18 <= 32, "a pair's place and 18 key bits share a word");
    constexpr int BITS = 9;
    constexpr uint32_t PLACE = (1u << IDXB) - 1u;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t seg = wave * (ITEMS * 64) + lane;
    const uint32_t lowmask = (1u << lowbits) - 1u;
    uint32_t word[ITEMS], val[ITEMS], none[1], high = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t idx = seg + i * 64;
        const uint32_t k = src[idx < n ? idx : n - 1u];
        word[i] = ((k & lowmask) << IDXB) | idx;  // positions >= n hold nothing: the passes leave them alone and they are not written
        if (i == 0) high = k & ~lowmask;          // (the same in every key of the bucket)
    }
    local_pass<THREADS, ITEMS, BITS, false, true>(word, none, s_keys, nullptr, s_hist, s_tmp, IDXB, n);
    // (every place of the workgroup's THREADS * ITEMS is read and staged, the ones behind the bucket with a payload nobody asks for: under
    //  `if (idx < n)` the compiler sinks each load into its branch -- thirteen memory round trips one after the other)
    const auto load_payloads = [&] {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t idx = seg + i * 64;
            val[i] = src_vals[idx < n ? idx : n - 1u];
        }
    };
    // (the payloads' loads start behind the second pass's scan: thirteen more registers through the whole pass would cost the third workgroup)
    local_pass<THREADS, ITEMS, BITS, false, true, uint32_t>(word, none, s_keys, nullptr, s_hist, s_tmp, IDXB + BITS, n, load_payloads);  // (ends behind a barrier: s_keys is free)
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) s_keys[seg + i * 64] = val[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) val[i] = s_keys[seg + i * 64 < n ? word[i] & PLACE : 0u];  // (a position behind the bucket holds no word: nothing to fetch)
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t idx = seg + i * 64;
        if (idx < n) {
            const uint32_t k = high | (word[i] >> IDXB);
            if constexpr (STREAM) __builtin_nontemporal_store(k, bucket + idx);
            else bucket[idx] = k;
        }
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t idx = seg + i * 64;
        if (idx < n) {
            if constexpr (STREAM) __builtin_nontemporal_store(val[i], bucket_vals + idx);
            else bucket_vals[idx] = val[i];
        }
    }
}

template <int THREADS, int ITEMS, bool PAIRS>
__device__ __forceinline__ void local_sort_bucket(uint32_t *bucket, uint32_t *bucket_vals, uint32_t n, uint32_t *s_keys,
                                                  uint32_t *s_vals, uint32_t *s_hist, uint32_t *s_tmp) {
    local_sort_bucket_to<THREADS, ITEMS, PAIRS>(bucket, bucket_vals, bucket, bucket_vals, n, s_keys, s_vals, s_hist, s_tmp);
}

// The local sort is the last kernel of a hybrid sort and LDS-bound: it has HBM time to spare, so it also clears the look-back
// status words for the NEXT sort (the counting read, which is HBM-bound, then skips its 15.6 MB of zero stores): workgroup b
// of `blocks` clears the b-th share of status[0, vecs).
struct StatusClear {
    uint4 *status;   // nullptr: nothing to clear
    uint32_t vecs;
};
__device__ __forceinline__ void clear_status_share(const StatusClear &sc, uint32_t threads) {
    if (sc.status == nullptr) return;
    const uint32_t per = (sc.vecs + gridDim.x - 1u) / gridDim.x;
    const uint32_t z0 = blockIdx.x * per, z1 = min(z0 + per, sc.vecs);
    for (uint32_t c = z0 + threadIdx.x; c < z1; c += threads) sc.status[c] = make_uint4(0, 0, 0, 0);
}

}  // namespace vrs
