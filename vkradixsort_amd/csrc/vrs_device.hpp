// vrs_device.hpp -- device-side building blocks shared by the kernel families of libvkradixsort_amd (gfx950, wave64):
// lane / digit helpers, the block scans, the stable scatter of one chunk through LDS (scatter_chunk) with its offset
// sources (contract offsets, decoupled look-back, reservation).  Included by vrs_contract.hip (contract stages K1-K4), vrs_one_call.hip / vrs_msd_hybrid.hip (the
// one-call sort K5 and its hybrid form K5b) and vrs_msd_pool.hip (the hybrid form without a counting read).
#pragma once
#include "vrs_kernels.h"

#include <hip/hip_ext.h>

#include <type_traits>

// Launch with optional timing events bound to the dispatch packet itself (hipExtLaunchKernel): unlike
// hipEventRecord brackets this adds no barrier packets between dependent kernels.
#define VRS_LAUNCH(kernel, grid, block, stream, ev, ...)                                                        \
    do {                                                                                                        \
        if ((ev).start != nullptr || (ev).stop != nullptr)                                                      \
            hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, (ev).start, (ev).stop, 0, __VA_ARGS__);       \
        else                                                                                                    \
            hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                                    \
    } while (0)

// Phase-timing hooks for tools/lab (compiled out of the product library).
#ifndef VRS_MARK
#define VRS_MARK(i)
#define VRS_MARK_FLUSH()
#endif
namespace vrs {

// a pass whose input is at least this large reads it with nontemporal loads (scatter_chunk, stream_in): half the memory-side cache
constexpr size_t kStreamInBytes = size_t(128) << 20;

// tile shapes of the one-call sort (measured choices: profiles/labs; lab builds that varied them are history)
constexpr int kDtUnroll = 8;  // 16-byte loads in flight per lane in the counting read
constexpr int kLbItems = 16;  // keys per thread of a look-back tile of uint32 keys: 8192-key tiles (12 / 20 / 24 measured the same or worse)
constexpr int kLbBatch = 4;
constexpr int kLbWaves = 8;   // waves of a look-back workgroup of the LSD passes

constexpr int kBins = 256;     // RADIX_SORT_BINS
constexpr int kThreads = 256;  // 4 wave64 per workgroup
constexpr int kWaves = kThreads / 64;

__device__ __forceinline__ uint32_t lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// the value unchanged, but opaque to the optimiser: makes it RECOMPUTE what depends on it instead of keeping it alive
__device__ __forceinline__ uint32_t opaque(uint32_t x) {
    asm volatile("" : "+v"(x));
    return x;
}

// number of set bits of `mask` strictly below the calling lane
__device__ __forceinline__ uint32_t count_below(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

// Keys are uint32 (the reference's SORT_32BIT, four passes) or uint64 (its SORT_64_BIT stub,
// MultiRadixSort.h:10-18 / MultiRadixSort.cpp:51-55: eight passes); every kernel is a template on the key type.
__device__ __forceinline__ uint32_t digit_of(uint32_t key, uint32_t shift) {
    return (key >> shift) & (kBins - 1);
}
__device__ __forceinline__ uint32_t digit_of(uint64_t key, uint32_t shift) {
    return static_cast<uint32_t>(key >> shift) & (kBins - 1);
}

template <typename K>
struct KeyVec;  // 16-byte vector of keys for the histogram's coalesced loads
template <>
struct KeyVec<uint32_t> {
    using type = uint4;
    static constexpr int kKeys = 4;
    static __device__ __forceinline__ uint32_t get(const uint4 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
};
template <>
struct KeyVec<uint64_t> {
    using type = ulonglong2;
    static constexpr int kKeys = 2;
    static __device__ __forceinline__ uint64_t get(const ulonglong2 &v, int i) { return i == 0 ? v.x : v.y; }
};

// a 16-byte vector of keys read ONCE by a kernel that only reads (histogram stage, counting read): a nontemporal load, which does not
// push the previous kernel's dirty lines out of the caches in front of it -- the histogram stage of a 10^8-key pass runs in 70 instead
// of 88 us, what it takes with nothing before it
template <typename Vec>
__device__ __forceinline__ Vec load_stream16(const Vec *p) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    static_assert(sizeof(Vec) == 16, "16-byte vectors");
    const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    Vec r;
    __builtin_memcpy(&r, &t, 16);
    return r;
}

// What a pass buckets by.  RadixDigit: the 8-bit digit at `shift` (the reference's passes).  SplitDigit: the
// index of the key range a key falls in, given up to 255 ascending splitters staged in LDS -- the multi-GPU
// range partition for keys whose top byte is too skewed to cut at byte boundaries (build extension).
template <typename K>
struct RadixDigit {
    uint32_t shift;
    K base = 0;  // the first MSD pass of a sort whose keys are known to start at `base` (vrs_sort_keys_u32_ranged): digit of key - base
    __device__ __forceinline__ uint32_t operator()(K key) const { return digit_of(static_cast<K>(key - base), shift); }
    // the key a ragged tile is padded with: it must carry the largest digit under every shift (it then ranks behind every real key)
    template <typename KK>
    __device__ __forceinline__ KK pad() const { return static_cast<KK>(base - static_cast<K>(1)); }
};
template <typename K>
struct SplitDigit {
    const K *splitters;  // LDS, ascending, 256 slots (slots >= count are never counted)
    uint32_t count;
    // number of splitters <= key, in eight fixed halving steps (branch-free upper bound over 255 slots)
    __device__ __forceinline__ uint32_t operator()(K key) const {
        uint32_t pos = 0;
#pragma unroll
        for (uint32_t step = 128; step >= 1; step >>= 1) {
            const uint32_t probe = pos + step - 1;
            pos += (probe < count && splitters[probe] <= key) ? step : 0u;
        }
        return pos;
    }
    template <typename KK>
    __device__ __forceinline__ KK pad() const { return static_cast<KK>(~static_cast<KK>(0)); }
};
// stage `count` splitters into LDS (all threads of the workgroup call this; ends with a barrier)
template <typename K>
__device__ __forceinline__ void stage_splitters(K *s_split, const K *splitters, uint32_t count) {
    for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) s_split[i] = i < count ? splitters[i] : static_cast<K>(~static_cast<K>(0));
    __syncthreads();
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
// K3 building block: stable scatter of one chunk of <= ITEMS*WAVES*64 keys by a workgroup of WAVES
// wave64.
//
// Layout in the chunk ("wave-striped"): wave v owns the contiguous segment
// [v*ITEMS*64, (v+1)*ITEMS*64); its item i, lane l is key index v*ITEMS*64 + i*64 + l, so every
// load instruction of a wave covers 256 contiguous bytes and (wave, item, lane) order == input
// order, which is what stability needs.
//
// Ranking (per wave, per item), RANK_BALLOT: eight __ballot votes -- one per digit bit -- give each
// lane the 64-bit mask of lanes holding the same digit ("match-any").  rank-in-wave = per-wave LDS
// counter of that digit + number of matching lanes below me (mbcnt); the highest matching lane bumps
// the counter by __popcll(mask).  LDS operations of one wave execute in order, so item i+1 sees
// item i's update without a barrier.
// RANK_ATOMIC: one returning LDS atomic add per key on the per-wave counter.  Correct only if the
// LDS serialises same-address lanes of one instruction in ascending lane order (observed on gfx950,
// not architecturally promised): selected only after vrs_debug_atomic_rank_selftest passes.
constexpr int RANK_BALLOT = 0;
constexpr int RANK_ATOMIC = 1;

template <typename K, int ITEMS, int WAVES, bool PAIRS = false>
struct ChunkSmem {
    K keys[ITEMS * WAVES * 64];  // re-bucketed keys, chunk order by digit
    uint32_t vals[PAIRS ? ITEMS * WAVES * 64 : 1];  // re-bucketed payloads (pairs only)
    uint32_t whist[WAVES][kBins];       // per-wave digit counters -> per-wave digit start positions
    uint32_t gbase[kBins];              // global offset of digit d minus its start inside the chunk
    uint32_t scan_tmp[WAVES];
    uint32_t lb_gave_up;                // look-back only: some digit's wait ran out of budget
};

// 64-bit mask of the lanes whose 8-bit digit equals mine ("match-any"), 4 VALU per digit bit:
//   m  = -bit            v_bfe_i32   (all ones in lanes whose bit is set)
//   B  = ballot(bit)     v_cmp_ne_u32 -> SGPR pair
//   peers &= ~(B ^ m)    v_bitop3_b32 (gfx950 three-input boolean, truth table 0x84 = b & ~(a ^ c)),
//                        once per 32-lane half: lanes that differ from me in this bit drop out.
__device__ __forceinline__ uint64_t match_any_digit(uint32_t d) {
    uint32_t lo = ~0u, hi = ~0u;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const uint32_t m = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(d), b, 1));
        const uint64_t B = __builtin_amdgcn_uicmp(m, 0u, 33 /* ICMP_NE */);
        lo = __builtin_amdgcn_bitop3_b32(m, lo, static_cast<uint32_t>(B), 0x84);
        hi = __builtin_amdgcn_bitop3_b32(m, hi, static_cast<uint32_t>(B >> 32), 0x84);
    }
    return (static_cast<uint64_t>(hi) << 32) | lo;
}

template <int WAVES>
__device__ __forceinline__ uint32_t block_exclusive_scan_w(uint32_t v, uint32_t *s_tmp, uint32_t lane,
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
    for (int j = 0; j < WAVES; ++j) base += (static_cast<uint32_t>(j) < wave) ? s_tmp[j] : 0u;
    return base + incl - v;
}

// ---------------------------------------------------------------------------------------------
// Decoupled look-back along a STREAM of tiles (the one-call sort, see "K5" below).  One 32-bit status word per
// (tile, digit):  bits 31:29 = tag of the pass that wrote it (pass + 1; 0 = never written: the counting read zeroes
// the region once per group of four passes, and a word left by an earlier pass simply reads as "not published"),
// bit 28 = 0 the tile's own count / 1 the inclusive count of the stream up to and including the tile,
// bits 27:0 the count (a stream is shorter than 2^28 keys: N < 2^30 and no stream is longer than 1.25 N / 8 + a tile).
// The word carries its own flag, so no fence is needed.
//
// All tiles of a stream are meant to run behind ONE XCD's L2 (block b -> XCD b % 8: observed, probed at context
// creation, not promised by HIP), so the words are published with L2-resident stores and polled with loads that
// bypass only the CU's L1: a hand-off costs an L2 round trip instead of a trip through the fabric (measured: 181
// vs 223 us per pass; writing every word through as well costs 30 us per pass).  That store is a workgroup-scope
// atomic store (global_store sc0: the line stays dirty in this XCD's L2) read by ANOTHER workgroup with an agent-scope
// load (global_load sc1: bypasses the reader's L1, served by the same L2) -- outside what the HSA memory model
// promises for inter-workgroup data, correct on gfx950 because the vector L1 is write-through and both workgroups sit
// behind the one L2 that holds the line (MI355X_MICROARCH.md, "stores of each flavour").  Placement is therefore never
// trusted: every workgroup compares HW_REG_XCC_ID with its stream's XCD, and one that finds itself behind another L2
// ("foreign") neither reads status words (it re-counts its stream's earlier tiles from the keys) nor publishes
// L2-resident ones (it stores write-through, sc1, which the agent-scope polls of the others do see).
// Progress never depends on another workgroup either: a tile polls an unpublished row at most `budget` times, then
// stops waiting and counts the digits of its stream's earlier keys itself (same result; the guide's "bound every spin").
constexpr uint32_t kLbInclusive = 1u << 28, kLbValue = (1u << 28) - 1u, kLbTagShift = 29;

__device__ __forceinline__ uint32_t lb_load(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_load sc1: L1 bypassed, L2 served
}
__device__ __forceinline__ void lb_store_through(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_store sc1: written through
}
__device__ __forceinline__ void lb_store_l2(uint32_t *p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // stays (dirty) in this XCD's L2
}

struct NoLookback {
    static constexpr bool kEnabled = false;
    static constexpr bool kReserves = false;
    static constexpr bool kPool = false;
};
// where a chunk's keys come from when they do not lie in one piece (scatter_chunk's SRC): lane p of every wave holds piece p --
// chunk positions [lo, lo + len) (lo relative to the chunk, as a wrapped unsigned) are the virtual slots slot, slot + 1, ...;
// virtual slots below n_virt lie in `regions`, the others in `overflow`.  Pieces [p0, p1) touch the chunk.
struct NoPieces {
    static constexpr bool kEnabled = false;
};
struct PieceSrc {
    static constexpr bool kEnabled = true;
    uint32_t lo, len, slot;       // per lane
    uint32_t p0, p1, first_slot;  // wave-uniform
    const uint32_t *regions, *overflow;
    uint32_t n_virt;
    const uint32_t *vregions = nullptr, *voverflow = nullptr;  // the payloads' twins of the two buffers (pairs only)
    using gptr = const uint32_t __attribute__((address_space(1))) *;  // (say that these are GLOBAL pointers, or the loads become flat loads)
    __device__ __forceinline__ gptr at(uint32_t v) const { return v < n_virt ? (gptr)regions + v : (gptr)overflow + (v - n_virt); }
    __device__ __forceinline__ gptr val_at(uint32_t v) const { return v < n_virt ? (gptr)vregions + v : (gptr)voverflow + (v - n_virt); }
};
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3u << 11) | 20u); }  // HW_REG_XCC_ID[3:0]
// byte x of the probed map = the XCC the blocks with blockIdx % 8 == x ran on
__device__ __forceinline__ uint32_t xcc_of(unsigned long long xcc_map, uint32_t x) { return static_cast<uint32_t>((xcc_map >> (8u * x)) & 0xFFu); }
// Placement drift: the dispatcher deals a launch's blocks out to the XCCs round-robin from an XCC of the hardware queue's own, and a
// stream may move to another queue after the context probed -- the kernels that lean on the probed order then run their
// placement-independent routes (exact, slower) until someone probes again.  The first blocks of such a kernel say so in a word of
// pinned host memory (nothing on the usual path: one compare per workgroup); the host probes again before its next sort.
__device__ __forceinline__ void report_drift(uint32_t *drift, unsigned long long xcc_map) {
    if (drift != nullptr && threadIdx.x == 0 && blockIdx.x < 64u && xcc_id() != xcc_of(xcc_map, blockIdx.x & 7u))
        __hip_atomic_fetch_add(drift, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct StreamLookback {
    static constexpr bool kEnabled = true;
    bool foreign = false;     // this workgroup is not behind its stream's L2 (workgroup-uniform)
    bool hold = false;        // test hook: this tile never publishes (its successors must stop waiting)
    uint32_t recounted = 0;   // foreign only: exclusive count of my digit over the stream's earlier tiles
    uint32_t *col = nullptr;  // status word of (tile 0 of my stream, digit == my thread)
    size_t stride = 0;        // words between consecutive tiles of the stream
    int index = 0;            // this tile's position in its stream
    uint32_t seed = 0;        // global offset of my digit at the start of the stream
    uint32_t tag = 0;         // (pass + 1) << kLbTagShift
    uint32_t budget = 0;      // polls of an unpublished row before giving up
    const void *stream_keys = nullptr;  // first key of the stream in the pass's input
    uint32_t done = 0;        // keys of the stream before this tile

    static constexpr bool kReserves = false;
    static constexpr bool kPool = false;
    __device__ __forceinline__ void publish(uint32_t v) const {
        if (hold) return;
        uint32_t *p = col + static_cast<size_t>(index) * stride;
        if (foreign) lb_store_through(p, tag | v); else lb_store_l2(p, tag | v);
    }
    // rows first, first-1, ...: the row before the stream's first tile reads as "inclusive, 0"
    __device__ __forceinline__ void fetch(int first, uint32_t (&v)[kLbBatch]) const {
#pragma unroll
        for (int r = 0; r < kLbBatch; ++r)
            v[r] = first - r >= 0 ? lb_load(col + static_cast<size_t>(first - r) * stride) : (tag | kLbInclusive);
    }
    // exclusive count of my digit over the tiles before mine; v = fetch(index - 1) issued earlier.
    // Every round trip consumes all rows that are published; at the first unpublished one the REST of the batch is
    // fetched again in one go (re-polling row by row would serialise one round trip per row).  gave_up: the budget
    // ran out on an unpublished row (the caller then counts the stream's earlier keys itself).
    __device__ __forceinline__ uint32_t resolve(uint32_t (&v)[kLbBatch], bool &gave_up) const {
        uint32_t acc = 0, polls = 0;
        [[maybe_unused]] uint32_t trips = 1;
        int first = index - 1;
        for (;;) {
            bool done_ = false, blocked = false;
            int consumed = 0;
#pragma unroll
            for (int r = 0; r < kLbBatch; ++r) {
                if (!done_ && !blocked) {
                    const uint32_t x = v[r];
                    if ((x >> kLbTagShift) != (tag >> kLbTagShift)) {
                        blocked = true;  // not published in this pass (yet)
                    } else {
                        acc += x & kLbValue;
                        consumed = r + 1;
                        done_ = (x & kLbInclusive) != 0u;
                    }
                }
            }
            if (done_) {
                return acc;
            }
            first -= consumed;
            ++trips;
            if (blocked) {
                if (++polls > budget) {
                    gave_up = true;
                    return 0u;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            fetch(first, v);
        }
    }
};

// Reservation instead of look-back (MSD passes over BARE keys; MsdPlan::cursor_a / cursor_b): my digit's range hands out places
// in the order tiles ask.  ONE atomic add per tile and digit in the L2 all tiles of the stream run behind -- no status rows to
// publish, poll and clear: 160 -> 143 us for the first MSD pass of 10^8 keys (profiles/labs/r03_reservation.txt).  The same
// interface as StreamLookback, so that scatter_chunk does not care; a type of its own, so that the look-back passes (LSD sorts,
// payloads), which are bound by their latency at small sizes, carry none of it.
struct StreamReserve {
    static constexpr bool kEnabled = true;
    static constexpr bool kReserves = true;
    static constexpr bool kPool = false;
    bool foreign = false;            // this workgroup is not behind its stream's L2: it takes room from the range's END (device scope)
    uint32_t recounted = 0;          // (unused: interface of StreamLookback)
    int index = 0;
    const void *stream_keys = nullptr;
    uint32_t done = 0;
    uint32_t seed = 0;               // where my digit's range starts
    uint32_t *cursor = nullptr;      // keys of the range placed so far (L2-local atomics)
    uint32_t *back = nullptr;        // keys taken from its end by foreign tiles (device-scope atomics)
    uint32_t region_len = 0;         // foreign only: keys the range holds
    uint32_t pad_keys = 0;           // padding keys of a ragged tile counted under my digit: they take no room
    mutable uint32_t reserved = 0;
    mutable bool reserved_yet = false;

    __device__ __forceinline__ void publish(uint32_t v) const {
        if (reserved_yet) return;  // the second call (the inclusive prefix) has nobody to tell
        reserved_yet = true;
        const uint32_t cnt = v - pad_keys;
        if (cnt) {
            if (foreign) {
                // (the range holds exactly all its keys; should a plan ever say otherwise, stay inside it rather than wrap around)
                const uint32_t taken = cnt + __hip_atomic_fetch_add(back, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                reserved = region_len >= taken ? region_len - taken : 0u;
            } else
                reserved = __hip_atomic_fetch_add(cursor, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __device__ __forceinline__ void fetch(int, uint32_t (&)[kLbBatch]) const {}
    // keys of my digit's range in front of this tile's: what the look-back would have answered for a stable pass
    __device__ __forceinline__ uint32_t resolve(uint32_t (&)[kLbBatch], bool &) const { return reserved; }
};

// cnt[0..256) += digit counts of keys[0, count) (all threads of the workgroup; barriers are the caller's)
template <typename K, typename DG>
__device__ __forceinline__ void recount_keys(uint32_t *cnt, const K *keys, uint32_t count, const DG &dg) {
    for (uint32_t j = threadIdx.x; j < count; j += blockDim.x) atomicAdd(&cnt[dg(keys[j])], 1u);
}

// `run_off`: thread t (< 256) holds the running global offset of digit t, advanced by this chunk's
// count of t.  `valid`: number of real keys in the chunk (the rest is padding that sorts last).
// FULL: valid == ITEMS*WAVES*64 is known, so no load or store is predicated.
//
// Every phase is written as "issue all ITEMS independent LDS/global operations, then consume":
// a workgroup is latency-bound (one pass over its keys, few waves), so dependent
// read -> wait -> write chains per item are what must not appear in the ISA.
// LB: StreamLookback obtains the digit offsets by decoupled look-back instead of from `run_off`.
// SRC (PieceSrc): the chunk's keys lie in several PIECES of two buffers -- the second MSD pass of the pool form reads a top byte's
// keys from the eight slices' primary and overflow regions, and a tile may straddle their ends.
template <typename K, int ITEMS, int WAVES, bool PAIRS, int RANK, bool FULL, typename DG, typename LB = NoLookback, typename SRC = NoPieces>
__device__ __forceinline__ void scatter_chunk(ChunkSmem<K, ITEMS, WAVES, PAIRS> &sm, const K *kin,
                                              const uint32_t *vin, K *kout, uint32_t *vout,
                                              uint32_t valid, const DG &dg, uint32_t &run_off, const LB lb = {},
                                              const SRC src = {}, bool stream_in = false) {
    constexpr uint32_t THREADS = WAVES * 64;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;

    VRS_MARK(0);
    K key[ITEMS];
    uint32_t val[PAIRS ? ITEMS : 1];
    const uint32_t seg = wave * (ITEMS * 64) + lane;
    // stream_in (workgroup-uniform): the pass's input is larger than the caches and nobody reads it again -- nontemporal loads, which
    // leave the memory-side cache to what the pass WRITES (the next kernel reads that): 10^8 keys, contract scatter 138 -> 128 us, the
    // MSD passes 146 -> 140, the pool form's first pass 154 -> 143; below about 3e7 keys everything fits the caches and it costs a
    // little instead (10^7 keys: 0.108 -> 0.111 ms), so the callers switch it by size
    if (FULL && !SRC::kEnabled && stream_in) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) key[i] = __builtin_nontemporal_load(kin + seg + i * 64);
        if constexpr (PAIRS) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) val[i] = __builtin_nontemporal_load(vin + seg + i * 64);
        }
    } else if constexpr (SRC::kEnabled) {
        // every position's virtual slot first (a scalar loop over the pieces the chunk touches: usually two or three), then all loads at once
        uint32_t vs[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) vs[i] = src.first_slot;  // (positions behind `valid`: some readable slot)
#pragma unroll 1
        for (uint32_t p = src.p0; p < src.p1; ++p) {
            const uint32_t lo = __builtin_amdgcn_readlane(src.lo, p), len = __builtin_amdgcn_readlane(src.len, p), vd = __builtin_amdgcn_readlane(src.slot, p);
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const uint32_t idx = seg + i * 64;
                vs[i] = idx - lo < len ? vd + (idx - lo) : vs[i];
            }
        }
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const K k = *src.at(vs[i]);
            key[i] = (FULL || seg + i * 64 < valid) ? k : dg.template pad<K>();
        }
        if constexpr (PAIRS) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) val[i] = *src.val_at(vs[i]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t idx = seg + i * 64;
            if constexpr (FULL) {
                key[i] = kin[idx];
            } else {
                // unpredicated load from a clamped index, then select: the padding key (all ones, seen from the digit's base) has
                // the largest digit under every shift and the highest chunk indices, so it ranks behind every real key
                const K k = kin[idx < valid ? idx : valid - 1u];
                key[i] = idx < valid ? k : dg.template pad<K>();
            }
        }
        if constexpr (PAIRS) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const uint32_t idx = seg + i * 64;
                val[i] = vin[FULL ? idx : (idx < valid ? idx : valid - 1u)];
            }
        }
    }
    {
        uint32_t *z = &sm.whist[0][0];
#pragma unroll
        for (int v = 0; v < 4; ++v) z[v * THREADS + tid] = 0;  // WAVES*256 words / THREADS = 4 each
        if constexpr (LB::kEnabled) {
            if (tid == 0) sm.lb_gave_up = 0;
        }
    }
    __syncthreads();
    VRS_MARK(1);

    // ---- rank inside the wave
    uint32_t rank[ITEMS];
    uint32_t *my_hist = sm.whist[wave];
    if constexpr (RANK == RANK_ATOMIC) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t d = dg(key[i]);
            // Skew guard: 64 lanes on ONE counter are served one after the other (28 instead of 9 cycles per wave
            // instruction) -- input whose 64 consecutive keys share the digit (sorted keys under an MSD digit, constant
            // bytes) would crawl.  A wave-uniform digit needs no atomic per lane: lane 0 adds 64, rank = old + lane.
            const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
            if (__ballot(d == d0) == ~0ull) {  // wave-uniform branch
                uint32_t old = 0;
                if (lane == 0u)
                    old = __hip_atomic_fetch_add(&my_hist[d0], 64u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                rank[i] = __builtin_amdgcn_readfirstlane(old) + lane;
            } else
                rank[i] = __hip_atomic_fetch_add(&my_hist[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t d = dg(key[i]);
            const uint64_t peers = match_any_digit(d);
            const uint32_t below = count_below(peers);
            // all lanes read the counter, then the lowest matching lane adds the group's size with a
            // NON-returning atomic: no value flows from the read into the add, so the LDS operations of
            // all items pipeline; the LDS executes a wave's operations in order, so item i+1's read
            // observes item i's add.
            const uint32_t prev = my_hist[d];
            if (below == 0u)
                __hip_atomic_fetch_add(&my_hist[d], static_cast<uint32_t>(__popcll(peers)), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_wave_barrier();
            rank[i] = prev + below;
        }
    }
    __syncthreads();
    VRS_MARK(2);

    // ---- thread t == digit t: digit starts inside the chunk, per-wave starts, global base
    uint32_t lb_total = 0, lb_excl = 0;
    uint32_t lb_rows[LB::kEnabled ? kLbBatch : 1];
    {
        uint32_t c[WAVES];
        uint32_t total = 0;
        if (tid < kBins) {
#pragma unroll
            for (int v = 0; v < WAVES; ++v) {
                c[v] = sm.whist[v][tid];
                total += c[v];
            }
        }
        const uint32_t excl = block_exclusive_scan_w<WAVES>(total, sm.scan_tmp, lane, wave);
        if (tid < kBins) {
            uint32_t acc = excl;
#pragma unroll
            for (int v = 0; v < WAVES; ++v) {
                sm.whist[v][tid] = acc;
                acc += c[v];
            }
            if constexpr (LB::kEnabled) {
                // publish my count first (successors can add it without waiting for my look-back), then put the
                // first batch of predecessor rows in flight: the re-bucketing below hides their latency.  (Both AFTER
                // the scan: issuing them before it measured 4 us per pass slower, profiles/labs/r02_lookback_order.txt)
                lb.publish(total);
                if (!lb.foreign) lb.fetch(lb.index - 1, lb_rows);
                lb_total = total;
                lb_excl = excl;
            } else {
                sm.gbase[tid] = run_off - excl;
                run_off += total;
            }
        }
    }
    __syncthreads();
    VRS_MARK(3);

    // ---- re-bucket through LDS: all counter reads first, then all key writes
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) rank[i] += my_hist[dg(key[i])];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) sm.keys[rank[i]] = key[i];
    if constexpr (PAIRS) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) sm.vals[rank[i]] = val[i];
    }
    uint32_t lb_inclusive = 0;
    if constexpr (LB::kEnabled) {
        if (tid < kBins) {
            bool gave_up = false;
            const uint32_t before = (lb.foreign && !LB::kReserves) ? lb.recounted : lb.resolve(lb_rows, gave_up);
            if (gave_up) sm.lb_gave_up = 1;
            lb_inclusive = kLbInclusive | (before + lb_total);  // published below, after the LDS reads of the write-out
            if constexpr (LB::kPool) lb.place(sm.gbase, tid, before, lb_excl);  // (the pool form's regions: vrs_msd_pool.hip)
            else sm.gbase[tid] = lb.seed + before - lb_excl;
        }
    }
    __syncthreads();
    VRS_MARK(4);
    if constexpr (LB::kEnabled && !LB::kReserves) {  // (a reservation never waits)
        if (sm.lb_gave_up) {  // workgroup-uniform, never in a healthy run: a predecessor did not publish in time
            if constexpr (LB::kPool) {
                // the pool form's stable passes: the sort is refused (the caller's keys have not moved); the tile's keys go to SOME
                // place inside its regions
                __syncthreads();
                if (tid < kBins) {
                    lb.refuse();
                    lb_inclusive = kLbInclusive | lb_total;
                    lb.place(sm.gbase, tid, 0u, lb_excl);
                }
                __syncthreads();
            } else {
                uint32_t *cnt = sm.whist[0];  // the per-wave counters are dead from here on
                __syncthreads();
                if (tid < kBins) cnt[tid] = 0;
                __syncthreads();
                recount_keys(cnt, static_cast<const K *>(lb.stream_keys), lb.done, dg);
                __syncthreads();
                if (tid < kBins) {
                    const uint32_t before = cnt[tid];
                    lb_inclusive = kLbInclusive | (before + lb_total);
                    sm.gbase[tid] = lb.seed + before - lb_excl;
                }
                __syncthreads();
            }
        }
    }

    // ---- write out: position p of the chunk goes to gbase[digit] + p; reads batched before stores
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) key[i] = sm.keys[i * THREADS + tid];
    uint32_t dst[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) dst[i] = sm.gbase[dg(key[i])] + (i * THREADS + tid);
    if constexpr (LB::kEnabled) {
        // The inclusive count goes out HERE, as the first store of the write-out, not right after the look-back: the
        // registers the status loads landed in are reused by the LDS reads above, so the compiler waits for vmcnt(0)
        // before them -- and loads and stores retire on one in-order counter, so with the store already issued that
        // wait would hold the whole write-out back until the store is acknowledged.
        if (tid < kBins) lb.publish(lb_inclusive);
    }
    if constexpr (LB::kPool) {
        // the pool form's first pass: a slot is a VIRTUAL one (partner buffer, then overflow scratch), and the one run of a region
        // that crosses the end of its primary part continues elsewhere
        lb.template store<K, ITEMS, THREADS, FULL>(sm.gbase, key, dst, kout, valid, dg);
    } else {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            if (FULL || i * THREADS + tid < valid) kout[dst[i]] = key[i];
        }
    }
    VRS_MARK(5);

    if constexpr (PAIRS) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) val[i] = sm.vals[i * THREADS + tid];
        if constexpr (LB::kPool) {
            lb.template store_values<ITEMS, THREADS, FULL>(val, dst, vout, valid);  // (dst: what store() made of it)
        } else {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                if (FULL || i * THREADS + tid < valid) vout[dst[i]] = val[i];
            }
        }
    }
    // the next chunk's first barrier (after it zeroes the counters) separates these LDS reads from its writes
}

// the hybrid forms (K5b in vrs_msd_hybrid.hip, vrs_msd_pool.hip): an MSD partition by the top kMsdBits bits of the key range
constexpr uint32_t kMsdBits = 14, kMsdBuckets = 1u << kMsdBits;
constexpr uint32_t kMsdMinShift = 13, kMsdMaxShift = 18;  // bucket shift of a 27 ... 32-bit key range
// words behind the counts ([16384] histogram + [8][256] slice counts): the probed bucket shift and the "a key lies above
// the probed range" flag
constexpr uint32_t kMsdProbeWord = kMsdBuckets + 8u * 256u, kMsdOverWord = kMsdProbeWord + 1u;
constexpr uint32_t kMsdSubBits = kMsdBits - 8, kMsdSub = 1u << kMsdSubBits;  // buckets per top byte: 64

// a digit of fewer than 8 bits: (key >> shift) & mask
struct BitsDigit {
    uint32_t shift, mask, base;  // base: see RadixDigit (uint32 keys only)
    __device__ __forceinline__ uint32_t operator()(uint32_t key) const { return ((key - base) >> shift) & mask; }
    __device__ __forceinline__ uint32_t operator()(uint64_t key) const { return static_cast<uint32_t>(key >> shift) & mask; }
    template <typename KK>
    __device__ __forceinline__ KK pad() const { return sizeof(KK) == 4 ? static_cast<KK>(base - 1u) : static_cast<KK>(~static_cast<KK>(0)); }
};

}  // namespace vrs
