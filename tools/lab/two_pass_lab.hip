// two_pass_lab.hip -- round 6 lab: is a sort of 10^8 uint32 keys in TWO kernels (16 bytes per key) faster than the pool form's three (24)?
//   P  first pass by the top 12 bits: persistent workgroups (one per CU, 144 KB of LDS), tiles of 28672 keys ranked in LDS (one 4096-counter
//      table), every (list, digit) run placed by ONE reservation (an L2-local atomic add per tile and digit; list = the XCC the workgroup
//      runs on, so that the partial lines at a run's ends meet their neighbours inside one L2);
//   L  one workgroup per bucket of about 24 400 keys (112 KB of the CU's 160 KB LDS): the low 20 bits in two LDS passes (12 bits, one
//      table, ties in any order; 8 bits, a table per wave, stable), written to the bucket's final place.
// Both are software pipelines inside the one workgroup a CU holds: the next unit's loads are issued before this unit's LDS phases, this
// unit's stores drain during the next one's.  What V1 of this lab showed (profiles/labs/r06_two_pass.txt): __syncthreads() waits for
// vmcnt(0) -- every outstanding load AND store -- so nothing in flight survives a barrier; the barriers here order LDS traffic only
// (s_waitcnt lgkmcnt(0); s_barrier).  -DLAB_SYNC_FULL puts __syncthreads() back (the A/B).
// Lab conditions: uniform random keys, regions of a fixed generous size (the product would size them from a sample as the pool form does).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DLAB_THREADS=512] [-DLAB_MARKS] tools/lab/two_pass_lab.hip -o tools/lab/two_pass_lab
//   tools/lab/two_pass_lab [n] [reps] [lists: 8 | 1]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));                 \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

#ifndef LAB_NEXT_AT
#define LAB_NEXT_AT 2  // where L issues the next bucket's loads: 1 = before the first scatter, 2 = before the second
#endif
#ifndef LAB_THREADS
#define LAB_THREADS 1024
#endif

namespace {

constexpr int kThreads = LAB_THREADS, kWaves = kThreads / 64;
constexpr int kTopBits = 12, kTop = 1 << kTopBits;
constexpr int kTile = 28672;                                          // slots: a tile of P, the capacity of L
constexpr int kItems = kTile / kThreads, kVec = kItems / 4;           // 28 / 7 (1024 threads), 56 / 14 (512)
constexpr int kBpt = kTop / kThreads;                                 // counters a thread scans: 4 or 8
constexpr int kRow2 = 256 + 64 + 2;  // words per wave table of L's second pass: 256 counters, a dummy per lane, 2 so that rows fall on different banks
constexpr int kGroups = kWaves / 4;  // L's second scan: thread t holds digit t / kGroups of waves 4 (t % kGroups) ..
static_assert(kThreads * kItems == kTile && kVec * 4 == kItems, "shape");
static_assert(256 * kGroups == kThreads, "every thread of the second scan holds one (digit, group of four waves)");

__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3u << 11) | 20u) & 7u; }
__device__ __forceinline__ uint32_t lds_add(uint32_t *p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t opaque(uint32_t x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ uint4 load_nt(const uint4 *p) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(t.x, t.y, t.z, t.w);
}
// word `w` of a buffer of less than 4 GB: base (uniform, SGPRs) + a 32-bit BYTE offset -- the form a global access takes without a 64-bit address per lane
__device__ __forceinline__ uint32_t *at_word(uint32_t *base, uint32_t w) { return reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(base) + (w << 2)); }
__device__ __forceinline__ const uint32_t *at_word(const uint32_t *base, uint32_t w) { return reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(base) + (w << 2)); }
__device__ __forceinline__ void lds_barrier() {
#ifdef LAB_SYNC_FULL
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}
// exclusive prefix of `v` over the workgroup's threads in thread order (one barrier inside; s_tmp: kWaves words, free again after the NEXT barrier)
__device__ __forceinline__ uint32_t block_excl(uint32_t v, uint32_t *s_tmp) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if (lane >= static_cast<uint32_t>(o)) incl += t;
    }
    if (lane == 63u) s_tmp[wave] = incl;
    lds_barrier();
    const uint32_t w = lane < static_cast<uint32_t>(kWaves) ? s_tmp[lane] : 0u;
    uint32_t before = lane < wave ? w : 0u;
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) before += __shfl_xor(before, o);
    return incl - v + __builtin_amdgcn_readfirstlane(before);
}
// a thread's kBpt consecutive counters of a 4096-counter table
struct Bins {
    uint32_t c[kBpt];
    __device__ __forceinline__ void load(const uint32_t *tbl) {
#pragma unroll
        for (int v = 0; v < kBpt / 4; ++v) {
            const uint4 q = reinterpret_cast<const uint4 *>(tbl)[(kBpt / 4) * opaque(threadIdx.x) + v];
            c[4 * v] = q.x, c[4 * v + 1] = q.y, c[4 * v + 2] = q.z, c[4 * v + 3] = q.w;
        }
    }
    __device__ __forceinline__ void store(uint32_t *tbl) const {
#pragma unroll
        for (int v = 0; v < kBpt / 4; ++v) reinterpret_cast<uint4 *>(tbl)[(kBpt / 4) * opaque(threadIdx.x) + v] = make_uint4(c[4 * v], c[4 * v + 1], c[4 * v + 2], c[4 * v + 3]);
    }
    __device__ __forceinline__ uint32_t total() const {
        uint32_t t = 0;
#pragma unroll
        for (int j = 0; j < kBpt; ++j) t += c[j];
        return t;
    }
};
__device__ __forceinline__ void zero_table(uint32_t *tbl, int words) {  // words: a multiple of 4
    for (int v = opaque(threadIdx.x); v < words / 4; v += kThreads) reinterpret_cast<uint4 *>(tbl)[v] = make_uint4(0, 0, 0, 0);
}

#ifdef LAB_MARKS
#define MARK(j) do { if (threadIdx.x == 0 && blockIdx.x == 3) marks[(mk_it * 8 + (j)) & 1023] = wall_clock64(); } while (0)
#else
#define MARK(j)
#endif

struct PArgs {
    unsigned long long *marks;
    const uint32_t *in;
    uint32_t *mid;
    uint32_t *cursors;   // [lists][4096]: keys of (list, digit) placed so far
    uint32_t *tickets;   // [8]: next tile of list x
    uint32_t *fail;
    uint32_t tiles_total, tiles_per_list, lists, region_cap, shift;
};

// ---- P: the first pass
// Registers (1024 threads: 128 per lane): at every moment at most three arrays of kItems words are alive -- the tile's keys, their ranks and
// either the LDS reads in flight or the NEXT tile's keys, whose loads are issued just before the scatter into LDS: they fly through the
// scatter and the write-out, this tile's stores drain through the next tile's ranking.
template <int LISTS>
__global__ __launch_bounds__(kThreads, 1) void first_pass_kernel(PArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[kTile];
    __shared__ __attribute__((aligned(16))) uint32_t s_tbl[kTop];
    __shared__ __attribute__((aligned(16))) uint32_t s_gbase[kTop];
    __shared__ uint32_t s_tmp[2][16];
    __shared__ uint32_t s_slot;
    const uint32_t tid = threadIdx.x;
    const uint32_t list = LISTS == 8 ? xcc_id() : 0u;
    const uint32_t t_lo = list * a.tiles_per_list, t_hi = min(t_lo + a.tiles_per_list, a.tiles_total);
    zero_table(s_tbl, kTop);
    if (tid == 0) s_slot = atomicAdd(&a.tickets[list], 1u);  // which of its list's workgroups this one is (lab: gridDim / LISTS of them per list)
    lds_barrier();
    const uint32_t stride = gridDim.x / LISTS;
    uint32_t tile = t_lo + s_slot, it = 0;
    uint32_t kc[kItems];
    const auto load_tile = [&](uint32_t (&k)[kItems], uint32_t t) {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.in + static_cast<size_t>(t) * kTile);
#pragma unroll
        for (int j = 0; j < kVec; ++j) {
            const uint4 q = load_nt(src + j * kThreads + tid);
            k[4 * j] = q.x, k[4 * j + 1] = q.y, k[4 * j + 2] = q.z, k[4 * j + 3] = q.w;
        }
    };
    if (tile < t_hi) load_tile(kc, tile);
    [[maybe_unused]] unsigned long long *marks = a.marks;
    [[maybe_unused]] uint32_t mk_it = 0;
    while (tile < t_hi) {
        const uint32_t next = tile + stride;
        const uint32_t tid = opaque(threadIdx.x);  // (or what depends on it alone is hoisted out of the loop and kept alive)
        uint32_t kn[kItems];  // (declared HERE: a loop-carried array would stay alive through the phases that do not need it)
        MARK(0);
        uint32_t rank[kItems];
#pragma unroll
        for (int i = 0; i < kItems; ++i) {
            rank[i] = lds_add(&s_tbl[kc[i] >> a.shift], 1u);
            if (i % 4 == 3) __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();
        MARK(1);
        // thread t owns digits kBpt t ..: their starts inside the tile, their reservations
        Bins c;
        c.load(s_tbl);
        uint32_t res[kBpt];
        {
            uint32_t *cur = a.cursors + list * kTop + kBpt * tid;
#pragma unroll
            for (int j = 0; j < kBpt; ++j)
                res[j] = __hip_atomic_fetch_add(cur + j, c.c[j], __ATOMIC_RELAXED, LISTS == 8 ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_AGENT);  // (8 lists: performed in this XCD's L2)
        }
        Bins e;
        {
            uint32_t acc = block_excl(c.total(), s_tmp[it & 1u]);
#pragma unroll
            for (int j = 0; j < kBpt; ++j) {
                e.c[j] = acc;
                acc += c.c[j];
            }
        }
        e.store(s_tbl);
        lds_barrier();
        MARK(2);
        {
            uint32_t t[kItems];
#pragma unroll
            for (int i = 0; i < kItems; ++i) t[i] = s_tbl[opaque(kc[i]) >> a.shift];
#pragma unroll
            for (int i = 0; i < kItems; ++i) rank[i] = opaque(rank[i] + t[i]);  // (opaque: the sum NOW, in one register -- not rank and base kept apart until the scatter)
        }
#if LAB_NEXT_AT != 0
        __builtin_amdgcn_sched_barrier(0);
        if (next < t_hi) load_tile(kn, next);
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int i = 0; i < kItems; ++i) s_keys[rank[i]] = kc[i];
        {
            Bins g;
            bool over = false;
#pragma unroll
            for (int j = 0; j < kBpt; ++j) {
                g.c[j] = (list * kTop + kBpt * tid + j) * a.region_cap + res[j] - e.c[j];
                over |= res[j] + c.c[j] > a.region_cap;
            }
            g.store(s_gbase);
            if (over) *a.fail = 1u;
        }
        lds_barrier();
        MARK(3);
        zero_table(s_tbl, kTop);
        {
            uint32_t key[kItems], dst[kItems];
#pragma unroll
            for (int i = 0; i < kItems; ++i) key[i] = s_keys[i * kThreads + tid];
#pragma unroll
            for (int i = 0; i < kItems; ++i) dst[i] = s_gbase[key[i] >> a.shift] + (i * kThreads + tid);
#pragma unroll
            for (int i = 0; i < kItems; ++i) *at_word(a.mid, dst[i]) = key[i];
        }
        MARK(4);
        lds_barrier();
        MARK(5);
#if LAB_NEXT_AT == 0
        if (next < t_hi) load_tile(kn, next);
#endif
#pragma unroll
        for (int i = 0; i < kItems; ++i) kc[i] = kn[i];
        tile = next;
        ++it;
        ++mk_it;
    }
}

// ---- plan: bucket totals and where every bucket starts in the sorted order
__global__ __launch_bounds__(1024) void plan_kernel(const uint32_t *cursors, uint32_t lists, uint32_t *begin /*[4097]*/) {
    __shared__ uint32_t s_tmp[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t c[4] = {0, 0, 0, 0};
    for (uint32_t x = 0; x < lists; ++x) {
        const uint4 q = reinterpret_cast<const uint4 *>(cursors + x * kTop)[tid];
        c[0] += q.x, c[1] += q.y, c[2] += q.z, c[3] += q.w;
    }
    const uint32_t v = c[0] + c[1] + c[2] + c[3];
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if (lane >= static_cast<uint32_t>(o)) incl += t;
    }
    if (lane == 63u) s_tmp[wave] = incl;
    __syncthreads();
    uint32_t excl = incl - v;
    for (uint32_t w = 0; w < wave; ++w) excl += s_tmp[w];
    reinterpret_cast<uint4 *>(begin)[tid] = make_uint4(excl, excl + c[0], excl + c[0] + c[1], excl + c[0] + c[1] + c[2]);
    if (tid == 1023u) begin[kTop] = excl + v;
}

struct LArgs {
    unsigned long long *marks;
    const uint32_t *mid;
    uint32_t *out;
    const uint32_t *cursors, *begin;
    uint32_t *fail;
    uint32_t lists, region_cap, buckets;
};

// the descriptor of bucket d, one word per lane: lanes 0..7 the keys of piece x (list x's share), lane 8 where the bucket starts in the output
__device__ __forceinline__ uint32_t bucket_desc(const LArgs &a, uint32_t d) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t v = 0;
    if (lane < a.lists) v = a.cursors[lane * kTop + d];
    else if (lane == 8u) v = a.begin[d];
    return v;
}
// A bucket is READ piece by piece: piece x by the waves kWpp x .. kWpp x + kWpp - 1, item i of a lane is key (wave % kWpp) * 64 + lane + 64 kWpp i
// of the piece -- one add per address, 256 contiguous bytes per wave and load.  Pass 1 ranks keys in any order, so where a key was read
// does not matter; items at or behind the piece's end hold nothing (a dummy counter, a dump slot).  Lab: a piece has at most
// 64 kWpp kItems = 3584 keys (uniform keys: 3052 +- 55) and at least 64 kWpp kCheckFrom.
constexpr int kWpp = kWaves / 8;
constexpr int kCheckFrom = kItems / 2;
struct Piece {
    uint32_t first;  // word of the middle buffer this lane's item 0 reads
    uint32_t idx0;   // its index inside the piece
    uint32_t cnt;    // keys of the piece (wave-uniform)
};
__device__ __forceinline__ Piece piece_of(const LArgs &a, uint32_t d, uint32_t desc, uint32_t tid) {
    const uint32_t wave = tid >> 6, lane = tid & 63u, x = __builtin_amdgcn_readfirstlane(wave / kWpp);
    Piece p;
    p.cnt = x < a.lists ? __builtin_amdgcn_readlane(desc, x) : 0u;
    p.idx0 = (wave % kWpp) * 64u + lane;
    p.first = (x * kTop + d) * a.region_cap + p.idx0;
    return p;
}
__device__ __forceinline__ void load_bucket(uint32_t (&k)[kItems], const LArgs &a, const Piece &pc) {
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        uint32_t off = 64u * kWpp * i;
        if (i >= kCheckFrom) off = min(pc.idx0 + off, pc.cnt - 1u) - pc.idx0;  // (a readable word)
        k[i] = *at_word(a.mid, pc.first + off);
    }
}

// the sort phases of one bucket whose keys are in k (as load_bucket left them), the next bucket's loads issued before the first scatter
template <int VEC, typename NEXT>  // VEC: 16-byte vectors per lane that cover the bucket's n positions
__device__ __forceinline__ void local_sort_phases(const LArgs &a, uint32_t (&k)[kItems], uint32_t cnt, uint32_t n, uint32_t out_begin, uint32_t *s_keys, uint32_t *s_t1,
                                                  uint32_t *s_t2, uint32_t *s_tmp, [[maybe_unused]] uint32_t mk_it, const NEXT &issue_next) {
    constexpr int ITEMS = 4 * VEC;
    // (opaque: or everything that depends on the thread index alone -- slot numbers, LDS and buffer offsets -- is hoisted out of the
    //  bucket loop and kept alive, a hundred registers and more)
    const uint32_t tid = opaque(threadIdx.x), lane = tid & 63u, wave = tid >> 6;
    const uint32_t idx0 = (wave % kWpp) * 64u + lane;
    [[maybe_unused]] unsigned long long *marks = a.marks;
    uint32_t rank[kItems];
    MARK(0);
    // ---- pass 1: bits 0..11, one table, ties in any order
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        uint32_t c = k[i] & 4095u;
        if (i >= kCheckFrom) c = (idx0 + 64u * kWpp * i < cnt) ? c : 4096u + lane;
        rank[i] = lds_add(&s_t1[c], 1u);
        if (i % 4 == 3) __builtin_amdgcn_sched_barrier(0);
    }
    lds_barrier();
    MARK(1);
    {
        Bins c, e;
        c.load(s_t1);
        uint32_t acc = block_excl(c.total(), s_tmp);
#pragma unroll
        for (int j = 0; j < kBpt; ++j) {
            e.c[j] = acc;
            acc += c.c[j];
        }
        e.store(s_t1);
    }
    lds_barrier();
    // position L goes to slot (L & ~255) | ((L & 63) << 2) | ((L >> 6) & 3): a lane's 16-byte read below is positions l, 64 + l, 128 + l, 192 + l;
    // an item without a key goes to a dump slot behind the bucket
    {
        uint32_t t[kItems];
#pragma unroll
        for (int i = 0; i < kItems; ++i) {
            uint32_t c = opaque(k[i]) & 4095u;
            if (i >= kCheckFrom) c = (idx0 + 64u * kWpp * i < cnt) ? c : 4096u + lane;
            t[i] = s_t1[c];
        }
#pragma unroll
        for (int i = 0; i < kItems; ++i) rank[i] = opaque(rank[i] + t[i]);  // (opaque: the sum NOW, in one register)
    }
#if LAB_NEXT_AT == 1
    __builtin_amdgcn_sched_barrier(0);
    issue_next();  // the next bucket's keys: in flight from here to the end of this bucket
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
        const uint32_t L = rank[i];
        uint32_t slot = (L & ~255u) | ((L & 63u) << 2) | ((L >> 6) & 3u);
        if (i >= kCheckFrom) slot = (idx0 + 64u * kWpp * i < cnt) ? slot : kTile + lane;
        s_keys[slot] = k[i];
        if (i % 4 == 3) __builtin_amdgcn_sched_barrier(0);
    }
    MARK(2);
    lds_barrier();
    MARK(3);
    zero_table(s_t1, kTop + 64);  // for the next bucket
    // ---- pass 2: bits 12..19, a table per wave, stable (instruction order, then lane order)
    const uint32_t seg = wave * (ITEMS * 64);
    uint32_t k2[ITEMS];
#pragma unroll
    for (int g = 0; g < VEC; ++g) {
        const uint4 t = reinterpret_cast<const uint4 *>(s_keys + seg + g * 256)[lane];
        k2[4 * g] = t.x, k2[4 * g + 1] = t.y, k2[4 * g + 2] = t.z, k2[4 * g + 3] = t.w;
    }
    uint32_t *my = s_t2 + wave * kRow2;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        uint32_t c = (k2[i] >> 12) & 255u;
        c = (seg + i * 64 + lane < n) ? c : 256u + lane;
        rank[i] = lds_add(&my[c], 1u);
        if (i % 4 == 3) __builtin_amdgcn_sched_barrier(0);
    }
    lds_barrier();
    MARK(4);
    {   // exclusive prefix over (digit, wave): thread t holds digit t / kGroups, waves 4 (t % kGroups) ..
        const uint32_t dg = tid / kGroups, w0 = 4u * (tid % kGroups);
        uint32_t c[4], total = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c[j] = s_t2[(w0 + j) * kRow2 + dg];
            total += c[j];
        }
        uint32_t acc = block_excl(total, s_tmp + 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s_t2[(w0 + j) * kRow2 + dg] = acc;
            acc += c[j];
        }
    }
    lds_barrier();
#pragma unroll
    for (int i0 = 0; i0 < ITEMS; i0 += 8) {  // in batches: the reads in flight take registers beside k2, rank and the next bucket's keys
        uint32_t t[8];
#pragma unroll
        for (int i = i0; i < i0 + 8 && i < ITEMS; ++i) {
            uint32_t c = (opaque(k2[i]) >> 12) & 255u;
            c = (seg + i * 64 + lane < n) ? c : 256u + lane;
            t[i - i0] = my[c];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = i0; i < i0 + 8 && i < ITEMS; ++i) {
            const uint32_t L = seg + i * 64 + lane;
            rank[i] = opaque(L < n ? rank[i] + t[i - i0] : L);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#if LAB_NEXT_AT == 2
    __builtin_amdgcn_sched_barrier(0);
    issue_next();  // the next bucket's keys: in flight through the second scatter and the write-out
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) s_keys[rank[i]] = k2[i];
    lds_barrier();
    MARK(5);
    for (uint32_t c = tid; c < kWaves * kRow2; c += kThreads) s_t2[c] = 0;  // for the next bucket
    uint32_t *dst = a.out + out_begin;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t p = i * kThreads + tid;
        const uint32_t v = s_keys[p];
        if (p < n) __builtin_nontemporal_store(v, at_word(dst, p));
    }
    MARK(6);
    lds_barrier();
    MARK(7);
}

__global__ __launch_bounds__(kThreads, 1) void local_sort_kernel(LArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[kTile + 64];
    __shared__ __attribute__((aligned(16))) uint32_t s_t1[kTop + 64];
    __shared__ __attribute__((aligned(16))) uint32_t s_t2[kWaves * kRow2];
    __shared__ uint32_t s_tmp[32];
    zero_table(s_t1, kTop + 64);
    for (uint32_t c = threadIdx.x; c < kWaves * kRow2; c += kThreads) s_t2[c] = 0;
    lds_barrier();
    // bucket j of this workgroup: d(j) = buckets - 1 - (blockIdx + j * grid); descriptors run two ahead, keys one ahead
    const auto bucket_at = [&](uint32_t j) { return a.buckets - 1u - (blockIdx.x + j * gridDim.x); };
    const auto exists = [&](uint32_t j) { return blockIdx.x + j * gridDim.x < a.buckets; };
    if (!exists(0)) return;
    uint32_t desc_c = bucket_desc(a, bucket_at(0));
    uint32_t desc_n = exists(1) ? bucket_desc(a, bucket_at(1)) : 0u;
    uint32_t kc[kItems];
    load_bucket(kc, a, piece_of(a, bucket_at(0), desc_c, threadIdx.x));
    for (uint32_t j = 0; exists(j); ++j) {
        const uint32_t tid = opaque(threadIdx.x);
        uint32_t n = 0, lo = 0xFFFFFFFFu, hi = 0;
#pragma unroll
        for (uint32_t x = 0; x < 8u; ++x) {
            const uint32_t c = __builtin_amdgcn_readlane(desc_c, x);
            n += c;
            lo = min(lo, c);
            hi = max(hi, c);
        }
        const uint32_t out_begin = __builtin_amdgcn_readlane(desc_c, 8);
        const uint32_t cnt = piece_of(a, bucket_at(j), desc_c, tid).cnt;
        uint32_t desc_nn = 0;
        uint32_t kn[kItems];  // (declared HERE: a loop-carried array would stay alive through the phases that do not need it)
        const auto issue_next = [&]() {
            if (exists(j + 2u)) desc_nn = bucket_desc(a, bucket_at(j + 2u));
            if (exists(j + 1u)) load_bucket(kn, a, piece_of(a, bucket_at(j + 1u), desc_n, tid));
        };
        // (lab: pieces the read shape does not take -- the product would read such a bucket another way, or leave the sort to the pool form)
        const bool takes = n != 0u && n <= static_cast<uint32_t>(kTile) && hi <= 64u * kWpp * kItems && (a.lists == 1u || lo >= 64u * kWpp * kCheckFrom);
        if (!takes) {
            *a.fail = 2u;
            issue_next();
        } else local_sort_phases<kVec>(a, kc, cnt, n, out_begin, s_keys, s_t1, s_t2, s_tmp, j, issue_next);  // (one shape: a second one behind a branch doubles what the allocator keeps alive)
#if LAB_NEXT_AT == 0
        issue_next();  // (no register prefetch: the next bucket's loads are issued behind this bucket's stores and awaited at once)
#endif
#pragma unroll
        for (int i = 0; i < kItems; ++i) kc[i] = kn[i];
        desc_c = desc_n;
        desc_n = desc_nn;
    }
}

__global__ void fill_random(uint32_t *k, uint32_t n, uint32_t seed) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t x = i * 0x9E3779B9u + seed;
        x ^= x >> 16, x *= 0x85EBCA6Bu, x ^= x >> 13, x *= 0xC2B2AE35u, x ^= x >> 16;
        k[i] = x;
    }
}
__global__ void check_sorted(const uint32_t *k, uint32_t n, unsigned long long *out) {
    unsigned long long bad = 0, sum = 0;
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        if (i + 1 < n && k[i] > k[i + 1]) ++bad;
        sum += k[i] * 0x9E3779B97F4A7C15ull + (k[i] ^ 0x5555u);
    }
    if (bad) atomicAdd(&out[0], bad);
    atomicAdd(&out[1], sum);
}

}  // namespace

int main(int argc, char **argv) {
    const uint32_t n_asked = argc > 1 ? static_cast<uint32_t>(atof(argv[1])) : 100000000u;
    const int reps = argc > 2 ? atoi(argv[2]) : 8;
    const uint32_t lists = argc > 3 ? static_cast<uint32_t>(atoi(argv[3])) : 8u;
    const uint32_t tiles = (n_asked + kTile - 1) / kTile, n = tiles * kTile, tpl = lists == 8u ? (tiles + 7u) / 8u : tiles;  // (the lab sorts whole tiles)
    // regions: a (list, digit) share is Binomial(n / lists, 1 / 4096): mean + 8 deviations, a multiple of 32 slots
    const double mean = static_cast<double>(n) / lists / kTop;
    const uint32_t region_cap = (static_cast<uint32_t>(mean + 8.0 * std::sqrt(mean)) + 31u) & ~31u;
    const size_t mid_slots = static_cast<size_t>(lists) * kTop * region_cap;
    std::printf("n %u (%u tiles of %d)  %d threads x %d keys  lists %u  region %u slots  middle buffer %.0f MB\n", n, tiles, kTile, kThreads, kItems, lists, region_cap, mid_slots * 4.0 / 1e6);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    uint32_t *in[3], *mid, *out, *cursors, *tickets, *fail, *begin;
    unsigned long long *chk, *marks;
    CK(hipMalloc(&marks, 2 * 8192));
    CK(hipMemset(marks, 0, 2 * 8192));
    for (auto &p : in) CK(hipMalloc(&p, 4ull * n));
    CK(hipMalloc(&mid, 4ull * mid_slots));
    CK(hipMalloc(&out, 4ull * n));
    CK(hipMalloc(&cursors, 4ull * 8 * kTop));
    CK(hipMalloc(&tickets, 64));
    CK(hipMalloc(&fail, 4));
    CK(hipMalloc(&begin, 4ull * (kTop + 4)));
    CK(hipMalloc(&chk, 32));
    CK(hipMemset(fail, 0, 4));
    hipEvent_t ev[4];
    for (auto &e : ev) CK(hipEventCreate(&e));
    int dev = 0, cus = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    double sum[3] = {0, 0, 0};
    for (int r = 0; r < reps + 2; ++r) {
        uint32_t *src = in[r % 3];
        hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, st, src, n, 777u + r);
        CK(hipMemsetAsync(chk, 0, 32, st));
        hipLaunchKernelGGL(check_sorted, dim3(2048), dim3(256), 0, st, src, n, chk + 2);
        CK(hipMemsetAsync(cursors, 0, 4ull * 8 * kTop, st));
        CK(hipMemsetAsync(tickets, 0, 64, st));
        PArgs pa{marks, src, mid, cursors, tickets, fail, tiles, tpl, lists, region_cap, 32u - kTopBits};
        CK(hipEventRecord(ev[0], st));
        if (lists == 8u) hipLaunchKernelGGL(first_pass_kernel<8>, dim3(cus), dim3(kThreads), 0, st, pa);
        else hipLaunchKernelGGL(first_pass_kernel<1>, dim3(cus), dim3(kThreads), 0, st, pa);
        CK(hipEventRecord(ev[1], st));
        hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(1024), 0, st, cursors, lists, begin);
        CK(hipEventRecord(ev[2], st));
        LArgs la{marks + 1024, mid, out, cursors, begin, fail, lists, region_cap, static_cast<uint32_t>(kTop)};
        hipLaunchKernelGGL(local_sort_kernel, dim3(cus), dim3(kThreads), 0, st, la);
        CK(hipEventRecord(ev[3], st));
        hipLaunchKernelGGL(check_sorted, dim3(2048), dim3(256), 0, st, out, n, chk);
        CK(hipStreamSynchronize(st));
        CK(hipGetLastError());
        unsigned long long hc[4];
        uint32_t hf = 0;
        CK(hipMemcpy(hc, chk, 32, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost));
        float t[3];
        for (int i = 0; i < 3; ++i) CK(hipEventElapsedTime(&t[i], ev[i], ev[i + 1]));
        std::printf("rep %d: P %.1f us  plan %.1f us  L %.1f us  total %.1f us  | out of order %llu  checksum %s  fail %u\n", r, t[0] * 1e3, t[1] * 1e3, t[2] * 1e3,
                    (t[0] + t[1] + t[2]) * 1e3, hc[0], hc[1] == hc[3] ? "ok" : "DIFFERS", hf);
        if (r >= 2)
            for (int i = 0; i < 3; ++i) sum[i] += t[i] * 1e3;
    }
#ifdef LAB_MARKS
    {
        std::vector<unsigned long long> hm(2048);
        CK(hipMemcpy(hm.data(), marks, 2 * 8192, hipMemcpyDeviceToHost));
        for (int kz = 0; kz < 2; ++kz) {
            std::printf("%s phases of workgroup 3 (us, 100 MHz clock), iterations 2..9:\n", kz ? "L" : "P");
            for (int itn = 2; itn < 10; ++itn) {
                const unsigned long long *m = hm.data() + 1024 * kz + 8 * itn;
                std::printf("  it %d:", itn);
                for (int j = 1; j < 8; ++j) std::printf(" %6.2f", m[j] >= m[j - 1] ? (m[j] - m[j - 1]) / 100.0 : -1.0);
                std::printf("   | whole %6.2f\n", (hm[1024 * kz + 8 * (itn + 1)] - m[0]) / 100.0);
            }
        }
    }
#endif
    const double tot = (sum[0] + sum[1] + sum[2]) / reps;
    std::printf("mean over %d: P %.1f  plan %.1f  L %.1f  total %.1f us  = %.1f Gkeys/s (16 B/key: %.2f of 8 TB/s)\n", reps, sum[0] / reps, sum[1] / reps, sum[2] / reps, tot,
                n / tot / 1e3, 16.0 * n / (tot * 1e-6) / 8e12);
    return 0;
}
