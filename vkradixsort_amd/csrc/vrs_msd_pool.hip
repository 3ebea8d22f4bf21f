// vrs_msd_pool.hip -- the hybrid form of the one-call sort WITHOUT a counting read (bare uint32 keys): 24 bytes per key.
//
// The reference reads the keys once per pass just to count them (multi_radixsort_histograms.comp:42-50); the counted hybrid
// form (vrs_kernels.hip, K5b) still reads them once for that.  Here they are not read for counting at all:
//
//   pool_sample_kernel   1/32 of the input (the first 256 keys of every 8192-key tile): probes the key range (bucket shift)
//                        and counts the top byte per input slice; its last workgroup lays out, for every (slice, top byte),
//                        a PRIMARY region of the partner buffer sized by the estimate (the estimates of a slice sum to its
//                        length, so the regions tile the n-key buffer exactly) and an OVERFLOW region in context scratch of six
//                        standard deviations of that estimate;
//   pool_pass_a_kernel   first MSD pass (top 8 bits of the key range).  Persistent workgroups, 64 per slice: a tile reserves
//                        its place in (slice, top byte)'s region with ONE atomic add on the region's cursor, in the L2 of the
//                        XCD that runs the slice (StreamReserve's idea, vrs_device.hpp) -- positions below the region's
//                        capacity are primary slots, the rest overflow slots.  On the way every key is counted in a 16384-bin
//                        histogram of the top 14 bits: packed 16-bit LDS counters that live as long as the workgroup and are
//                        flushed to memory once (64-bit atomics on counter pairs), so the flush costs 0.04 atomics per key;
//   pool_plan_kernel     ONE workgroup: exact bucket offsets from the histogram, the second pass's tile tables from the cursors,
//                        the verdict -- no region overflowed, no key outside the sampled range, every bucket fits the local
//                        sort -- and the host head;
//   pool_pass_b_kernel   second MSD pass (the next 6 bits): walks every (top byte, slice) share -- primary part, then overflow
//                        part -- in tiles and writes every bucket to its FINAL range by reservation, exactly like the counted
//                        form's second pass (scatter_chunk with two sources);
//   the local sort       msd_local_sort_*_kernel of the counted form, unchanged.
//
// Nothing here is assumed about the data: a sample that misjudges a region (keys whose distribution changes inside a tile
// with the tile's period, say) makes the first pass flag the sort, the plan refuse, and the caller run the counted form on the
// untouched input.  The first pass is not stable (arrival order inside a region); bare keys do not care.
#include "vrs_device.hpp"

#include <algorithm>
#include <cmath>

namespace vrs {

namespace {

constexpr uint32_t kPoolBuckets = kMsdBucketCount;        // 16384
constexpr uint32_t kPoolMinShift = 13, kPoolMaxShift = 18;  // a 27 ... 32-bit key range (the counted form's rule)
constexpr uint32_t kPoolFlushAt = 65535u - kPoolTile;     // a 16-bit counter may take one more tile below this
constexpr float kPoolSigmas = 6.0f;                       // overflow room, in standard deviations of the region's estimate
constexpr uint32_t kPoolRoomFloor = 160;

__device__ __forceinline__ uint32_t xcc_of(unsigned long long xcc_map, uint32_t x) {
    return static_cast<uint32_t>((xcc_map >> (8u * x)) & 0xFFu);
}

// ---------------------------------------------------------------------------------------------
// The sample.  Workgroup g takes tiles [32 g, 32 g + 32) of the input; wave w of it the tiles 32 g + w + 4 j.
__global__ __launch_bounds__(256) void pool_sample_kernel(const uint32_t *__restrict__ keys, uint32_t n, uint32_t key_base,
                                                          PoolStreams ps, PoolPlan *__restrict__ pool, uint32_t overflow_capacity) {
    constexpr int kPerWave = kPoolSampleTiles / 4;        // tiles per wave
    constexpr int kLoads = kPoolSampleKeys / 64;          // 4-byte loads per lane and tile
    __shared__ uint32_t s_hist[2][256];
    __shared__ uint32_t s_or, s_last;
    __shared__ uint32_t s_wave[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    s_hist[0][tid] = 0;
    s_hist[1][tid] = 0;
    if (tid == 0) s_or = 0;
    // this workgroup's share of the sample first (its loads fly while the probe is reduced)
    const uint32_t t0 = blockIdx.x * kPoolSampleTiles;
    const uint32_t s0 = t0 / ps.tiles_per_stream;
    uint32_t k[kPerWave][kLoads];
#pragma unroll
    for (int j = 0; j < kPerWave; ++j) {
        const uint64_t begin = static_cast<uint64_t>(t0 + wave + 4u * j) * kPoolTile;
#pragma unroll
        for (int c = 0; c < kLoads; ++c) {
            const uint64_t idx = begin + c * 64u + lane;
            k[j][c] = keys[idx < n ? idx : n - 1u];
        }
    }
    // The buckets are the top 14 bits of the key RANGE: every workgroup ORs the same strided 4096 keys (as the counted form's
    // counting read does) and derives the same shift; a key outside that range is flagged by the first pass.
    {
        const uint32_t samples = min(n, 4096u);
        const uint64_t stride = n / samples;
        uint32_t acc = 0;
        for (uint32_t i = tid; i < samples; i += 256u) acc |= keys[static_cast<uint64_t>(i) * stride] - key_base;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc |= __shfl_down(acc, o);
        __syncthreads();
        if (lane == 0u && acc) atomicOr(&s_or, acc);
    }
    __syncthreads();
    const uint32_t bits = s_or ? 32u - static_cast<uint32_t>(__clz(static_cast<int>(s_or))) : 0u;
    const uint32_t shift = bits > kMsdBits ? bits - kMsdBits : 0u;
    const uint32_t dshift = shift + (kMsdBits - 8u);  // the first pass's digit: the top 8 bits of the range
#pragma unroll
    for (int j = 0; j < kPerWave; ++j) {
        const uint32_t t = t0 + wave + 4u * j;
        const uint64_t begin = static_cast<uint64_t>(t) * kPoolTile;
        const uint32_t h = t / ps.tiles_per_stream - s0;  // 0 or 1: a workgroup's 32 tiles touch at most two slices
#pragma unroll
        for (int c = 0; c < kLoads; ++c) {
            if (begin + c * 64u + lane < n) {
                const uint32_t d = min((k[j][c] - key_base) >> dshift, 255u);
                atomicAdd(&s_hist[h & 1u][d], 1u);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t h = 0; h < 2u; ++h) {
        const uint32_t v = s_hist[h][tid];
        if (v) __hip_atomic_fetch_add(&pool->sample[s0 + h][tid], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // The workgroup that finishes LAST lays the regions out.  Everything handed over is an agent-scope atomic on both sides
    // (the counts above, the ticket, the loads below), performed where every XCD sees it: no fence is needed, only that
    // this workgroup's adds have been performed before its ticket is drawn (vmcnt(0) in every wave, then the barrier).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const uint32_t ticket = __hip_atomic_fetch_add(&pool->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = ticket == gridDim.x - 1u ? 1u : 0u;
        if (s_last) __hip_atomic_store(&pool->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    // thread d: top byte d of all eight slices.  Regions are laid out top byte by top byte, slice by slice.
    uint32_t cap[8], room[8], caps = 0, rooms = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const uint32_t m = __hip_atomic_load(&pool->sample[s][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&pool->sample[s][tid], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero for the next sort
        const uint32_t len = ps.len[s], sampled = ps.sampled[s];
        // estimate: the slice's keys in proportion to the sample's; the estimates of a slice sum to at most its length
        const uint32_t est = sampled ? static_cast<uint32_t>(static_cast<uint64_t>(m) * len / sampled) : 0u;
        cap[s] = est & ~31u;
        // the estimate scales m sampled keys up by r = len / sampled: its standard deviation is sqrt(r * est)
        const float r = sampled ? static_cast<float>(len) / static_cast<float>(sampled) : 1.0f;
        const uint32_t dev = static_cast<uint32_t>(kPoolSigmas * sqrtf(r * static_cast<float>(est)));
        room[s] = len ? (dev + (est - cap[s]) + kPoolRoomFloor + 31u) & ~31u : 0u;
        caps += cap[s];
        rooms += room[s];
    }
    // exclusive prefix of (caps, rooms) over the 256 threads
    uint32_t ic = caps, ir = rooms;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t a = __shfl_up(ic, o), b = __shfl_up(ir, o);
        if (lane >= static_cast<uint32_t>(o)) {
            ic += a;
            ir += b;
        }
    }
    __shared__ uint32_t s_room[4];
    if (lane == 63u) {
        s_wave[wave] = ic;
        s_room[wave] = ir;
    }
    __syncthreads();
    uint32_t bc = ic - caps, br = ir - rooms, total_room = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4u; ++w) {
        bc += w < wave ? s_wave[w] : 0u;
        br += w < wave ? s_room[w] : 0u;
        total_room += s_room[w];
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        pool->base[s][tid] = bc;
        pool->cap[s][tid] = cap[s];
        pool->obase[s][tid] = br;
        pool->ocap[s][tid] = room[s];
        bc += cap[s];
        br += room[s];
    }
    if (tid == 0) {
        pool->shift = shift;
        // a key range below 27 bits is left to the LSD passes, like the counted form does (vrs_kernels.hip, msd_plan_kernel)
        pool->armed = (shift >= kPoolMinShift && shift <= kPoolMaxShift && total_room <= overflow_capacity) ? 1u : 0u;
    }
}

// ---------------------------------------------------------------------------------------------
// First pass.
#ifdef VRS_POOL_LAB_MARKS  // lab (tools/lab/pool_lab.hip): thread 0 of every workgroup sums the cycles between phase marks
__device__ unsigned long long *g_pool_marks;
#define POOL_MARK(k)                                            \
    do {                                                        \
        if (threadIdx.x == 0) {                                 \
            const unsigned long long t_ = __builtin_readcyclecounter(); \
            sm.marks[k] += t_ - sm.mark_last;                   \
            sm.mark_last = t_;                                  \
        }                                                       \
    } while (0)
#else
#define POOL_MARK(k)
#endif
struct alignas(16) PoolSmem {
#ifdef VRS_POOL_LAB_MARKS
    unsigned long long marks[12], mark_last;
#endif
    uint32_t keys[kPoolTile];      // re-bucketed keys, tile order by top byte
    alignas(16) uint32_t whist[8][kBins];  // per-wave digit counters -> per-wave digit starts
    alignas(16) uint32_t dstart[kBins];    // where digit d's run starts inside the tile
    uint32_t gbase[kBins];         // virtual slot of the digit's first key minus its start inside the tile (slots >= n: overflow scratch)
    uint32_t gbase2[kBins];        //   ... of the part of the run behind `split` (a run that crosses its primary region's end)
    uint32_t split[kBins];         // tile position from which gbase2 applies (0xFFFFFFFF: nowhere)
    uint32_t scan_tmp[8];
    uint32_t acc[kBins];           // keys of top byte d counted into the packed counters since their last flush
    uint32_t flags[2];             // by tile parity.  1: some run crosses its primary region's end; 4: flush the bucket counters behind this tile
#ifdef VRS_POOL_LAB_SMALL_HIST  // lab: what would three workgroups per CU buy (wrong counts: the plan refuses)
    uint32_t hist[64];
#else
    uint32_t hist[kPoolBuckets / 2];  // packed 16-bit counters of the 16384 buckets, as long as the workgroup lives
#endif
};

__device__ __forceinline__ void pool_flush_hist(uint32_t *s_hist, uint32_t *__restrict__ hist) {
#ifdef VRS_POOL_LAB_SMALL_HIST
    for (uint32_t w = threadIdx.x; w < 64u; w += 512u) {
#else
    for (uint32_t w = threadIdx.x; w < kPoolBuckets / 2u; w += 512u) {
#endif
        const uint32_t x = s_hist[w];
        if (x) {
            s_hist[w] = 0;
            // buckets 2 w and 2 w + 1 in one 64-bit add (neither count reaches 2^32: no carry between them)
            const unsigned long long v = (static_cast<unsigned long long>(x >> 16) << 32) | (x & 0xFFFFu);
            __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(hist) + w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// One tile of 8192 keys by a workgroup of 512 threads x 16 keys: rank inside the wave with returning LDS atomics (RANK_ATOMIC:
// the library runs the hybrid forms only after the lane-order self-test passed), count the buckets, reserve, re-bucket through
// LDS, write every top byte's run to the slots it reserved -- scatter_chunk's job (vrs_device.hpp), laid out for a PERSISTENT
// workgroup of which a CU holds only two (the bucket counters take 32 KB of LDS each), so that every barrier and every exposed
// latency counts:
//  * key[]: the tile's keys already in registers, in ANY assignment to threads (the pass is not stable); next[] / next_kin: the
//    keys of the workgroup's NEXT tile are asked for in the middle of this one -- behind the reservation atomic, in front of
//    the write-out's stores (loads and stores retire on one in-order counter: a load issued behind the stores would wait for
//    all of them) -- as four 16-byte loads per thread;
//  * three barriers instead of scatter_chunk's five: EVERY wave reads all eight per-wave count rows (lane l: digits 4l..4l+3,
//    one ds_read_b128 per row) and computes the digit prefix and its own starts redundantly, so no thread hands a scan
//    result to another; a wave's row is zeroed again by the wave itself at the end of the tile, off the critical path;
//  * the reservation atomic of digit 4l + w is issued by lane l of wave w < 4 as soon as the digit's total is known, before
//    the prefix is scanned.
// parity: tiles alternate between two flag words (a slow wave may still read the previous tile's).
template <bool FULL, bool PREFETCH>
__device__ __forceinline__ uint32_t pool_tile_a(PoolSmem &sm, uint32_t (&key)[16], uint32_t (&next)[16], const uint32_t *__restrict__ next_kin,
                                            uint32_t *__restrict__ kout,
                                            uint32_t *__restrict__ overflow, uint32_t overflow_last, uint32_t n_virt, uint32_t valid, uint32_t shift,
                                            uint32_t key_base, uint32_t *__restrict__ cursor_row, const PoolPlan *__restrict__ pool,
                                            uint32_t s_out, uint32_t parity, uint32_t &over, uint32_t *fail_word) {
    constexpr int ITEMS = 16, WAVES = 8;
    constexpr uint32_t THREADS = WAVES * 64;
    // (the thread index opaque to the optimiser: everything derived from it -- sixteen tile positions, LDS addresses, the
    // digit this lane owns -- is then recomputed per tile, a few VALU instructions, instead of being kept alive across the
    // persistent loop in registers the tile needs: the compiler spills such invariants, and a reload is a VMEM operation
    // that waits behind every load and store in flight)
    const uint32_t tid = opaque(threadIdx.x), lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t dshift = kMsdBits - 8u;  // bucket -> top byte
    POOL_MARK(0);
    if (tid == 0) sm.flags[parity] = 0;  // (read behind this tile's last barrier; the other word may still be read by a slow wave)

    // ---- rank inside the wave (this wave's own row of counters, zero since the wave's previous tile); count the bucket
    uint32_t rank2[ITEMS / 2];  // two tile positions (< 8192) per register: the registers decide between 128 and spills
    uint32_t *my_hist = sm.whist[wave];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        uint32_t rank_i;
        const uint32_t raw = (key[i] - key_base) >> shift;
        // ragged tile (wave-striped 4-byte loads, pool_ragged_tile_a): positions >= valid hold the padding key
        const bool real = FULL || wave * (ITEMS * 64) + i * 64 + lane < valid;
        if (real) over |= raw >> kMsdBits;
        const uint32_t b = min(raw, kPoolBuckets - 1u);
        const uint32_t d = b >> dshift;
#ifdef VRS_POOL_LAB_SMALL_HIST
        const uint32_t hw = (b >> 1) & 63u, hv = 1u << ((b & 1u) << 4);
#else
        const uint32_t hw = b >> 1, hv = 1u << ((b & 1u) << 4);
#endif
        const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
        if (__ballot(d == d0) == ~0ull) {  // wave-uniform: one add by lane 0 instead of 64 on one counter (scatter_chunk's skew guard)
            uint32_t old = 0;
            if (lane == 0u) old = __hip_atomic_fetch_add(&my_hist[d0], 64u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            rank_i = __builtin_amdgcn_readfirstlane(old) + lane;
            const uint32_t b0 = __builtin_amdgcn_readfirstlane(b);
            if (__ballot(real && b == b0) == ~0ull) {  // ... and one bucket: sorted or constant keys
                if (lane == 0u) __hip_atomic_fetch_add(&sm.hist[__builtin_amdgcn_readfirstlane(hw)], 64u << ((b0 & 1u) << 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (real) {
                __hip_atomic_fetch_add(&sm.hist[hw], hv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
            rank_i = __hip_atomic_fetch_add(&my_hist[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (real) __hip_atomic_fetch_add(&sm.hist[hw], hv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (i & 1) rank2[i >> 1] |= rank_i << 16; else rank2[i >> 1] = rank_i;
    }
    POOL_MARK(2);
    __syncthreads();
    POOL_MARK(3);

    // ---- waves 0-3: lane l of wave w owns digit D = 64 w + l (consecutive lanes, consecutive cursors: one wave's atomic touches
    // two cache lines of the row -- with the digits dealt four to a lane it touched eight, every line took four times the
    // atomics, and the pass 60 % longer).  Its total and its reservation first: the scan below hides part of the round trip.
    const uint32_t D = tid & 255u;
    uint32_t cnt = 0, reserved = 0, rb = 0, rc = 0, ob = 0, oc = 0;
    if (wave < 4u) {  // wave-uniform
        uint32_t dtot = 0;
#pragma unroll
        for (int v = 0; v < WAVES; ++v) dtot += sm.whist[v][D];
        cnt = dtot - ((!FULL && D == 255u) ? kPoolTile - valid : 0u);  // padding keys take no room
        // ONE atomic add in the L2 this workgroup's CU sits behind: every workgroup that adds to this row sits behind the same
        if (cnt) reserved = __hip_atomic_fetch_add(cursor_row + D, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // this digit's region: four L2-resident words, in flight beside the atomic
        rb = pool->base[s_out][D];
        rc = pool->cap[s_out][D];
        ob = pool->obase[s_out][D];
        oc = pool->ocap[s_out][D];
        // keys counted into the 16-bit bucket counters of this top byte since the last flush: none may pass 65535
        const uint32_t acc = sm.acc[D] + dtot;
        sm.acc[D] = acc;
        if (acc > kPoolFlushAt) atomicOr(&sm.flags[parity], 4u);
    }
    // ---- every wave: lane l takes digits 4l .. 4l + 3 -- their totals, their starts inside the tile, this wave's own starts
    uint32_t tot[4] = {0, 0, 0, 0}, mine[4] = {0, 0, 0, 0};
#pragma unroll
    for (int v = 0; v < WAVES; ++v) {
        const uint4 c = reinterpret_cast<const uint4 *>(sm.whist[v])[lane];
        const uint32_t m = static_cast<uint32_t>(v) < wave ? ~0u : 0u;  // wave-uniform
        mine[0] += c.x & m;
        mine[1] += c.y & m;
        mine[2] += c.z & m;
        mine[3] += c.w & m;
        tot[0] += c.x;
        tot[1] += c.y;
        tot[2] += c.z;
        tot[3] += c.w;
        if (v == 3) __builtin_amdgcn_sched_barrier(0);  // two batches of four reads: eight at once cost 32 registers beside keys and ranks
    }
    __builtin_amdgcn_sched_barrier(0);  // (the prefetch's 16 registers only once the count rows' are free)
    if constexpr (PREFETCH) {
        const uint4 *nv = reinterpret_cast<const uint4 *>(next_kin);
#pragma unroll
        for (int i = 0; i < ITEMS / 4; ++i) {
            const uint4 q = nv[i * THREADS + tid];
            next[4 * i] = q.x;
            next[4 * i + 1] = q.y;
            next[4 * i + 2] = q.z;
            next[4 * i + 3] = q.w;
        }
    }
    // exclusive prefix of the digit totals over the lanes (every wave computes the same)
    const uint32_t four = tot[0] + tot[1] + tot[2] + tot[3];
    uint32_t incl = four;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if (lane >= static_cast<uint32_t>(o)) incl += t;
    }
    uint32_t start[4];
    start[0] = incl - four;
    start[1] = start[0] + tot[0];
    start[2] = start[1] + tot[1];
    start[3] = start[2] + tot[2];
    if (wave == 0u) reinterpret_cast<uint4 *>(sm.dstart)[lane] = make_uint4(start[0], start[1], start[2], start[3]);
    POOL_MARK(4);
    __syncthreads();  // every wave has read every row's counts
    POOL_MARK(5);

    // ---- this wave's starts into its own row, then (same wave: the LDS keeps its operations in order) the re-bucketing
    reinterpret_cast<uint4 *>(my_hist)[lane] = make_uint4(start[0] + mine[0], start[1] + mine[1], start[2] + mine[2], start[3] + mine[3]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // (the keys opaque to the optimiser from here on: it would otherwise keep the sixteen digits of the ranking phase alive across
    // the scan -- registers the scan's count rows and the prefetch need -- instead of recomputing them, two VALU instructions each)
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) key[i] = opaque(key[i]);
#pragma unroll
    for (int j = 0; j < ITEMS / 2; ++j) {
        const uint32_t s0 = my_hist[min((key[2 * j] - key_base) >> shift, kPoolBuckets - 1u) >> dshift];
        const uint32_t s1 = my_hist[min((key[2 * j + 1] - key_base) >> shift, kPoolBuckets - 1u) >> dshift];
        rank2[j] += s0 | (s1 << 16);  // (no carry: every position is below 8192)
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) sm.keys[(i & 1) ? rank2[i >> 1] >> 16 : rank2[i >> 1] & 0xFFFFu] = key[i];
    if (wave < 4u) {
        const uint32_t excl = sm.dstart[D];
        // run [reserved, reserved + cnt) of the region's position space: positions below rc are primary slots rb + position,
        // the others overflow slots (virtual n_virt + ob + position - rc)
        const uint32_t end = reserved + cnt;
        uint32_t g, g2 = 0, sp = 0xFFFFFFFFu;
        bool bad = false;
        if (end <= rc) {
            g = rb + reserved - excl;
        } else if (reserved >= rc) {
            bad = end - rc > oc;
            g = n_virt + ob + (reserved - rc) - excl;
        } else {  // the one run of this region that crosses the end of its primary part
            bad = end - rc > oc;
            sp = excl + (rc - reserved);
            g = rb + reserved - excl;
            g2 = n_virt + ob - sp;
            atomicOr(&sm.flags[parity], 1u);
        }
        // a region out of room: the sort is refused (the keys of this run still go somewhere inside the scratch, see below)
        if (cnt && bad) __hip_atomic_fetch_or(fail_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sm.gbase[D] = g;
        sm.gbase2[D] = g2;
        sm.split[D] = sp;
    }
    POOL_MARK(6);
    __syncthreads();
    POOL_MARK(7);
    const uint32_t flags = sm.flags[parity];  // workgroup-uniform

    // ---- write out: tile position q goes to slot gbase[digit] + q; reads batched before stores.  EVERY path through here issues
    // the same 16 stores (a run beyond its region's room -- the sort is refused then -- is clamped into the scratch instead of
    // skipped): the compiler's wait for the prefetched keys of the next tile is "all but the 16 youngest operations" only if
    // no path has fewer.
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) key[i] = sm.keys[i * THREADS + tid];
    uint32_t dst[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) dst[i] = sm.gbase[min((key[i] - key_base) >> shift, kPoolBuckets - 1u) >> dshift] + (i * THREADS + tid);
    if (flags & 1u) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t d = min((key[i] - key_base) >> shift, kPoolBuckets - 1u) >> dshift;
            if (i * THREADS + tid >= sm.split[d]) dst[i] = sm.gbase2[d] + (i * THREADS + tid);
        }
    }
    // this wave's row of counters, zero for the wave's next tile (nobody else reads it before that tile's first barrier)
    {
        const uint32_t z = tid >> 10;  // zero, made here (a constant zero vector would be one more invariant held across the loop -- and spilled)
        reinterpret_cast<uint4 *>(my_hist)[lane] = make_uint4(z, z, z, z);
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        // (a ragged tile's padding keys rank last: their slots are the first ones behind the last key's, or clamped -- harmless,
        // nobody reads them)
        const bool primary = dst[i] < n_virt;
        uint32_t *p = primary ? kout + dst[i] : overflow + min(dst[i] - n_virt, overflow_last);
        if (FULL || i * THREADS + tid < valid) *p = key[i];
    }
    POOL_MARK(8);
    return flags;
}

// out of line: at most eight tiles of a sort are ragged, and their index arithmetic must not sit in the registers of the loop.
// Returns "a key lies outside the probed range".
__device__ __attribute__((noinline)) uint32_t pool_ragged_tile_a(PoolSmem &sm, const uint32_t *__restrict__ kin, uint32_t *__restrict__ kout,
                                                                 uint32_t *__restrict__ overflow, uint32_t overflow_last, uint32_t n_virt,
                                                                 uint32_t valid, uint32_t shift, uint32_t key_base,
                                                                 uint32_t *__restrict__ cursor_row, const PoolPlan *__restrict__ pool,
                                                                 uint32_t s_out, uint32_t parity, uint32_t *fail_word) {
    const uint32_t seg = (threadIdx.x >> 6) * 1024u + (threadIdx.x & 63u);
    uint32_t key[16], none[16], over = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t idx = seg + i * 64;
        const uint32_t k = kin[idx < valid ? idx : valid - 1u];
        key[i] = idx < valid ? k : key_base - 1u;  // the padding key: top byte 255 of the range, the highest tile positions
    }
    // (a flush this tile asks for is the kernel's final one)
    (void)pool_tile_a<false, false>(sm, key, none, nullptr, kout, overflow, overflow_last, n_virt, valid, shift, key_base, cursor_row, pool, s_out,
                                    parity, over, fail_word);
    return over;
}

// grid = 8 * wgs_per_stream workgroups; workgroup b walks tiles (b >> 3), (b >> 3) + wgs_per_stream, ... of slice b & 7 -- the
// slice whose keys the sample counted for the regions of row b & 7, and (observed placement: block b on XCD b % 8) the row
// whose cursors live in this CU's L2.  Neither is assumed: the row a workgroup ADDS TO is chosen by the XCC it finds itself
// on, so that a row's L2-local atomics always meet in one L2, whichever slice the workgroup reads.
__global__ __launch_bounds__(512, 4) void pool_pass_a_kernel(const uint32_t *__restrict__ keys_in, uint32_t *__restrict__ keys_out,
                                                             uint32_t *__restrict__ overflow, uint32_t n, uint32_t key_base,
                                                             PoolStreams ps, PoolPlan *__restrict__ pool, MsdPlan *__restrict__ msd,
                                                             uint32_t *__restrict__ hist, unsigned long long xcc_map,
                                                             uint32_t wgs_per_stream, int misplace, uint32_t overflow_capacity) {
    __shared__ PoolSmem sm;
    if (pool->armed == 0u) return;  // uniform: the sample kernel did not lay regions out (key range below 27 bits)
    const uint32_t tid = threadIdx.x;
    const uint32_t r = blockIdx.x >> 3;
    // misplace (test hook): odd rows of workgroups read the neighbouring slice
    const uint32_t s_in = (blockIdx.x + (misplace ? (r & 1u) : 0u)) & 7u;
    const uint32_t my_xcc = xcc_id();
    uint32_t s_out = blockIdx.x & 7u;
    if (xcc_of(xcc_map, s_out) != my_xcc) {  // not where block b % 8 was observed to run: the row of the L2 this CU does sit behind
        s_out = 8u;
        for (uint32_t x = 0; x < 8u; ++x)
            if (s_out == 8u && xcc_of(xcc_map, x) == my_xcc) s_out = x;
    }
    if (s_out == 8u) {  // behind an L2 the probe never saw: no row is safe to add to -- the sort is refused
        if (tid == 0) __hip_atomic_fetch_or(&pool->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    for (uint32_t w = tid; w < sizeof(sm.hist) / 4u; w += 512u) sm.hist[w] = 0;  // (the first tile's first barrier orders this)
#ifdef VRS_POOL_LAB_MARKS
    if (tid == 0) {
        for (int k = 0; k < 12; ++k) sm.marks[k] = 0;
        sm.mark_last = __builtin_readcyclecounter();
    }
#endif
    if (tid < kBins) sm.acc[tid] = 0;
    for (uint32_t w = tid; w < 8u * kBins; w += 512u) (&sm.whist[0][0])[w] = 0;  // (afterwards every wave re-zeroes its own row, tile by tile)
    const uint32_t shift = pool->shift;
    const uint32_t len = ps.len[s_in];
    const uint32_t tiles = (len + kPoolTile - 1u) / kPoolTile;
    const uint32_t *kin = keys_in + ps.start[s_in];
    uint32_t *cursor_row = &msd->cursor_a[s_out][0];
    uint32_t over = 0, parity = 0;
    // Full tiles in a software pipeline: the keys of the workgroup's next tile are in flight while this one is written out.
    // Two copies of the tile body, ka -> kb and kb -> ka, so that no register is copied at the seam (a copy would wait for
    // the loads AND, with them, for every store issued since).  The last tile prefetches the slice's last full tile again (one
    // tile for all 64 workgroups of the slice: served by L2) -- a path with fewer loads would weaken every wait behind it.
    const uint32_t full = len / kPoolTile;
    const uint32_t overflow_last = overflow_capacity - 1u;
    uint32_t ka[16], kb[16];
    if (r < full) {
        const uint4 *v = reinterpret_cast<const uint4 *>(kin + r * kPoolTile);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 q = v[i * 512 + tid];
            ka[4 * i] = q.x;
            ka[4 * i + 1] = q.y;
            ka[4 * i + 2] = q.z;
            ka[4 * i + 3] = q.w;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the loop is entered with nothing in flight, like its back edge minus the stores
    }
    __syncthreads();  // the counters are zero
    const auto flush_if = [&](uint32_t flags) {
        if (flags & 4u) {  // a 16-bit bucket counter could overflow in the next tile (thousands of keys per tile in one bucket)
            __syncthreads();
            pool_flush_hist(sm.hist, hist);
            if (tid < kBins) sm.acc[tid] = 0;
            __syncthreads();  // the next tile counts at once
        }
    };
    for (uint32_t i = r; i < full;) {
        uint32_t nxt = min(i + wgs_per_stream, full - 1u);
        flush_if(pool_tile_a<true, true>(sm, ka, kb, kin + nxt * kPoolTile, keys_out, overflow, overflow_last, n, kPoolTile, shift, key_base, cursor_row,
                                         pool, s_out, parity, over, &pool->fail));
        i += wgs_per_stream;
        parity ^= 1u;
        if (i >= full) break;
        nxt = min(i + wgs_per_stream, full - 1u);
        flush_if(pool_tile_a<true, true>(sm, kb, ka, kin + nxt * kPoolTile, keys_out, overflow, overflow_last, n, kPoolTile, shift, key_base, cursor_row,
                                         pool, s_out, parity, over, &pool->fail));
        i += wgs_per_stream;
        parity ^= 1u;
    }
    // the slice's ragged last tile, by the workgroup whose turn it is
    if (full < tiles && full % wgs_per_stream == r)
        over |= pool_ragged_tile_a(sm, kin + full * kPoolTile, keys_out, overflow, overflow_last, n, len - full * kPoolTile, shift, key_base, cursor_row,
                                   pool, s_out, parity, &pool->fail);
    __syncthreads();
    POOL_MARK(9);
    pool_flush_hist(sm.hist, hist);
#ifdef VRS_POOL_LAB_MARKS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    POOL_MARK(10);
    if (tid == 0)
        for (int k = 0; k < 12; ++k) g_pool_marks[static_cast<size_t>(blockIdx.x) * 12 + k] = sm.marks[k];
#endif
    if (__syncthreads_or(static_cast<int>(over)) && tid == 0)  // a key above the probed range (or below the promised floor)
        __hip_atomic_fetch_or(&pool->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// The plan, once the first pass has run: ONE workgroup of 1024 threads (msd_plan_kernel's job in the counted form).
__global__ __launch_bounds__(1024) void pool_plan_kernel(uint32_t *__restrict__ counts, MsdPlan *__restrict__ msd,
                                                        PoolPlan *__restrict__ pool, OnesweepPlanHead *__restrict__ dev_head,
                                                        OnesweepPlanHead *host_head, uint32_t stamp, uint32_t n,
                                                        uint32_t tiles_b_cap, uint32_t local_cap, uint32_t *host_log) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_start[kBins + 1];      // where top byte a starts in the sorted order
    __shared__ uint32_t s_tiles[8][kBins];       // [XCD x][entry e = 8 k + s]: tiles of slice s's share of top byte x + 8 k
    __shared__ uint32_t s_max, s_tiles_b, s_bad, s_ok;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    constexpr uint32_t kPer = kPoolBuckets / 1024u;  // 16 buckets per thread
    const uint32_t shift = pool->shift, armed = pool->armed, failed = pool->fail;
    uint32_t c[kPer], cur[8];
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
#pragma unroll
    for (int s = 0; s < 8; ++s) cur[s] = tid < kBins ? msd->cursor_a[s][tid] : 0u;
    if (tid == 0) {
        s_max = 0;
        s_tiles_b = 0;
        s_bad = 0;
    }
    // (1) exclusive prefix over the 16384 buckets; the histogram is left zeroed for the next sort
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
            if ((b & (kMsdSub - 1u)) == 0u) s_start[b >> kMsdSubBits] = run;
            run += c[j];
        }
        uint4 *vb = reinterpret_cast<uint4 *>(msd->base + tid * kPer);
#pragma unroll
        for (uint32_t j = 0; j < kPer / 4u; ++j) vb[j] = make_uint4(start[4 * j], start[4 * j + 1], start[4 * j + 2], start[4 * j + 3]);
    }
    if (tid == 1023u) {
        msd->base[kPoolBuckets] = run;
        s_start[kBins] = run;
        if (run != n) s_bad = 1;  // keys the first pass did not count: it dropped a tile (a region overflowed) or did not run
    }
    __syncthreads();
    // (2) the second pass's tiles: top byte a = tid, its eight shares; what the cursors say must be what the histogram says
    if (tid < kBins) {
        uint32_t keys_a = 0;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            keys_a += cur[s];
            s_tiles[tid & 7u][(tid >> 3) * 8u + s] = (cur[s] + kPoolTile - 1u) / kPoolTile;
        }
        if (keys_a != s_start[tid + 1] - s_start[tid]) s_bad = 1;
    }
    __syncthreads();
    if (wave < 8u) {  // wave x: the exclusive prefix of XCD x's 256 entries, four per lane
        uint32_t t[4], tot = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[j] = s_tiles[wave][4 * lane + j];
            tot += t[j];
        }
        uint32_t inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(inc, o);
            if (lane >= static_cast<uint32_t>(o)) inc += u;
        }
        uint32_t a = inc - tot;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pool->tiles_b[wave][4 * lane + j] = a;
            a += t[j];
        }
        if (lane == 63u) {
            pool->tiles_b[wave][kBins] = a;
            atomicMax(&s_tiles_b, a);
        }
    }
    __syncthreads();
    if (tid == 0) {
        s_ok = (armed != 0u && failed == 0u && s_bad == 0u && shift >= kPoolMinShift && shift <= kPoolMaxShift && s_max <= local_cap &&
                s_tiles_b <= tiles_b_cap)
                   ? 1u
                   : 0u;
        pool->fail = 0;  // re-armed for the next sort
        msd->shift = shift;
        msd->ok = s_ok;
        msd->sub_bits = kMsdSubBits;
        dev_head->msd_ok = s_ok;
        dev_head->msd_tiles_b = s_tiles_b;
        dev_head->msd_max_bucket = s_max;
        dev_head->lsd_missing = 1u;
        if (host_head) {
            __hip_atomic_store(&host_head->lsd_missing, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&host_head->msd_ok, s_ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&host_head->msd_tiles_b, s_tiles_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&host_head->msd_max_bucket, s_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (host_log)
                __hip_atomic_store(&host_log[stamp & (kMsdLogWords - 1u)], (stamp << 1) | s_ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
            __hip_atomic_store(&host_head->ready, stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Second pass: block b -> XCD b % 8, which walks the shares (top byte x + 8 k, slice s) in entry order e = 8 k + s; every
// bucket's keys go to its final range by reservation on the bucket's cursor (MsdPlan::cursor_b, in this XCD's L2; a tile off
// its XCD takes room from the range's END with a device-scope atomic: StreamReserve, vrs_device.hpp).
__global__ __launch_bounds__(512, 4) void pool_pass_b_kernel(const uint32_t *__restrict__ regions, const uint32_t *__restrict__ overflow,
                                                             uint32_t *__restrict__ keys_out, MsdPlan *__restrict__ msd,
                                                             const PoolPlan *__restrict__ pool, unsigned long long xcc_map,
                                                             uint32_t key_base) {
    __shared__ ChunkSmem<uint32_t, 16, 8, false> sm;
    const uint32_t x = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const uint32_t *pt = pool->tiles_b[x];
    if (msd->ok == 0u || j >= pt[kBins]) return;  // uniform (enqueued before the plan was known: it may have said no)
    uint32_t e = 0;  // the entry whose tiles contain j: largest e with pt[e] <= j
#pragma unroll
    for (uint32_t step = 128; step >= 1; step >>= 1)
        if (pt[e + step] <= j) e += step;
    const uint32_t a = x + 8u * (e >> 3), s = e & 7u, i = j - pt[e];
    const uint32_t keys_sa = msd->cursor_a[s][a];               // keys of this share (the first pass's cursor)
    const uint32_t prim = min(keys_sa, pool->cap[s][a]);        // ... of them in the primary region, the rest in the overflow region
    const uint32_t done = i * kPoolTile;
    const uint32_t valid = min(kPoolTile, keys_sa - done);
    const uint32_t split = prim > done ? prim - done : 0u;      // leading tile positions that lie in the primary region
    const uint32_t *kin0 = regions + pool->base[s][a] + done;
    const uint32_t *kin1 = overflow + pool->obase[s][a] + (static_cast<int64_t>(done) - static_cast<int64_t>(prim));
    BitsDigit dg{msd->shift, kMsdSub - 1u, key_base};
    StreamReserve lb;
    const uint32_t b = (a << kMsdSubBits) + min(threadIdx.x & 255u, kMsdSub - 1u);
    lb.foreign = xcc_id() != xcc_of(xcc_map, x);
    lb.cursor = &msd->cursor_b[b];
    lb.back = &msd->back_b[b];
    lb.pad_keys = (threadIdx.x & 255u) == dg(dg.template pad<uint32_t>()) ? kPoolTile - valid : 0u;
    lb.seed = msd->base[b];
    if (lb.foreign) lb.region_len = msd->base[b + 1u] - lb.seed;
    uint32_t unused = 0;
    if (valid == kPoolTile)
        scatter_chunk<uint32_t, 16, 8, false, RANK_ATOMIC, true, BitsDigit, StreamReserve, true>(sm, kin0, nullptr, keys_out, nullptr, valid, dg, unused, lb, kin1, split);
    else
        scatter_chunk<uint32_t, 16, 8, false, RANK_ATOMIC, false, BitsDigit, StreamReserve, true>(sm, kin0, nullptr, keys_out, nullptr, valid, dg, unused, lb, kin1, split);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host side
PoolStreams pool_streams(uint32_t n) {
    PoolStreams ps{};
    ps.tiles_total = (n + kPoolTile - 1u) / kPoolTile;
    ps.tiles_per_stream = std::max<uint32_t>((ps.tiles_total + 7u) / 8u, 1u);
    for (uint32_t s = 0; s < 8u; ++s) {
        const uint64_t a = std::min<uint64_t>(static_cast<uint64_t>(s) * ps.tiles_per_stream * kPoolTile, n);
        const uint64_t b = std::min<uint64_t>(static_cast<uint64_t>(s + 1u) * ps.tiles_per_stream * kPoolTile, n);
        ps.start[s] = static_cast<uint32_t>(a);
        ps.len[s] = static_cast<uint32_t>(b - a);
        const uint32_t full = ps.len[s] / kPoolTile, rest = ps.len[s] % kPoolTile;
        ps.sampled[s] = full * kPoolSampleKeys + std::min(rest, kPoolSampleKeys);
    }
    return ps;
}

uint32_t pool_overflow_capacity(uint32_t n) {
    // sum over the 2048 regions of [six deviations of an estimate scaled up 32-fold + rounding + floor], bounded by
    // Cauchy-Schwarz: sum sqrt(r e_i) <= sqrt(2048 r n); r is 32 but for the slices' ragged last tiles
    const double room = 6.0 * std::sqrt(2048.0 * 33.0 * static_cast<double>(n)) + 2048.0 * (kPoolRoomFloor + 64.0);
    return static_cast<uint32_t>(std::min<double>(room, 1u << 28)) & ~31u;
}

uint32_t pool_tiles_b_cap(uint32_t n, bool blind) {
    const uint32_t tiles = (n + kPoolTile - 1u) / kPoolTile, even = (tiles + 7u) / 8u;
    // an XCD walks 32 top bytes x 8 shares, each rounded up to whole tiles; blind, the grid IS the cap: little slack
    return (blind ? even + even / 16u : even + even / 4u) + 256u + 40u;
}

hipError_t launch_pool_sample(hipStream_t stream, const uint32_t *keys, uint32_t n, uint32_t key_base, const PoolStreams &ps,
                              PoolPlan *pool, uint32_t overflow_capacity, LaunchEvents ev) {
    if (n == 0) return hipErrorInvalidValue;
    const uint32_t grid = (ps.tiles_total + kPoolSampleTiles - 1u) / kPoolSampleTiles;
    VRS_LAUNCH(pool_sample_kernel, dim3(grid), dim3(256), stream, ev, keys, n, key_base, ps, pool, overflow_capacity);
    return hipGetLastError();
}

hipError_t launch_pool_pass_a(hipStream_t stream, const uint32_t *keys_in, uint32_t *keys_out, uint32_t *overflow, uint32_t n,
                              uint32_t key_base, const PoolStreams &ps, PoolPlan *pool, MsdPlan *msd, uint32_t *hist,
                              unsigned long long xcc_map, int compute_units, bool misplace, uint32_t overflow_capacity, LaunchEvents ev) {
    // two 512-thread workgroups per CU, an eighth of them per slice -- no more than the slice has tiles
#ifndef VRS_POOL_LAB_WGS_PER_CU
#define VRS_POOL_LAB_WGS_PER_CU 2
#endif
    const uint32_t resident = std::max<uint32_t>(static_cast<uint32_t>(compute_units) * VRS_POOL_LAB_WGS_PER_CU / 8u, 1u);
    const uint32_t wgs = std::min(resident, ps.tiles_per_stream);
    VRS_LAUNCH(pool_pass_a_kernel, dim3(8u * wgs), dim3(512), stream, ev, keys_in, keys_out, overflow, n, key_base, ps, pool, msd, hist,
               xcc_map, wgs, misplace ? 1 : 0, overflow_capacity);
    return hipGetLastError();
}

hipError_t launch_pool_plan(hipStream_t stream, uint32_t *hist, MsdPlan *msd, PoolPlan *pool, OnesweepPlanHead *dev_head,
                            OnesweepPlanHead *host_head, uint32_t stamp, uint32_t n, uint32_t tiles_b_cap, uint32_t local_cap,
                            uint32_t *host_log) {
    hipLaunchKernelGGL(pool_plan_kernel, dim3(1), dim3(1024), 0, stream, hist, msd, pool, dev_head, host_head, stamp, n, tiles_b_cap,
                       local_cap, host_log);
    return hipGetLastError();
}

hipError_t launch_pool_pass_b(hipStream_t stream, const uint32_t *regions, const uint32_t *overflow, uint32_t *keys_out, MsdPlan *msd,
                              const PoolPlan *pool, uint32_t tiles_b, unsigned long long xcc_map, uint32_t key_base, LaunchEvents ev) {
    if (tiles_b == 0) return hipSuccess;
    VRS_LAUNCH(pool_pass_b_kernel, dim3(8u * tiles_b), dim3(512), stream, ev, regions, overflow, keys_out, msd, pool, xcc_map, key_base);
    return hipGetLastError();
}

}  // namespace vrs
