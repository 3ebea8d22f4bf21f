// vrs_msd_pool.hip -- the hybrid form of the one-call sort WITHOUT a counting read (bare uint32 keys): 24 bytes per key.
//
// The reference reads the keys once per pass just to count them (multi_radixsort_histograms.comp:42-50); the counted hybrid
// form (vrs_msd_hybrid.hip, K5b) still reads them once for that.  Here they are not read for counting at all, and no pass waits
// for a histogram either:
//
//   pool_sample_kernel      1/32 of the input (the first 256 keys of every 8192-key tile): probes the key range (bucket shift)
//                           and counts the top byte per input slice; pool_layout_kernel (one workgroup behind it) lays out, for every
//                           (slice, top byte), a PRIMARY region of the partner buffer sized by the estimate (the estimates of a slice sum to its
//                           length, so the regions tile the n-key buffer) and an OVERFLOW region in context scratch of six
//                           standard deviations of that estimate;
//   pool_pass_a_kernel      first MSD pass (top 8 bits of the key range): a tile reserves its place in (slice, top byte)'s region
//                           with ONE atomic add on the region's cursor, in the L2 of the XCD that runs the slice (StreamReserve's
//                           idea, vrs_device.hpp) -- positions below the region's capacity are primary slots, the rest overflow;
//   pool_plan_kernel        ONE workgroup: top-byte totals are the cursors' sums, exact -- where every top byte starts in the
//                           sorted order, the second pass's tile tables, verdict 1;
//   pool_pass_b_kernel      second MSD pass (the next 6 bits) WITHOUT any global offset: a tile groups its own 8192 keys by the 6
//                           bits in LDS and writes them back to the slots it read them from, 16 bytes per lane, and leaves a
//                           row of 64 (offset, count) pairs.  A bucket is then a set of RUNS, one per tile of its top byte;
//   pool_runs_kernel        one workgroup per top byte: rows -> run descriptors per bucket, exact bucket starts (MsdPlan::base),
//                           the largest bucket.  Up to here the caller's buffer has not been written;
//   pool_local_sort_kernel  one workgroup per bucket -- every one derives verdict 2 from the same three words, workgroup 0 tells
//                           the host: gathers the bucket's runs (about 48 of about 128 keys), sorts the keys by
//                           their low 18 bits inside LDS (lean_sort_body, vrs_local_sort.hpp) and stores the bucket at its final
//                           place in the caller's buffer.
//
// Nothing here is assumed about the data: a sample that misjudges a region (keys whose distribution changes inside a tile
// with the tile's period, say) makes the first pass flag the sort, a verdict refuse, and the caller run the counted form on the
// untouched input.  Neither MSD pass is stable (arrival order inside a region, any order inside a run); bare keys do not care.
#include "vrs_local_sort.hpp"

#include <algorithm>
#include <cmath>

namespace vrs {

namespace {

constexpr uint32_t kPoolBuckets = kMsdBucketCount;        // 16384
constexpr uint32_t kPoolMinShift = 13, kPoolMaxShift = 18;  // a 27 ... 32-bit key range (the counted form's rule)
constexpr uint32_t kPoolFlushAt = 65535u - kPoolTile;     // a 16-bit counter may take one more tile below this
constexpr float kPoolSigmas = 6.0f;                       // overflow room, in standard deviations of the region's estimate
constexpr uint32_t kPoolRoomFloor = 320;

__device__ __forceinline__ uint32_t xcc_of(unsigned long long xcc_map, uint32_t x) {
    return static_cast<uint32_t>((xcc_map >> (8u * x)) & 0xFFu);
}

// ---------------------------------------------------------------------------------------------
// The sample.  Workgroup g takes tiles [32 g, 32 g + 32) of the input; wave w of it the tiles 32 g + w + 4 j.
__global__ __launch_bounds__(256) void pool_sample_kernel(const uint32_t *__restrict__ keys, uint32_t n, uint32_t key_base,
                                                          PoolStreams ps, PoolPlan *__restrict__ pool) {
    constexpr int kPerWave = kPoolSampleTiles / 4;        // tiles per wave
    constexpr int kLoads = kPoolSampleKeys / 64;          // 4-byte loads per lane and tile
    __shared__ uint32_t s_hist[2][256];
    __shared__ uint32_t s_or;
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
            k[j][c] = __builtin_nontemporal_load(keys + (idx < n ? idx : n - 1u));  // (read again by the first pass: nothing to keep)
        }
    }
    // The buckets are the top 14 bits of the key RANGE: every workgroup ORs the same strided 4096 keys (as the counted form's
    // counting read does) and derives the same shift; a key outside that range is flagged by the first pass.
    {
        // (512 keys, not the counting read's 4096: hundreds of workgroups asking the L2s for the same lines at the same time
        // is what this costs -- 4096 lines took 11 of the kernel's 30 us)
        const uint32_t samples = min(n, 512u);
        const uint64_t stride = n / max(samples, 1u);
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
    if (blockIdx.x == 0 && tid == 0) pool->shift = shift;  // (every workgroup derived the same)
}

// The regions, from the sample: ONE workgroup of 256 threads, thread d = top byte d of all eight slices.  (A kernel of its own:
// as the sample kernel's last workgroup -- a ticket, which needs every workgroup's adds acknowledged first -- it took 13 us.)
__global__ __launch_bounds__(256) void pool_layout_kernel(PoolStreams ps, PoolPlan *__restrict__ pool, uint32_t overflow_capacity) {
    __shared__ uint32_t s_wave[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t shift = pool->shift;
    // thread d: top byte d of all eight slices.  Regions are laid out top byte by top byte, slice by slice.
    uint32_t cap[8], room[8], caps = 0, rooms = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const uint32_t m = pool->sample[s][tid];
        pool->sample[s][tid] = 0;  // zero for the next sort
        const uint32_t len = ps.len[s], sampled = ps.sampled[s];
        // estimate: the slice's keys in proportion to the sample's, r = len / sampled for one sampled key (float arithmetic,
        // rounded DOWN by a hair: the estimates of a slice must sum to at most its length -- the primary regions tile the buffer)
        const float r = sampled ? static_cast<float>(len) / static_cast<float>(sampled) : 1.0f;
        const uint32_t est = min(static_cast<uint32_t>(static_cast<float>(m) * r * 0.999999f), len);
        cap[s] = est & ~31u;
        // the estimate scales m sampled keys up by r: its standard deviation is sqrt(r * est) -- of the TRUE expectation, which a
        // small m underrates (m = 0 is what a region of 5 r keys shows once in 150 sorts): one more sampled key under the root,
        // and a floor of 10 r keys
        const uint32_t dev = static_cast<uint32_t>(kPoolSigmas * sqrtf(r * (static_cast<float>(est) + r)));
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
        pool->fail = 0;  // re-armed here: the passes of THIS sort set it, its local sort reads it
        // a key range below 27 bits is left to the LSD passes, like the counted form does (vrs_msd_hybrid.hip, msd_plan_kernel)
        pool->armed = (shift >= kPoolMinShift && shift <= kPoolMaxShift && total_room <= overflow_capacity) ? 1u : 0u;
    }
}

// ---------------------------------------------------------------------------------------------
// First pass: scatter_chunk (vrs_device.hpp) with the regions as its offset source.
// (The digit is the counted form's: RadixDigit on the top 8 bits of the key range.  A key OUTSIDE the probed range lands in
// some region here; the second pass, which looks at every key anyway, flags it.)
// Reservation in sampled regions: StreamReserve's one atomic add per tile and digit (in the L2 every workgroup that adds to this
// row of cursors sits behind), then positions below the region's capacity are primary slots, the rest overflow slots.
static_assert(offsetof(PoolPlan, cap) == offsetof(PoolPlan, base) + 8 * 256 * 4 && offsetof(PoolPlan, obase) == offsetof(PoolPlan, base) + 16 * 256 * 4 &&
                  offsetof(PoolPlan, ocap) == offsetof(PoolPlan, base) + 24 * 256 * 4,
              "PoolReserve reads a region's four words 2048 words apart");
struct PoolReserve {
    static constexpr bool kEnabled = true;
    static constexpr bool kReserves = true;
    static constexpr bool kPool = true;
    bool foreign = false;            // (interface of StreamLookback: unused -- the row is chosen by the XCC the workgroup runs on)
    uint32_t recounted = 0;
    int index = 0;
    const void *stream_keys = nullptr;
    uint32_t done = 0;
    uint32_t seed = 0;
    uint32_t *cursor = nullptr;      // this thread's digit's cursor
    const uint32_t *region = nullptr;  // &PoolPlan::base[row][digit]: base, cap, obase, ocap are 8 * 256 words apart
    uint32_t pad_keys = 0;
    uint32_t n_virt = 0;             // virtual slots >= n_virt are overflow slots
    uint32_t *overflow = nullptr;
    uint32_t overflow_last = 0;
    uint32_t *gbase2 = nullptr, *split = nullptr, *flags = nullptr;  // LDS: see place()
    uint32_t *fail_word = nullptr;
    mutable uint32_t reserved = 0, cnt = 0, rb = 0, rc = 0, ob = 0, oc = 0;
    mutable bool reserved_yet = false;

    __device__ __forceinline__ void publish(uint32_t v) const {
        if (reserved_yet) return;  // the second call (the inclusive prefix) has nobody to tell
        reserved_yet = true;
        cnt = v - pad_keys;
        if (cnt) reserved = __hip_atomic_fetch_add(cursor, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        rb = region[0];  // four L2-resident words, in flight beside the atomic while the keys are re-bucketed
        rc = region[8 * 256];
        ob = region[16 * 256];
        oc = region[24 * 256];
    }
    __device__ __forceinline__ void fetch(int, uint32_t (&)[kLbBatch]) const {}
    __device__ __forceinline__ uint32_t resolve(uint32_t (&)[kLbBatch], bool &) const { return reserved; }
    // run [reserved, reserved + cnt) of the region's position space; excl: where the digit's run starts inside the tile
    __device__ __forceinline__ void place(uint32_t *gbase, uint32_t tid, uint32_t, uint32_t excl) const {
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
            atomicOr(flags, 1u);
        }
        // a region out of room: the sort is refused (the keys of this run still go somewhere inside the scratch: store())
        if (cnt && bad) __hip_atomic_fetch_or(fail_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gbase[tid] = g;
        gbase2[tid] = g2;
        split[tid] = sp;
    }
    template <typename K, int ITEMS, uint32_t THREADS, bool FULL, typename DG>
    __device__ __forceinline__ void store(const uint32_t *, const K (&key)[ITEMS], uint32_t (&dst)[ITEMS], K *kout, uint32_t valid, const DG &dg) const {
        const uint32_t tid = threadIdx.x;
        if (*flags & 1u) {  // workgroup-uniform
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                const uint32_t d = dg(key[i]);
                if (i * THREADS + tid >= split[d]) dst[i] = gbase2[d] + (i * THREADS + tid);
            }
        }
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            K *p = dst[i] < n_virt ? kout + dst[i] : overflow + min(dst[i] - n_virt, overflow_last);
            if (FULL || i * THREADS + tid < valid) *p = key[i];
        }
    }
};

// grid = 8 * tiles_per_stream workgroups; workgroup b takes tile b >> 3 of slice b & 7 -- the slice whose keys the sample counted
// for the regions of row b & 7, and (observed placement: block b on XCD b % 8) the row whose cursors live in this CU's L2.
// Neither is assumed: the row a workgroup ADDS TO is chosen by the XCC it finds itself on, so that a row's L2-local atomics
// always meet in one L2, whichever slice the workgroup reads.
__global__ __launch_bounds__(512, 4) void pool_pass_a_kernel(const uint32_t *__restrict__ keys_in, uint32_t *__restrict__ keys_out,
                                                             uint32_t *__restrict__ overflow, uint32_t n, uint32_t key_base,
                                                             PoolStreams ps, PoolPlan *__restrict__ pool, MsdPlan *__restrict__ msd,
                                                             unsigned long long xcc_map, int misplace, uint32_t overflow_capacity) {
    __shared__ ChunkSmem<uint32_t, 16, 8, false> sm;
    __shared__ uint32_t s_gbase2[kBins], s_split[kBins], s_flags;
    if (pool->armed == 0u) return;  // uniform: the sample kernel did not lay regions out (key range below 27 bits)
    const uint32_t i = blockIdx.x >> 3;
    // misplace (test hook): odd tiles are read from the neighbouring slice
    const uint32_t s_in = (blockIdx.x + (misplace ? (i & 1u) : 0u)) & 7u;
    const uint32_t len = ps.len[s_in];
    const uint32_t done = i * kPoolTile;
    if (done >= len) return;
    const uint32_t my_xcc = xcc_id();
    uint32_t s_out = blockIdx.x & 7u;
    if (xcc_of(xcc_map, s_out) != my_xcc) {  // not where block b % 8 was observed to run: the row of the L2 this CU does sit behind
        s_out = 8u;
        for (uint32_t x = 0; x < 8u; ++x)
            if (s_out == 8u && xcc_of(xcc_map, x) == my_xcc) s_out = x;
    }
    if (s_out == 8u) {  // behind an L2 the probe never saw: no row is safe to add to -- the sort is refused
        if (threadIdx.x == 0) __hip_atomic_fetch_or(&pool->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (threadIdx.x == 0) s_flags = 0;  // (set behind the chunk's barriers, read behind its last one)
    const uint32_t valid = min(kPoolTile, len - done);
    const uint32_t *kin = keys_in + ps.start[s_in] + done;
    const uint32_t d = threadIdx.x & 255u;
    RadixDigit<uint32_t> dg;
    dg.shift = pool->shift + (kMsdBits - 8u);
    dg.base = key_base;
    PoolReserve lb;
    lb.cursor = &msd->cursor_a[s_out][d];
    lb.region = &pool->base[s_out][d];
    lb.pad_keys = d == 255u ? kPoolTile - valid : 0u;
    lb.n_virt = n;
    lb.overflow = overflow;
    lb.overflow_last = overflow_capacity - 1u;
    lb.gbase2 = s_gbase2;
    lb.split = s_split;
    lb.flags = &s_flags;
    lb.fail_word = &pool->fail;
    uint32_t unused = 0;
    if (valid == kPoolTile)
        scatter_chunk<uint32_t, 16, 8, false, RANK_ATOMIC, true>(sm, kin, nullptr, keys_out, nullptr, valid, dg, unused, lb);
    else
        scatter_chunk<uint32_t, 16, 8, false, RANK_ATOMIC, false>(sm, kin, nullptr, keys_out, nullptr, valid, dg, unused, lb);
}

// ---------------------------------------------------------------------------------------------
// The plan, once the first pass has run: ONE workgroup of 256 threads, thread a = top byte a.
__global__ __launch_bounds__(256) void pool_plan_kernel(MsdPlan *__restrict__ msd, PoolPlan *__restrict__ pool, uint32_t n, uint32_t tiles_b_cap) {
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_tiles[8][kBins];       // [XCD x][entry e = 8 k + s]: tiles of slice s's share of top byte x + 8 k
    __shared__ uint32_t s_tiles_b, s_bad;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t shift = pool->shift, armed = pool->armed, failed = pool->fail;
    if (tid == 0) {
        s_tiles_b = 0;
        s_bad = 0;
    }
    __syncthreads();
    uint32_t keys_a = 0, tiles_a = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const uint32_t c = msd->cursor_a[s][tid];
        const uint32_t t = (c + kPoolTile - 1u) / kPoolTile;
        keys_a += c;
        tiles_a += t;
        s_tiles[tid & 7u][(tid >> 3) * 8u + s] = t;
    }
    if (tiles_a > kPoolMaxTiles) s_bad = 1;  // a bucket of this top byte would have more runs than the local sort gathers
    pool->top_tiles[tid] = tiles_a;
    uint32_t incl = keys_a;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if (lane >= static_cast<uint32_t>(o)) incl += t;
    }
    if (lane == 63u) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = incl - keys_a;
    for (uint32_t w = 0; w < wave; ++w) before += s_wave[w];
    pool->top_base[tid] = before;
    if (tid == 255u) {
        pool->top_base[256] = before + keys_a;
        if (before + keys_a != n) s_bad = 1;  // keys the first pass did not place: it did not run, or a workgroup left early
    }
    // the exclusive prefix of every XCD's 256 entries: wave w takes XCDs w and w + 4, four entries per lane
    for (uint32_t x = wave; x < 8u; x += 4u) {
        uint32_t t[4], tot = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[j] = s_tiles[x][4 * lane + j];
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
            pool->tiles_b[x][4 * lane + j] = a;
            a += t[j];
        }
        if (lane == 63u) {
            pool->tiles_b[x][kBins] = a;
            atomicMax(&s_tiles_b, a);
        }
    }
    __syncthreads();
    if (tid == 0) {
        pool->ok_a = (armed != 0u && failed == 0u && s_bad == 0u && shift >= kPoolMinShift && shift <= kPoolMaxShift && s_tiles_b <= tiles_b_cap) ? 1u : 0u;
        pool->max_bucket = 0;  // (PoolPlan::fail stays: the second pass may still set it; the next sort's sample kernel re-arms it)
        msd->shift = shift;
        msd->sub_bits = kMsdSubBits;
        msd->ok = 0;           // the local sort decides
    }
}

// ---------------------------------------------------------------------------------------------
// Second pass: block b -> XCD b % 8 (no correctness in that), which walks the shares (top byte x + 8 k, slice s) in entry order
// e = 8 k + s.  A tile reads 8192 slots of its share -- the share's positions run through its primary region, then its overflow
// region -- groups the keys by the 6 bits below the top byte in LDS and writes them back to the SAME slots.  No offset from
// anywhere: the tile's row says where each of its 64 runs starts and how long it is.
struct alignas(16) PoolSmemB {
    uint32_t keys[kPoolTile];
    alignas(16) uint32_t whist[8][kMsdSub];  // per-wave counters of the 64 digits -> per-wave starts
};

// vector: a full tile inside the primary region (16-byte aligned): 16-byte loads and stores.  The general form (a share's ragged
// last tile, a tile that runs on into the overflow region: one tile in seven) goes through LDS with ROLLED loops on both sides --
// its index arithmetic unrolled would cost the common form its third workgroup per CU, and out of line (a call) it cost 80 bytes
// of scratch per lane saved and restored through HBM: 164 MB per sort, a fifth of this pass's traffic (profiles/r04: 968 -> 8xx MB).
__device__ __forceinline__ uint32_t pool_tile_b(PoolSmemB &sm, bool vector, uint32_t *__restrict__ t0, uint32_t *__restrict__ t1, uint32_t split,
                                                uint32_t valid, uint32_t shift, uint32_t key_base, uint32_t *__restrict__ row) {
    constexpr int ITEMS = 16, WAVES = 8;
    constexpr uint32_t THREADS = WAVES * 64;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t key[ITEMS];
    if (vector) {  // (workgroup-uniform)
        const uint4 *v = reinterpret_cast<const uint4 *>(t0);
#pragma unroll
        for (int i = 0; i < ITEMS / 4; ++i) {
            const uint4 q = v[i * THREADS + tid];
            key[4 * i] = q.x;
            key[4 * i + 1] = q.y;
            key[4 * i + 2] = q.z;
            key[4 * i + 3] = q.w;
        }
    } else {  // tile positions below `split` at t0, the others at t1; positions >= valid hold the padding key (digit 63, ranks last)
#pragma unroll 1
        for (uint32_t q = tid; q < kPoolTile; q += THREADS) sm.keys[q] = q < valid ? *(q < split ? t0 + q : t1 + q) : key_base - 1u;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) key[i] = sm.keys[wave * (ITEMS * 64) + i * 64 + lane];
        // (the keys are written back to sm.keys two barriers further on)
    }
    uint32_t *my_hist = sm.whist[wave];
    my_hist[lane] = 0;  // this wave's own row (64 counters): its LDS operations stay in order
    // a key outside the probed range has bits above the range's 14 + shift (a range of 32 bits has no such key)
    const uint32_t above = shift + kMsdBits < 32u ? ~0u << (shift + kMsdBits) : 0u;
    uint32_t rank2[ITEMS / 2], over = 0;  // two 13-bit ranks per register: 8 registers fewer, and the kernel fits a fourth workgroup per CU
    if (vector) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) over |= (key[i] - key_base) & above;
    } else {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
            if (wave * (ITEMS * 64) + i * 64 + lane < valid) over |= (key[i] - key_base) & above;  // (the padding key is no key)
    }
    // (key_base is a multiple of 2^24 and the 6 bits end at or below bit 24: the digit needs no subtraction)
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t d = (key[i] >> shift) & (kMsdSub - 1u);
        const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
        uint32_t r;
        if (__ballot(d == d0) == ~0ull) {
            uint32_t old = 0;
            if (lane == 0u) old = __hip_atomic_fetch_add(&my_hist[d0], 64u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            r = __builtin_amdgcn_readfirstlane(old) + lane;
        } else {
            r = __hip_atomic_fetch_add(&my_hist[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (i & 1) rank2[i / 2] |= r << 16;
        else rank2[i / 2] = r;
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    // every wave: lane l = digit l -- its total, its start inside the tile, this wave's own start
    uint32_t tot = 0, mine = 0;
#pragma unroll
    for (int v = 0; v < WAVES; ++v) {
        const uint32_t c = sm.whist[v][lane];
        mine += static_cast<uint32_t>(v) < wave ? c : 0u;
        tot += c;
    }
    uint32_t incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if (lane >= static_cast<uint32_t>(o)) incl += t;
    }
    const uint32_t start = incl - tot;
    if (wave == 0u) {  // the tile's row: where run l starts inside the tile, and its keys (the padding of a ragged tile is not a key)
        const uint32_t real = tot - (lane == kMsdSub - 1u ? kPoolTile - valid : 0u);
        row[lane] = start | (real << 16);
    }
    __syncthreads();  // every wave has read every row's counts
    my_hist[lane] = start + mine;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)  // (ranks stay below 8192: no carry; opaque: the digits are computed again, not kept in 16 registers across the barriers)
        rank2[i / 2] += my_hist[(opaque(key[i]) >> shift) & (kMsdSub - 1u)] << (16 * (i & 1));
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) sm.keys[(rank2[i / 2] >> (16 * (i & 1))) & 0xFFFFu] = key[i];
    __syncthreads();
    // back to the slots the tile was read from, in tile order
    if (vector) {
        uint4 *v = reinterpret_cast<uint4 *>(t0);
#pragma unroll
        for (int i = 0; i < ITEMS / 4; ++i) v[i * THREADS + tid] = reinterpret_cast<const uint4 *>(sm.keys)[i * THREADS + tid];
    } else {
#pragma unroll 1
        for (uint32_t q = tid; q < valid; q += THREADS) *(q < split ? t0 + q : t1 + q) = sm.keys[q];
    }
    return over;
}

__global__ __launch_bounds__(512, 8) void pool_pass_b_kernel(uint32_t *__restrict__ regions, uint32_t *__restrict__ overflow, const MsdPlan *__restrict__ msd,
                                                             PoolPlan *__restrict__ pool, uint32_t *__restrict__ rows, uint32_t key_base) {
    __shared__ PoolSmemB sm;
    const uint32_t x = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const uint32_t *pt = pool->tiles_b[x];
    if (pool->ok_a == 0u || j >= pt[kBins]) return;  // uniform (enqueued before the plan was known: it may have said no)
    uint32_t e = 0;  // the entry whose tiles contain j: largest e with pt[e] <= j
#pragma unroll
    for (uint32_t step = 128; step >= 1; step >>= 1)
        if (pt[e + step] <= j) e += step;
    const uint32_t a = x + 8u * (e >> 3), s = e & 7u, i = j - pt[e];  // (a share's tiles in descending order, for what the first pass wrote last: no difference)
    const uint32_t keys_sa = msd->cursor_a[s][a];               // keys of this share (the first pass's cursor)
    const uint32_t prim = min(keys_sa, pool->cap[s][a]);        // ... of them in the primary region, the rest in the overflow region
    const uint32_t done = i * kPoolTile;
    const uint32_t valid = min(kPoolTile, keys_sa - done);
    const uint32_t split = prim > done ? prim - done : 0u;      // leading tile positions that lie in the primary region
    uint32_t *t0 = regions + pool->base[s][a] + done;
    uint32_t *t1 = overflow + pool->obase[s][a] + (static_cast<int64_t>(done) - static_cast<int64_t>(prim));
    uint32_t *row = rows + static_cast<size_t>(blockIdx.x) * kMsdSub;
    const bool vector = valid == kPoolTile && split >= kPoolTile && (reinterpret_cast<uintptr_t>(t0) & 15u) == 0u;  // workgroup-uniform
    const uint32_t over = pool_tile_b(sm, vector, t0, t1, split, valid, msd->shift, key_base, row);
    // a key above the probed range (or below the promised floor): the local sort, which gives the last verdict, sees this
    if (__ballot(over != 0u) != 0ull && (threadIdx.x & 63u) == 0u) __hip_atomic_fetch_or(&pool->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// Runs: workgroup a = top byte a (256 threads).  The rows of its tiles -> for every bucket (a, c) its 64 run descriptors
// (slot r = the run of the top byte's r-th tile; slots R .. R + 7 = the overflow piece of a run whose tile crosses its share's
// primary region's end, one per slice, usually empty; the rest empty), its start in the sorted order, its size.
// (256 / 512 / 1024 threads: 16.8 / 15.3 / 15.0 us -- the kernel is a chain of dependent loads, not of work)
constexpr uint32_t kRunsThreads = 512, kRunsQ = kRunsThreads / kMsdSub, kRunsPer = (kPoolRunSlots + kRunsQ - 1u) / kRunsQ;
__global__ __launch_bounds__(kRunsThreads) void pool_runs_kernel(MsdPlan *__restrict__ msd, PoolPlan *__restrict__ pool, const uint32_t *__restrict__ rows,
                                                        PoolRun *__restrict__ runs, uint32_t n) {
    __shared__ uint32_t s_rows[kPoolMaxTiles][kMsdSub];
    __shared__ uint2 s_desc[kMsdSub][kPoolRunSlots + 1];  // [bucket][run slot] (+1: the buckets' rows on different banks)
    __shared__ uint32_t s_pos[kPoolMaxTiles], s_prim[kPoolMaxTiles], s_base[kPoolMaxTiles], s_obase[kPoolMaxTiles], s_share[kPoolMaxTiles],
        s_rowidx[kPoolMaxTiles];
    __shared__ uint32_t s_part[kRunsQ][kMsdSub], s_tot[kMsdSub];
    const uint32_t tid = threadIdx.x, a = blockIdx.x, x = a & 7u, e0 = (a >> 3) * 8u;
    const uint32_t ok_a = pool->ok_a;
    uint32_t biggest = 0;
    if (ok_a) {
        const uint32_t *pt = pool->tiles_b[x];
        const uint32_t R = pool->top_tiles[a];  // <= kPoolMaxTiles (verdict 1)
        if (tid < R) {  // run r = tid: which share, which tile of it
            const uint32_t j = pt[e0] + tid;
            uint32_t s = 0;
#pragma unroll
            for (uint32_t q = 1; q < 8u; ++q) s += pt[e0 + q] <= j ? 1u : 0u;  // entries are non-decreasing: the share whose tiles contain j
            const uint32_t i = j - pt[e0 + s];
            const uint32_t keys_sa = msd->cursor_a[s][a];
            s_share[tid] = s;
            s_pos[tid] = i * kPoolTile;                               // the tile's first position inside its share
            s_prim[tid] = min(keys_sa, pool->cap[s][a]);              // positions below this lie in the primary region
            s_base[tid] = pool->base[s][a];
            s_obase[tid] = pool->obase[s][a];
            s_rowidx[tid] = j * 8u + x;                               // the tile's row: the second pass's block index
        }
        for (uint32_t w = tid; w < kMsdSub * (kPoolRunSlots + 1u); w += kRunsThreads) (&s_desc[0][0])[w] = make_uint2(0, 0);
        __syncthreads();
        for (uint32_t w = tid; w < R * kMsdSub; w += kRunsThreads)  // the rows, coalesced (14 KB per workgroup)
            s_rows[w >> kMsdSubBits][w & (kMsdSub - 1u)] = rows[static_cast<size_t>(s_rowidx[w >> kMsdSubBits]) * kMsdSub + (w & (kMsdSub - 1u))];
        __syncthreads();
        const uint32_t c = tid & (kMsdSub - 1u), q = tid >> kMsdSubBits;
        {   // thread (q, c): the runs r in [kRunsPer q, kRunsPer q + kRunsPer) of bucket (a, c)
            uint32_t total = 0;
            for (uint32_t r = kRunsPer * q; r < min(kRunsPer * q + kRunsPer, R); ++r) {
                const uint32_t w = s_rows[r][c];
                const uint32_t off = w & 0xFFFFu, len = w >> 16;
                const uint32_t pos = s_pos[r] + off, prim = s_prim[r];
                uint2 d;
                if (pos + len <= prim) {
                    d = make_uint2(s_base[r] + pos, len);
                } else if (pos >= prim) {
                    d = make_uint2(n + s_obase[r] + (pos - prim), len);
                } else {  // the run crosses from the primary region into the overflow region: two pieces (one such tile per share at most)
                    d = make_uint2(s_base[r] + pos, prim - pos);
                    s_desc[c][R + s_share[r]] = make_uint2(n + s_obase[r], pos + len - prim);
                }
                s_desc[c][r] = d;
                total += d.y;  // (a crossing run's second piece is counted with the pieces below)
            }
            s_part[q][c] = total;
        }
        __syncthreads();
        {   // every run's place inside the bucket: (keys before the run) | (the run's keys) << 16; the pieces behind the runs, the
            // unused slots behind those (empty, at the bucket's end: the local sort reads the bucket's size off the last slot)
            uint32_t off = 0;
            for (uint32_t qq = 0; qq < q; ++qq) off += s_part[qq][c];
            for (uint32_t r = kRunsPer * q; r < min(kRunsPer * q + kRunsPer, R); ++r) {
                const uint32_t len = s_desc[c][r].y;
                s_desc[c][r].y = min(off, 0xFFFFu) | (len << 16);  // (a bucket beyond 65535 keys is refused anyway)
                off += len;
            }
            if (q == kRunsQ - 1u) {
                // the second pieces of crossing runs (at most one per slice, usually none) move up behind the runs: the local sort
                // stops at the last slot that holds keys
                uint32_t w = R;
                for (uint32_t r = R; r < R + 8u; ++r) {
                    const uint2 d = s_desc[c][r];
                    s_desc[c][r] = make_uint2(0, 0);
                    if (d.y) {
                        s_desc[c][w++] = make_uint2(d.x, min(off, 0xFFFFu) | (d.y << 16));
                        off += d.y;
                    }
                }
                for (uint32_t r = w; r < kPoolRunSlots; ++r) s_desc[c][r].y = min(off, 0xFFFFu);  // empty, at the bucket's end (slot 63 says its size)
                s_tot[c] = off;  // the bucket's keys
            }
        }
        __syncthreads();
        if (tid < kMsdSub) {  // where the bucket starts: the top byte's start + the buckets before it
            const uint32_t total = s_tot[tid];
            uint32_t incl = total;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(incl, o);
                if (tid >= static_cast<uint32_t>(o)) incl += t;
            }
            msd->base[a * kMsdSub + tid] = pool->top_base[a] + incl - total;
            biggest = total;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) biggest = max(biggest, __shfl_down(biggest, o));
        }
        // the descriptors, bucket by bucket: what one local-sort workgroup reads is 512 contiguous bytes
        uint2 *out = reinterpret_cast<uint2 *>(runs) + static_cast<size_t>(a) * kMsdSub * kPoolRunSlots;
        for (uint32_t w = tid; w < kMsdSub * kPoolRunSlots; w += kRunsThreads) out[w] = s_desc[w / kPoolRunSlots][w % kPoolRunSlots];
        if (a == 255u && tid == 0) msd->base[kPoolBuckets] = n;
    }
    // (verdict 2 is the local sort's: every one of its workgroups reads ok_a, fail and max_bucket -- all final when it starts)
    if (tid == 0 && biggest) __hip_atomic_fetch_max(&pool->max_bucket, biggest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// Local sort: workgroup b = bucket b.  The bucket's keys lie in up to 64 runs.  They are first copied, run by run, into the
// sort's LDS buffer -- wave w takes runs w, w + WAVES, ...; consecutive lanes read consecutive keys of a run: coalesced, no search
// -- at word mis + (key's index in the bucket), mis = the misalignment of the bucket's FINAL place: from there every thread reads
// its 16-byte vectors exactly as lean_sort_bucket (vrs_msd_hybrid.hip) reads them from the bucket in global memory.
struct PoolGather {
    const uint32_t *regions, *overflow;
    uint32_t n_virt;
    // (the pointers cross a function boundary: say that they are GLOBAL ones, or the loads become flat loads)
    using gptr = const uint32_t __attribute__((address_space(1))) *;
    __device__ __forceinline__ gptr at(uint32_t v) const { return v < n_virt ? (gptr)regions + v : (gptr)overflow + (v - n_virt); }
};

// s_keys[mis + g] = key g of the bucket; all threads of the workgroup, ends with a barrier.  dx, dy: lane l holds run l's
// descriptor (virtual slot of its first key; keys of the bucket before it | its keys << 16) -- every wave has loaded all 64, so a
// run's descriptor is two v_readlane away and its address a scalar: no LDS, no barrier, no search in front of the loads.
// (inline: a call would save and restore the callee's registers through scratch, 300 bytes per thread of a kernel with 4 M threads)
template <int THREADS>
__device__ __forceinline__ void pool_stage(const PoolGather &gt, uint32_t dx, uint32_t dy, uint32_t *s_keys, uint32_t mis) {
    constexpr uint32_t WAVES = THREADS / 64, PER = 64 / WAVES;  // runs per wave
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // ALL of the wave's runs at once, three 64-key chunks of each (a run is about 128 keys): every load of the wave is in flight
    // before the first is written to LDS -- a workgroup lives for one memory latency here, and four would be four.  The loads of
    // the first two chunks are unconditional, from a clamped index (a predicated load is a branch per lane); the third chunk and
    // everything behind the last slot that holds keys are skipped wave by wave (scalar branches).
    const uint32_t used = 64u - static_cast<uint32_t>(__clzll(static_cast<long long>(__ballot((dy >> 16) != 0u) | 1ull)));  // slots [0, used) may hold keys
    uint32_t x[PER][3];
    uint32_t slot[PER], pk[PER];  // (wave-uniform: scalar registers)
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
        const uint32_t r = wave + u * WAVES;
        slot[u] = __builtin_amdgcn_readlane(dx, r);
        pk[u] = __builtin_amdgcn_readlane(dy, r);
        x[u][0] = x[u][1] = x[u][2] = 0;
        if (r < used) {
            const uint32_t len = pk[u] >> 16;
            const PoolGather::gptr src = gt.at(slot[u]);  // (an empty run's slot is 0: a valid address)
            const uint32_t last = len ? len - 1u : 0u;
            // (plain loads: nontemporal ones here measured 204 instead of 200 us)
            x[u][0] = src[min(lane, last)];
            x[u][1] = src[min(64u + lane, last)];
            if (len > 128u) x[u][2] = src[min(128u + lane, last)];
        }
    }
#pragma unroll
    for (uint32_t u = 0; u < PER; ++u) {
        const uint32_t len = pk[u] >> 16, dst = mis + (pk[u] & 0xFFFFu);
        if (wave + u * WAVES < used) {
#pragma unroll
            for (uint32_t c = 0; c < 3u; ++c) {
                const uint32_t idx = c * 64u + lane;
                if (idx < len) s_keys[dst + idx] = x[u][c];
            }
            if (len > 192u) {  // a long run (skewed keys): the rest of it, 64 keys at a time
                const PoolGather::gptr src = gt.at(slot[u]);
                for (uint32_t idx = 192u + lane; idx < len; idx += 64u) s_keys[dst + idx] = src[idx];
            }
        }
    }
    __syncthreads();
}

template <int THREADS, int VEC>
__device__ __forceinline__ void pool_read_staged(uint32_t (&k)[4 * VEC], const uint32_t *s_keys, uint32_t nvec) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        uint32_t v = j * THREADS + threadIdx.x;
        if (j == VEC - 1) v = v < nvec ? v : nvec - 1u;  // only the last row can reach behind the bucket
        const uint4 t = reinterpret_cast<const uint4 *>(s_keys)[v];
        k[4 * j] = t.x;
        k[4 * j + 1] = t.y;
        k[4 * j + 2] = t.z;
        k[4 * j + 3] = t.w;
    }
}

template <int THREADS, int VEC>
__device__ __attribute__((noinline)) void pool_sort_guarded(const PoolGather gt, uint32_t dx, uint32_t dy, uint32_t *abase, uint32_t mis, uint32_t n,
                                                           uint32_t *s_keys, uint32_t *s_hist2, uint32_t *s_tmp, uint32_t guards) {
    // (the common path has consumed the staged keys' registers: stage again -- run by run, in a loop: this function's registers
    // are the KERNEL's registers, whichever path a bucket takes, and the form of pool_stage with every load in flight at once
    // would cost every workgroup its occupancy)
    __syncthreads();
    {
        constexpr uint32_t WAVES = THREADS / 64;
        const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll 1
        for (uint32_t r = wave; r < kPoolRunSlots; r += WAVES) {
            const uint32_t pk = __builtin_amdgcn_readlane(dy, r);
            const PoolGather::gptr src = gt.at(__builtin_amdgcn_readlane(dx, r));
            const uint32_t dst = mis + (pk & 0xFFFFu);
#pragma unroll 1
            for (uint32_t idx = lane; idx < (pk >> 16); idx += 64u) s_keys[dst + idx] = src[idx];
        }
        __syncthreads();
    }
    uint32_t k[4 * VEC];
    pool_read_staged<THREADS, VEC>(k, s_keys, (mis + n + 3u) / 4u);
    __syncthreads();  // every vector is in registers before pass 1 writes the buffer
    lean_sort_body<THREADS, VEC, true, true>(k, abase, mis, n, s_keys, s_hist2, s_tmp, (guards & 1u) != 0u, (guards & 2u) != 0u);
}

template <int THREADS, int VEC>
__device__ __forceinline__ void pool_sort_bucket(const PoolGather &gt, uint32_t dx, uint32_t dy, uint32_t *abase, uint32_t mis, uint32_t n, uint32_t *s_keys,
                                                 uint32_t *s_hist2, uint32_t *s_tmp) {
    constexpr int WAVES = THREADS / 64;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t k[4 * VEC];
    pool_read_staged<THREADS, VEC>(k, s_keys, (mis + n + 3u) / 4u);
    {   // every counter table zeroed here (lean_sort_bucket does the same): WAVES tables of pass 2, then pass 1's
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
    __syncthreads();  // (also: every vector is in registers before pass 1 writes the buffer)
    uint32_t guards = 0;
#pragma unroll
    for (int v = 0; v < WAVES; ++v) guards |= s_tmp[16 + v];
    guards = __builtin_amdgcn_readfirstlane(guards);
    // (two copies of the rest, the guarded one out of line and staging again: see lean_sort_bucket, vrs_msd_hybrid.hip)
    if (guards == 0u) lean_sort_body<THREADS, VEC, false, true>(k, abase, mis, n, s_keys, s_hist2, s_tmp, false, false);
    else pool_sort_guarded<THREADS, VEC>(gt, dx, dy, abase, mis, n, s_keys, s_hist2, s_tmp, guards);
}

template <int THREADS>
__global__ __launch_bounds__(THREADS, 4) void pool_local_sort_kernel(const uint32_t *__restrict__ regions, const uint32_t *__restrict__ overflow,
                                                                     uint32_t *__restrict__ keys_out, uint32_t n_virt, MsdPlan *__restrict__ msd,
                                                                     const PoolPlan *__restrict__ pool, const PoolRun *__restrict__ runs,
                                                                     uint32_t *__restrict__ cursors, OnesweepPlanHead *__restrict__ dev_head,
                                                                     OnesweepPlanHead *host_head, uint32_t stamp) {
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[THREADS * 4 * kLeanMaxVec + 4];
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[(THREADS / 64 + 1) * kLeanRow];
    __shared__ uint32_t s_tmp[32];
    // (bucket = block index: neighbouring buckets run on different XCDs and fetch the cache lines they share twice -- 22 % more
    // bytes than the keys -- but with XCD-contiguous ranges of buckets, xcd_contiguous_tile, the gather measured 178 instead of 160 us)
    // (groups of 2 .. 16 neighbouring buckets per XCD, so that the lines neighbours share are fetched once: no difference, 218 us)
    // ... and the LAST bucket first: the second pass wrote the top bytes in ascending order, the highest are what the memory-side
    // cache still holds (215 -> 208 us)
    const uint32_t b = kPoolBuckets - 1u - blockIdx.x;
    // The bucket's start and its run descriptors are asked for BEFORE the verdict is looked at (both tables exist whatever it
    // says): a workgroup lives for a few memory latencies, and the verdict's words would be one more in front of these.
    // every wave: lane l = run l's descriptor (512 contiguous bytes of the table)
    const uint32_t begin = msd->base[b];
    const uint2 d = reinterpret_cast<const uint2 *>(runs)[static_cast<size_t>(b) * kPoolRunSlots + (threadIdx.x & 63u)];
    // Verdict 2, by every workgroup from the same three words (all final when this kernel starts): verdict 1 said yes, no pass
    // flagged the sort, and the largest bucket fits this kernel's shape.  Workgroup 0 tells the host.
    const uint32_t mx = pool->max_bucket;
    const uint32_t ok = (pool->ok_a != 0u && pool->fail == 0u && mx <= THREADS * 4u * kLeanMaxVec - 3u) ? 1u : 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        msd->ok = ok;
        dev_head->msd_ok = ok;
        dev_head->msd_max_bucket = mx;
        dev_head->lsd_missing = 1u;
        if (host_head) {
            __hip_atomic_store(&host_head->lsd_missing, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&host_head->msd_ok, ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&host_head->msd_max_bucket, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
            __hip_atomic_store(&host_head->ready, stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (ok == 0u) return;  // (enqueued before the verdicts were known, and one said no)
    // the first pass's cursors, zero for the next sort (the counted form's local sort does the same: rearm_reservation)
    if (blockIdx.x < 2u * kStreams)
        for (uint32_t c = threadIdx.x; c < 256u; c += THREADS) cursors[blockIdx.x * 256u + c] = 0;
    const uint32_t pk63 = __builtin_amdgcn_readlane(d.y, 63);
    const uint32_t n = (pk63 & 0xFFFFu) + (pk63 >> 16);
    const uint32_t mis = static_cast<uint32_t>((reinterpret_cast<uintptr_t>(keys_out + begin) >> 2) & 3u);
    if (n == 0 || mis + n > THREADS * 4u * kLeanMaxVec) return;  // uniform; above the capacity cannot happen (verdict 2 would have said no)
    uint32_t *abase = keys_out + begin - mis;
    const PoolGather gt{regions, overflow, n_virt};
    pool_stage<THREADS>(gt, d.x, d.y, s_keys, mis);
    switch ((mis + n + 4u * THREADS - 1u) / (4u * THREADS)) {  // rows of THREADS vectors the bucket touches
        case 1: pool_sort_bucket<THREADS, 1>(gt, d.x, d.y, abase, mis, n, s_keys, s_hist, s_tmp); break;
        case 2: pool_sort_bucket<THREADS, 2>(gt, d.x, d.y, abase, mis, n, s_keys, s_hist, s_tmp); break;
        case 3: pool_sort_bucket<THREADS, 3>(gt, d.x, d.y, abase, mis, n, s_keys, s_hist, s_tmp); break;
        case 4: pool_sort_bucket<THREADS, 4>(gt, d.x, d.y, abase, mis, n, s_keys, s_hist, s_tmp); break;
        case 5: pool_sort_bucket<THREADS, 5>(gt, d.x, d.y, abase, mis, n, s_keys, s_hist, s_tmp); break;
        case 6: pool_sort_bucket<THREADS, 6>(gt, d.x, d.y, abase, mis, n, s_keys, s_hist, s_tmp); break;
        default: pool_sort_bucket<THREADS, 7>(gt, d.x, d.y, abase, mis, n, s_keys, s_hist, s_tmp); break;
    }
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
    const double room = 6.0 * std::sqrt(2048.0 * 33.0 * (static_cast<double>(n) + 2048.0 * 33.0)) + 2048.0 * (kPoolRoomFloor + 64.0);
    return static_cast<uint32_t>(std::min<double>(room, 1u << 28)) & ~31u;
}

uint32_t pool_tiles_b_cap(uint32_t n) {
    const uint32_t tiles = (n + kPoolTile - 1u) / kPoolTile, even = (tiles + 7u) / 8u;
    // an XCD walks 32 top bytes x 8 shares, each rounded up to whole tiles; the grid is sized before the plan is known
    return even + even / 8u + 256u + 40u;
}

size_t pool_rows_bytes(uint32_t n) { return static_cast<size_t>(8u) * pool_tiles_b_cap(n) * kMsdSub * sizeof(uint32_t); }

uint32_t pool_local_capacity(bool big) { return (big ? 512u : 256u) * 4u * kLeanMaxVec - 3u; }

hipError_t launch_pool_sample(hipStream_t stream, const uint32_t *keys, uint32_t n, uint32_t key_base, const PoolStreams &ps,
                              PoolPlan *pool, uint32_t overflow_capacity, LaunchEvents ev) {
    if (n == 0 || ps.tiles_per_stream < kPoolSampleTiles) return hipErrorInvalidValue;  // (a sample workgroup's tiles span at most two slices)
    const uint32_t grid = (ps.tiles_total + kPoolSampleTiles - 1u) / kPoolSampleTiles;
    VRS_LAUNCH(pool_sample_kernel, dim3(grid), dim3(256), stream, ev, keys, n, key_base, ps, pool);
    hipLaunchKernelGGL(pool_layout_kernel, dim3(1), dim3(256), 0, stream, ps, pool, overflow_capacity);
    return hipGetLastError();
}

hipError_t launch_pool_pass_a(hipStream_t stream, const uint32_t *keys_in, uint32_t *keys_out, uint32_t *overflow, uint32_t n,
                              uint32_t key_base, const PoolStreams &ps, PoolPlan *pool, MsdPlan *msd, unsigned long long xcc_map,
                              bool misplace, uint32_t overflow_capacity, LaunchEvents ev) {
    VRS_LAUNCH(pool_pass_a_kernel, dim3(8u * ps.tiles_per_stream), dim3(512), stream, ev, keys_in, keys_out, overflow, n, key_base, ps, pool, msd,
               xcc_map, misplace ? 1 : 0, overflow_capacity);
    return hipGetLastError();
}

hipError_t launch_pool_plan(hipStream_t stream, MsdPlan *msd, PoolPlan *pool, uint32_t n, uint32_t tiles_b_cap) {
    hipLaunchKernelGGL(pool_plan_kernel, dim3(1), dim3(256), 0, stream, msd, pool, n, tiles_b_cap);
    return hipGetLastError();
}

hipError_t launch_pool_pass_b(hipStream_t stream, uint32_t *regions, uint32_t *overflow, uint32_t n, MsdPlan *msd, PoolPlan *pool,
                              uint32_t *rows, uint32_t tiles_b, uint32_t key_base, LaunchEvents ev) {
    (void)n;
    if (tiles_b == 0) return hipSuccess;
    VRS_LAUNCH(pool_pass_b_kernel, dim3(8u * tiles_b), dim3(512), stream, ev, regions, overflow, msd, pool, rows, key_base);
    return hipGetLastError();
}

hipError_t launch_pool_runs(hipStream_t stream, MsdPlan *msd, PoolPlan *pool, const uint32_t *rows, PoolRun *runs, uint32_t n) {
    hipLaunchKernelGGL(pool_runs_kernel, dim3(256), dim3(kRunsThreads), 0, stream, msd, pool, rows, runs, n);
    return hipGetLastError();
}

hipError_t launch_pool_local_sort(hipStream_t stream, const uint32_t *regions, const uint32_t *overflow, uint32_t *keys_out, uint32_t n,
                                  MsdPlan *msd, const PoolPlan *pool, const PoolRun *runs, bool big, OnesweepPlanHead *dev_head,
                                  OnesweepPlanHead *host_head, uint32_t stamp, LaunchEvents ev) {
    if (big)
        VRS_LAUNCH(pool_local_sort_kernel<512>, dim3(kMsdBucketCount), dim3(512), stream, ev, regions, overflow, keys_out, n, msd, pool, runs,
                   &msd->cursor_a[0][0], dev_head, host_head, stamp);
    else
        VRS_LAUNCH(pool_local_sort_kernel<256>, dim3(kMsdBucketCount), dim3(256), stream, ev, regions, overflow, keys_out, n, msd, pool, runs,
                   &msd->cursor_a[0][0], dev_head, host_head, stamp);
    return hipGetLastError();
}

}  // namespace vrs
