// vrs_msd_pool.hip -- the hybrid form of the one-call sort WITHOUT a counting read (bare uint32 keys): 24 bytes per key.
//
// The reference reads the keys once per pass just to count them (multi_radixsort_histograms.comp:42-50); the counted hybrid
// form (vrs_msd_hybrid.hip, K5b) still reads them once for that.  Here they are not read for counting at all, and no pass waits
// for a histogram either:
//
//   pool_sample_kernel      1/32 of the input (the first 256 keys of every 8192-key tile): probes the key range (bucket shift),
//                           counts the sampled keys by bucket (top 14 bits of the range) and, per input slice, by top byte;
//                           pool_layout_kernel (one workgroup behind it) lays out, for every (slice, top byte), a PRIMARY region of the
//                           partner buffer sized by the estimate (the estimates of a slice sum to its length, so the regions tile the
//                           n-key buffer) and an OVERFLOW region in context scratch of six standard deviations of that estimate;
//   pool_pass_a_kernel      first MSD pass (top 8 bits of the key range): a tile reserves its place in (slice, top byte)'s region
//                           with ONE atomic add on the region's cursor, in the L2 of the XCD that runs the slice (StreamReserve's
//                           idea, vrs_device.hpp) -- positions below the region's capacity are primary slots, the rest overflow;
//   pool_plan_kernel        ONE workgroup: top-byte totals are the cursors' sums, exact -- where every top byte starts in the
//                           sorted order, the second pass's tile tables; and a region of the SLACK buffer (context scratch, about
//                           1.5 n slots) for each of the 16384 buckets: its share of the top byte's exact total as the sample saw
//                           it plus six standard deviations, a multiple of 4 slots; verdict 1;
//   pool_pass_b_kernel      second MSD pass (the next 6 bits), regions -> slack buffer: scatter_chunk with one L2-local reservation
//                           per tile and bucket (SlackReserve) -- all tiles of a top byte run behind one L2.  It looks at every key:
//                           one outside the probed range, a bucket that outgrows its region or the local sort's capacity flag the sort;
//   pool_local_sort_kernel  (vrs_msd_pool_local.hip) one workgroup per bucket -- every one derives verdict 2 from the same two words, workgroup 0 tells
//                           the host: reads the bucket (ONE contiguous, 16-byte aligned piece of the slack buffer) in 16-byte
//                           vectors, sorts the keys by their low 18 bits inside LDS (lean_sort_body, vrs_local_sort.hpp) and
//                           streams the bucket to its final place in the caller's buffer: its top byte's exact start + the second
//                           pass's counts of the buckets before it (64 words, one load per lane).
//
// ("Top byte" = the first pass's digit: 8 bits of the key range until late in round 5, 7 by default since -- and 7 for the second pass:
// the same 16384 buckets, cut 128 x 128, `top_bits` below)
//
// (Round 4 grouped every second-pass tile IN PLACE and gathered every bucket from about 48 runs: no third buffer, but a gather that
// cost the local sort 45 us at 10^8 keys, a run-descriptor kernel, and at most 56 tiles per top byte.  CHANGELOG, round 5.)
//
// Nothing here is assumed about the data: a sample that misjudges a region (keys whose distribution changes inside a tile
// with the tile's period, say) makes a pass flag the sort, a verdict refuse, and the caller run the counted form on the
// untouched input.  Neither MSD pass is stable (arrival order inside a region); bare keys do not care.
#include "vrs_local_sort.hpp"

#include <algorithm>
#include <cmath>

namespace vrs {

namespace {

constexpr uint32_t kPoolMinShift = 13, kPoolMaxShift = 18;  // a 27 ... 32-bit key range (the counted form's rule)
constexpr float kPoolSigmas = 6.0f;                       // overflow room, in standard deviations of the region's estimate

// ---------------------------------------------------------------------------------------------
// The sample.  Workgroup g takes tiles [32 g, 32 g + 32) of the input; wave w of it the tiles 32 g + w + 4 j.
__global__ __launch_bounds__(256) void pool_sample_kernel(const uint32_t *__restrict__ keys, uint32_t n, uint32_t key_base,
                                                          PoolStreams ps, PoolPlan *__restrict__ pool, uint32_t top_bits) {
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
    // counting read does) and derives the same shift; a key outside that range is flagged by the second pass.
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
    const uint32_t dshift = shift + (kMsdBits - top_bits);  // the first pass's digit: the top 8 (lab: 7) bits of the range
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
__global__ __launch_bounds__(256) void pool_layout_kernel(PoolStreams ps, PoolPlan *__restrict__ pool, uint32_t overflow_capacity, uint32_t par) {
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
        pool->fail[par] = 0;  // re-armed here: the passes of THIS sort set it, its local sort reads it
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
    __device__ __forceinline__ void refuse() const {}  // (a reservation never waits: scatter_chunk's give-up path is dead code here)
    // run [reserved, reserved + cnt) of the region's position space; excl: where the digit's run starts inside the tile
    __device__ __forceinline__ void place(uint32_t *gbase, uint32_t tid, uint32_t, uint32_t excl) const { place_run(gbase, tid, reserved, excl); }
    __device__ __forceinline__ void place_run(uint32_t *gbase, uint32_t tid, uint32_t reserved, uint32_t excl) const {
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
    // payloads (the stable form below): to the twins of the two buffers, at the slots store() decided
    uint32_t *overflow_values = nullptr;
    template <int ITEMS, uint32_t THREADS, bool FULL>
    __device__ __forceinline__ void store_values(const uint32_t (&val)[ITEMS], const uint32_t (&dst)[ITEMS], uint32_t *vout, uint32_t valid) const {
        const uint32_t tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            uint32_t *p = dst[i] < n_virt ? vout + dst[i] : overflow_values + min(dst[i] - n_virt, overflow_last);
            if (FULL || i * THREADS + tid < valid) *p = val[i];
        }
    }
};

// The STABLE first pass (key + payload pairs): the same regions, the same split of a share's positions into primary and overflow
// slots -- but a tile's place in its share is its RANK there, the keys of the slice's earlier tiles with that top byte, which the
// decoupled look-back of the counted form's passes delivers (StreamLookback, vrs_device.hpp: one chain per slice, every tile of it
// behind the same L2).  Rank order is input order, so a share read primary part first, overflow part second, is read in input
// order.  The cursors are still added to (nobody waits for the answer): the plan kernel takes the shares' exact sizes from them.
struct PoolLookback : StreamLookback {
    static constexpr bool kPool = true;
    PoolReserve at;  // the regions' arithmetic (its reservation is not used)
    __device__ __forceinline__ void publish(uint32_t v) const {
        if (!at.reserved_yet) {  // the first call: this tile's count
            at.reserved_yet = true;
            at.cnt = v - at.pad_keys;
            if (at.cnt) __hip_atomic_fetch_add(at.cursor, at.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            at.rb = at.region[0];
            at.rc = at.region[8 * 256];
            at.ob = at.region[16 * 256];
            at.oc = at.region[24 * 256];
        }
        StreamLookback::publish(v);
    }
    __device__ __forceinline__ void refuse() const { __hip_atomic_fetch_or(at.fail_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ void place(uint32_t *gbase, uint32_t tid, uint32_t before, uint32_t excl) const { at.place_run(gbase, tid, before, excl); }
    template <typename K, int ITEMS, uint32_t THREADS, bool FULL, typename DG>
    __device__ __forceinline__ void store(const uint32_t *g, const K (&key)[ITEMS], uint32_t (&dst)[ITEMS], K *kout, uint32_t valid, const DG &dg) const {
        at.template store<K, ITEMS, THREADS, FULL>(g, key, dst, kout, valid, dg);
    }
    template <int ITEMS, uint32_t THREADS, bool FULL>
    __device__ __forceinline__ void store_values(const uint32_t (&val)[ITEMS], const uint32_t (&dst)[ITEMS], uint32_t *vout, uint32_t valid) const {
        at.template store_values<ITEMS, THREADS, FULL>(val, dst, vout, valid);
    }
};

// the place of the XCC this workgroup runs on in the probed order (8: an XCC the probe never saw)
__device__ __forceinline__ uint32_t xcc_place(unsigned long long xcc_map) {
    const uint32_t my_xcc = xcc_id();
    uint32_t x = 8u;
#pragma unroll
    for (uint32_t q = 0; q < 8u; ++q)
        if (x == 8u && xcc_of(xcc_map, q) == my_xcc) x = q;
    return x;
}

// grid = 8 * tiles_per_stream workgroups; a workgroup on the XCC of place x takes tile b >> 3 of slice x -- the slice whose keys the
// sample counted for the regions of row x, the row whose cursors live in this CU's L2.  (Observed placement: block b on the XCC of
// place (b + r) % 8 with r fixed for a queue -- and a stream may move to another queue: r at the probe is not r now.  Nothing but
// "the blocks of a group of eight run on eight XCCs" is used, and that is checked: PoolPlan::claim_a.)
template <bool PAIRS>
__global__ __launch_bounds__(512, 4) void pool_pass_a_kernel(const uint32_t *__restrict__ keys_in, uint32_t *__restrict__ keys_out,
                                                             uint32_t *__restrict__ overflow, uint32_t n, uint32_t key_base,
                                                             PoolStreams ps, PoolPlan *__restrict__ pool, MsdPlan *__restrict__ msd,
                                                             unsigned long long xcc_map, int misplace, uint32_t overflow_capacity, uint32_t par,
                                                             PoolPayloads pv, uint32_t top_bits) {
    __shared__ ChunkSmem<uint32_t, 16, 8, PAIRS> sm;
    __shared__ uint32_t s_gbase2[kBins], s_split[kBins], s_flags;
    if (pool->armed == 0u) return;  // uniform: the sample kernel did not lay regions out (key range below 27 bits)
    const uint32_t i = blockIdx.x >> 3;
    const uint32_t s_out = xcc_place(xcc_map);
    if (s_out == 8u) {  // behind an L2 the probe never saw: no row is safe to add to -- the sort is refused
        if (threadIdx.x == 0) __hip_atomic_fetch_or(&pool->fail[par], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&pool->claim_a[s_out * kPoolMaxTilesA + i], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (L2-local: every claim on this word comes from this XCC)
    // misplace (test hook): odd tiles are read from the neighbouring slice
    const uint32_t s_in = (s_out + (misplace ? (i & 1u) : 0u)) & 7u;
    const uint32_t len = ps.len[s_in];
    const uint32_t done = i * kPoolTile;
    if (done >= len) return;
    if (threadIdx.x == 0) s_flags = 0;  // (set behind the chunk's barriers, read behind its last one)
    const uint32_t valid = min(kPoolTile, len - done);
    const uint32_t *kin = keys_in + ps.start[s_in] + done;
    const uint32_t d = threadIdx.x & 255u;
    RadixDigit<uint32_t> dg;
    dg.shift = pool->shift + (kMsdBits - top_bits);
    dg.base = key_base;
    PoolReserve at;
    at.cursor = &msd->cursor_a[s_out][d];
    at.region = &pool->base[s_out][d];
    at.pad_keys = d == dg(dg.template pad<uint32_t>()) ? kPoolTile - valid : 0u;  // (the padding key carries the largest digit: 255, or 127 of 7 bits)
    at.n_virt = n;
    at.overflow = overflow;
    at.overflow_last = overflow_capacity - 1u;
    at.gbase2 = s_gbase2;
    at.split = s_split;
    at.flags = &s_flags;
    at.fail_word = &pool->fail[par];
    uint32_t unused = 0;
    const bool stream_in = static_cast<size_t>(n) * sizeof(uint32_t) >= kStreamInBytes;  // (inputs beyond the caches: vrs_device.hpp)
    if constexpr (PAIRS) {
        // payloads: the tile's place in (slice, top byte)'s share is its rank there (PoolLookback) -- one chain per slice, row i of it
        at.overflow_values = pv.overflow_values;
        PoolLookback lb;
        lb.at = at;
        lb.stream_keys = keys_in + ps.start[s_in];
        lb.done = done;
        lb.col = pv.status + static_cast<size_t>(s_out) * kBins + d;
        lb.stride = static_cast<size_t>(8) * kBins;
        lb.index = static_cast<int>(i);
        lb.tag = 1u << kLbTagShift;
        lb.budget = pv.spin_budget;
        lb.hold = pv.hold_tile >= 0 && i == static_cast<uint32_t>(pv.hold_tile);
        const uint32_t *vin = pv.values_home + ps.start[s_in] + done;
        if (valid == kPoolTile)
            scatter_chunk<uint32_t, 16, 8, true, RANK_ATOMIC, true>(sm, kin, vin, keys_out, pv.values_partner, valid, dg, unused, lb, NoPieces{}, stream_in);
        else
            scatter_chunk<uint32_t, 16, 8, true, RANK_ATOMIC, false>(sm, kin, vin, keys_out, pv.values_partner, valid, dg, unused, lb);
    } else {
        if (valid == kPoolTile)
            scatter_chunk<uint32_t, 16, 8, false, RANK_ATOMIC, true>(sm, kin, nullptr, keys_out, nullptr, valid, dg, unused, at, NoPieces{}, stream_in);
        else
            scatter_chunk<uint32_t, 16, 8, false, RANK_ATOMIC, false>(sm, kin, nullptr, keys_out, nullptr, valid, dg, unused, at);
    }
}

// ---------------------------------------------------------------------------------------------
// The plan, once the first pass has run: workgroup a = top byte a (256 threads).  Every workgroup derives what it needs from the
// first pass's 2048 cursors -- the EXACT keys of every (slice, top byte) share -- alone, so the 256 of them run side by side:
//   * where the top byte starts in the sorted order (top_base), its row of 16 pieces, its place in its XCD's tile table;
//   * the top byte's part of the slack buffer: space(c) slots for a top byte of c keys, a bound on what its 64 regions may take
//     (below) that depends on c only -- so the part's start is a sum over the top bytes before it;
//   * the 64 regions: the workgroup SAMPLES its top byte where the first pass left it (the leading 256 keys of every 8192 of each
//     piece: 1/32 of the keys, about 50 KB, counted by the next 6 bits in LDS) and gives bucket b the share m_b / m of the exact
//     total c, six standard deviations of that estimate -- one sampled key stands for R keys: sqrt(R (est + R)), the layout kernel's
//     rule -- and a floor, rounded up to a multiple of 4 slots (every region starts on a 16-byte boundary);
//   * workgroup 255: verdict 1.
// space(c) = c + 6 sqrt(S R (c + S R)) + S (floor + 4) bounds the sum of the S = 64 or 128 rooms (Cauchy-Schwarz over sum est_b <= c).
constexpr float kPoolR = 40.0f;  // keys one sampled key stands for: 32, with a margin
template <uint32_t SUB>  // buckets per top byte
__device__ __forceinline__ uint32_t pool_space(uint32_t c) {
    if (c == 0u) return 0u;
    const float x = static_cast<float>(c), s = static_cast<float>(SUB);
    return (c + static_cast<uint32_t>(kPoolSigmas * sqrtf(s * kPoolR * (x + s * kPoolR))) + SUB * (kPoolRoomFloor + 4u) + 3u) & ~3u;
}
// GROUPED (vrs_msd_finish_grouped_counts_u32: the second half alone, for keys that ARE grouped by their top byte and whose caller
// knows how many each top byte holds -- a rank of the multi-GPU step after the exchange): the totals come from the caller (`groups`,
// by value), a top byte is ONE piece of `regions`, no claims of a first pass to check; the rest is the same.
template <uint32_t SUBBITS, bool GROUPED>
__global__ __launch_bounds__(512) void pool_plan_kernel(MsdPlan *__restrict__ msd, PoolPlan *__restrict__ pool, uint32_t n, uint32_t tiles_b_cap,
                                                       uint32_t slack_capacity, const uint32_t *__restrict__ regions, const uint32_t *__restrict__ overflow,
                                                       uint32_t key_base, PoolStreams ps, PoolGroups groups, uint32_t par, uint32_t keep, uint32_t top_bits) {
    // keep (a sort that started in a KEPT layout, launch_pool_sample): the buckets' slack regions are kept too -- PoolPlan::sub_start as the
    // context's last taken sort of this size left it; no sample of the first pass's output, no rooms to size (the second pass verifies
    // them as ever: a bucket out of room flags the sort, which then runs again with samples of its own)
    constexpr uint32_t THREADS = 512, WAVES = THREADS / 64, SUB = 1u << SUBBITS, PER = SUB / 64u;
    __shared__ uint32_t s_c[kBins];              // keys of top byte t
    __shared__ uint32_t s_red[4][WAVES];
    __shared__ uint32_t s_hist[WAVES][SUB];      // sampled keys of this top byte by bucket, one row per wave
    __shared__ uint2 s_piece[16];
    __shared__ uint32_t s_first[17];
    __shared__ uint32_t s_bad;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, a = blockIdx.x;
    // bucket index of a key: (key - key_base) >> bshift (GROUPED: the bits below the top byte; a sort: the probed range's)
    const uint32_t shift = GROUPED ? 18u : pool->shift, bshift = shift + kMsdBits - top_bits - SUBBITS;
    if (!GROUPED && tid < 256u) {  // the first pass's claims: (list x, tile i) exactly once for every tile of the grid (the 256 x 256 first threads take one each)
        const uint32_t w = a * 256u + tid, x = w / kPoolMaxTilesA, i = w % kPoolMaxTilesA;
        if (x < 8u && i < ps.tiles_per_stream) {
            const uint32_t claims = pool->claim_a[w];
            pool->claim_a[w] = 0;
            if (claims != 1u && pool->armed != 0u) __hip_atomic_fetch_or(&pool->fail[par], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // thread t < 256: top byte t's exact total; thread p < 16: piece p of THIS top byte
    uint32_t c_t = 0;
    if (tid < kBins) {
        if constexpr (GROUPED) {
            c_t = groups.count[tid];
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) c_t += msd->cursor_a[s][tid];
        }
        s_c[tid] = c_t;
    }
    uint32_t plen = 0, pslot = 0;
    if (tid < 16u) {
        if constexpr (GROUPED) {
            // (two pieces: the keys in the grouped buffer, the own keys elsewhere; their slots are known behind the sums below)
            plen = tid == 0u ? groups.count[a] - groups.own[a] : tid == 1u ? groups.own[a] : 0u;
        } else {
            const uint32_t s = tid >> 1;
            const uint32_t c = msd->cursor_a[s][a], prim = min(c, pool->cap[s][a]);
            // (a share that outgrew its room -- the first pass flagged the sort, nothing below will be used -- must still not send the
            // sample behind the overflow scratch: a cursor counts what was ASKED for)
            plen = (tid & 1u) ? min(c - prim, pool->ocap[s][a]) : prim;
            pslot = (tid & 1u) ? n + pool->obase[s][a] : pool->base[s][a];
        }
    }
#pragma unroll
    for (uint32_t q = 0; q < PER; ++q) s_hist[wave][lane + 64u * q] = 0;
    if (tid == 0) s_bad = 0;
    // sums over the top bytes before this one: keys, slack space, tiles of the same XCD
    const uint32_t tiles_t = (c_t + kPoolTile - 1u) / kPoolTile;
    uint32_t r0 = tid < a ? c_t : 0u, r1 = tid < a ? pool_space<SUB>(c_t) : 0u, r2 = (tid < a && ((tid ^ a) & 7u) == 0u) ? tiles_t : 0u;
    uint32_t r3 = (GROUPED && tid < a) ? groups.own[tid & 255u] : 0u;  // own keys of the top bytes before this one
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        r0 += __shfl_xor(r0, o);
        r1 += __shfl_xor(r1, o);
        r2 += __shfl_xor(r2, o);
        if constexpr (GROUPED) r3 += __shfl_xor(r3, o);
    }
    if (lane == 0u) {
        s_red[0][wave] = r0;
        s_red[1][wave] = r1;
        s_red[2][wave] = r2;
        s_red[3][wave] = r3;
    }
    if (tid < 16u) {
        // the row of pieces: .x = keys up to and including the piece (a scan over 16 lanes), .y = its first virtual slot; and the
        // pieces' CHUNKS (the leading 256 keys of every 8192): where each piece's chunks start in the list of all
        const uint32_t chunks = (plen + kPoolTile - 1u) / kPoolTile;
        uint32_t pend = plen, cend = chunks;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const uint32_t t = __shfl_up(pend, o), u = __shfl_up(cend, o);
            if (lane >= static_cast<uint32_t>(o)) {
                pend += t;
                cend += u;
            }
        }
        if constexpr (!GROUPED) {
            pool->pieces[a][tid] = make_uint2(pend, pslot);
            s_piece[tid] = make_uint2(plen, pslot);
        }
        s_first[tid] = cend - chunks;
        if (tid == 15u) s_first[16] = cend;
    }
    __syncthreads();
    uint32_t top = 0, part = 0, tiles_before = 0, own_before = 0;
#pragma unroll
    for (uint32_t w = 0; w < WAVES; ++w) {
        top += s_red[0][w];
        part += s_red[1][w];
        tiles_before += s_red[2][w];
        own_before += s_red[3][w];
    }
    const uint32_t c_a = s_c[a];
    if constexpr (GROUPED) {  // the top byte: a piece of `regions` at its place in the grouped keys, then (if any) its own keys elsewhere
        if (tid < 16u) {
            const uint32_t own_a = groups.own[a], here = c_a - own_a, own_slot = n + groups.own_first + own_before;  // (virtual slots from n on: the second buffer)
            pool->pieces[a][tid] = make_uint2(tid == 0u ? here : c_a, tid == 0u ? top : own_slot);
            s_piece[tid] = make_uint2(tid == 0u ? here : tid == 1u ? own_a : 0u, tid == 0u ? top : own_slot);
        }
        if (a == 0u && tid == 0u) pool->fail[par] = 0;  // (a sort's layout kernel re-arms it; nobody sets it before the second pass here)
        __syncthreads();
    }
    {   // the second pass's tile map: thread i = tile i of this top byte (a tile is 8192 positions of the top byte's run of keys)
        const uint32_t tiles_a = (c_a + kPoolTile - 1u) / kPoolTile, x = a & 7u;
        for (uint32_t i = tid; i < tiles_a; i += THREADS) {
            const uint32_t tile_lo = i * kPoolTile;
            uint32_t plo = 0, slot0 = kPoolTileGeneral;
#pragma unroll
            for (uint32_t p = 0; p < 16u; ++p) {  // the piece that holds the tile's first position
                const uint2 pc = s_piece[p];
                if (tile_lo >= plo && tile_lo + kPoolTile <= plo + pc.x) slot0 = pc.y + (tile_lo - plo);  // ... and all of the tile
                plo += pc.x;
            }
            if (tiles_before + i < kPoolMaxTilesB) pool->tile_map[x][tiles_before + i] = make_uint2(slot0, a | (i << 8));
        }
    }
    // The sample: thread (h, t) = (tid / 256, tid % 256) reads key t of the chunks q = h, h + 2, ..., 32 of them in flight at a time
    // (a piece after the other would be sixteen dependent round trips).  Chunk q belongs to the last piece whose first chunk is <= q.
    const uint32_t chunks_all = keep ? 0u : s_first[16], half = tid >> 8, t = tid & 255u;
    for (uint32_t q0 = half; q0 < chunks_all; q0 += 64u) {
        uint32_t k[32], live = 0;
#pragma unroll
        for (uint32_t u = 0; u < 32u; ++u) {
            const uint32_t q = min(q0 + 2u * u, chunks_all - 1u);
            uint32_t p = 0;
#pragma unroll
            for (uint32_t step = 8; step >= 1; step >>= 1)
                if (s_first[p + step] <= q) p += step;  // (empty pieces share their successor's first chunk: skipped)
            const uint2 pc = s_piece[p];
            const uint32_t idx = (q - s_first[p]) * kPoolTile + t;
            live |= (q0 + 2u * u < chunks_all && idx < pc.x) ? 1u << u : 0u;
            const uint32_t v = pc.y + (idx < pc.x ? idx : 0u);
            k[u] = *(v < n ? regions + v : overflow + (v - n));
        }
#pragma unroll
        for (uint32_t u = 0; u < 32u; ++u)
            if (live & (1u << u)) atomicAdd(&s_hist[wave][((k[u] - key_base) >> bshift) & (SUB - 1u)], 1u);
    }
    __syncthreads();
    const uint32_t top_bytes = GROUPED ? groups.top_bytes : 1u << top_bits;  // (the tables hold top_bytes << SUBBITS buckets: the workgroups behind them have none)
    if (wave == 0u && a < top_bytes && keep) {  // the regions stay: the second pass counts from zero
#pragma unroll
        for (uint32_t q = 0; q < PER; ++q) pool->sub_cursor[a * SUB + PER * lane + q] = 0;
    } else if (wave == 0u && a < top_bytes) {  // lane l = buckets [PER l, PER l + PER) of the top byte
        uint32_t m_b[PER], m = 0;
#pragma unroll
        for (uint32_t q = 0; q < PER; ++q) {
            m_b[q] = 0;
#pragma unroll
            for (uint32_t w = 0; w < WAVES; ++w) m_b[q] += s_hist[w][PER * lane + q];
            m += m_b[q];
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m += __shfl_xor(m, o);
        const float total = static_cast<float>(c_a);
        uint32_t room[PER], rooms = 0;
#pragma unroll
        for (uint32_t q = 0; q < PER; ++q) {
            const float est = m ? floorf(total * (static_cast<float>(m_b[q]) / static_cast<float>(m)) * 0.999999f) : 0.0f;
            const float want = fminf(est + kPoolSigmas * sqrtf(kPoolR * (est + kPoolR)), total) + static_cast<float>(kPoolRoomFloor);
            room[q] = c_a ? (static_cast<uint32_t>(want) + 3u) & ~3u : 0u;
            rooms += room[q];
        }
        uint32_t incl = rooms;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(incl, o);
            if (lane >= static_cast<uint32_t>(o)) incl += u;
        }
        uint32_t at = part + incl - rooms;
#pragma unroll
        for (uint32_t q = 0; q < PER; ++q) {
            const uint32_t b = a * SUB + PER * lane + q;
            pool->sub_start[b] = at;
            pool->sub_cursor[b] = 0;  // the second pass counts from zero
            at += room[q];
        }
        if (a == top_bytes - 1u && lane == 63u) pool->sub_start[top_bytes * SUB] = at;
    }
    if (tid == 0) {
        pool->top_base[a] = top;
        pool->tiles_b[a & 7u][a >> 3] = tiles_before;
        if (a >= 248u) pool->tiles_b[a & 7u][32] = tiles_before + (c_a + kPoolTile - 1u) / kPoolTile;
    }
    if (a == 255u) {  // verdict 1 (every total is a function of the cursors: this workgroup has them all)
        uint32_t sum = c_t, space = pool_space<SUB>(c_t);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            sum += __shfl_xor(sum, o);
            space += __shfl_xor(space, o);
        }
        if (lane == 0u) {  // (the sums above were read before the barrier in front of the sample's counting)
            s_red[0][wave] = sum;
            s_red[1][wave] = space;
        }
        if (tid < 8u) {  // XCD tid's tiles
            uint32_t acc = 0;
            for (uint32_t k = 0; k < 32u; ++k) acc += (s_c[tid + 8u * k] + kPoolTile - 1u) / kPoolTile;
            if (acc > tiles_b_cap) atomicOr(&s_bad, 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t all = 0, room = 0;
#pragma unroll
            for (uint32_t w = 0; w < WAVES; ++w) {
                all += s_red[0][w];
                room += s_red[1][w];
            }
            pool->top_base[256] = all;
            // all != n: keys the first pass did not place (it did not run, or a workgroup left early); the last kPoolTile slots of the
            // slack buffer are where refused runs are dumped
            if constexpr (GROUPED)
                pool->ok_a = (s_bad == 0u && all == n && room <= slack_capacity - kPoolTile) ? 1u : 0u;  // (all != n: the caller's counts are not these keys')
            else
                pool->ok_a = (pool->armed != 0u && pool->fail[par] == 0u && s_bad == 0u && all == n && shift >= kPoolMinShift && shift <= kPoolMaxShift &&
                              (keep != 0u || room <= slack_capacity - kPoolTile))  // (kept regions: the table that fit this buffer when it was made)
                                 ? 1u
                                 : 0u;
            // (PoolPlan::fail stays: the second pass may still set it; the next sort's layout kernel re-arms it)
            pool->max_bucket = 0;
            pool->fail[par ^ 1u] = 0;  // the NEXT sort's word (its first pass may be its first kernel)
            if constexpr (GROUPED) pool->shift = shift;
            msd->shift = shift;
            msd->sub_bits = SUBBITS;
            msd->ok = 0;  // the local sort decides
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Second pass: block b -> XCD b % 8, which walks the shares (top byte x + 8 k, slice s) in entry order e = 8 k + s.  A tile reads
// 8192 slots of its share -- the share's positions run through its primary region, then its overflow region -- and scatters them by
// the 6 bits below the top byte into the buckets' slack regions.
// SlackReserve: StreamReserve's one atomic add per tile and digit -- all tiles of a top byte run behind the L2 that holds its 64
// cursors -- with the region's capacity looked at: a run that does not fit (its region, or the local sort that follows) flags the sort
// and is dumped in the buffer's last tile.  The write-out also checks every key against the probed range.
struct SlackReserve {
    static constexpr bool kEnabled = true;
    static constexpr bool kReserves = true;
    static constexpr bool kPool = true;
    bool foreign = false;            // (interface of StreamLookback: a workgroup off its XCD never gets this far)
    uint32_t recounted = 0;
    int index = 0;
    const void *stream_keys = nullptr;
    uint32_t done = 0;
    uint32_t seed = 0;
    uint32_t *cursor = nullptr;      // this thread's bucket's cursor
    uint32_t start = 0, cap = 0;     // the bucket's region: first slot, its slots
    uint32_t local_cap = 0;          // keys the enqueued local sort takes per bucket
    uint32_t *max_bucket = nullptr;
    uint32_t dump = 0;               // first slot of the dump tile
    uint32_t pad_keys = 0;
    uint32_t above = 0, key_base = 0;  // bits no key of the probed range has
    uint32_t *fail_word = nullptr;
    mutable uint32_t reserved = 0, cnt = 0;
    mutable bool reserved_yet = false;

    __device__ __forceinline__ void publish(uint32_t v) const {
        if (reserved_yet) return;  // the second call (the inclusive prefix) has nobody to tell
        reserved_yet = true;
        cnt = v - pad_keys;
        if (cnt) reserved = __hip_atomic_fetch_add(cursor, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ void fetch(int, uint32_t (&)[kLbBatch]) const {}
    __device__ __forceinline__ uint32_t resolve(uint32_t (&)[kLbBatch], bool &) const { return reserved; }
    __device__ __forceinline__ void refuse() const {}  // (a reservation never waits)
    // run [reserved, reserved + cnt) of the region; excl: where the digit's run starts inside the tile
    __device__ __forceinline__ void place(uint32_t *gbase, uint32_t tid, uint32_t, uint32_t excl) const { place_run(gbase, tid, reserved, excl); }
    __device__ __forceinline__ void place_run(uint32_t *gbase, uint32_t tid, uint32_t reserved, uint32_t excl) const {
        const uint32_t end = reserved + cnt;
        const bool bad = cnt != 0u && end > cap;
        if (bad) {
            __hip_atomic_fetch_or(fail_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (cnt != 0u && end > local_cap) {
            // in its region, but more than the local sort enqueued behind this pass takes: that one leaves at once, and a larger one
            // can finish the sort (the host asks for it when it sees this)
            __hip_atomic_fetch_or(fail_word, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_max(max_bucket, end, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        gbase[tid] = (bad ? dump : start + reserved) - excl;
    }
    template <typename K, int ITEMS, uint32_t THREADS, bool FULL, typename DG>
    __device__ __forceinline__ void store(const uint32_t *, const K (&key)[ITEMS], uint32_t (&dst)[ITEMS], K *kout, uint32_t valid, const DG &) const {
        const uint32_t tid = threadIdx.x;
        uint32_t over = 0;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            if (FULL || i * THREADS + tid < valid) {  // (the padding key of a ragged tile is no key)
                over |= (key[i] - key_base) & above;
                kout[dst[i]] = key[i];
            }
        }
        // a key above the probed range (or below the promised floor): the local sort, which gives the last verdict, sees this
        if (__ballot(over != 0u) != 0ull && (tid & 63u) == 0u) __hip_atomic_fetch_or(fail_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    template <int ITEMS, uint32_t THREADS, bool FULL>
    __device__ __forceinline__ void store_values(const uint32_t (&val)[ITEMS], const uint32_t (&dst)[ITEMS], uint32_t *vout, uint32_t valid) const {
        const uint32_t tid = threadIdx.x;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
            if (FULL || i * THREADS + tid < valid) vout[dst[i]] = val[i];
    }
};

// The STABLE second pass (pairs): a tile's place in a bucket's region is its rank in the bucket -- the keys of the top byte's earlier
// tiles with that digit: one look-back chain per top byte (the counted form's second pass, msd_pass_b_kernel, has the same chains).
// The buckets' cursors are still added to: the local sort reads the buckets' sizes there.
struct SlackLookback : StreamLookback {
    static constexpr bool kPool = true;
    SlackReserve at;
    __device__ __forceinline__ void publish(uint32_t v) const {
        if (!at.reserved_yet) {
            at.reserved_yet = true;
            at.cnt = v - at.pad_keys;
            if (at.cnt) __hip_atomic_fetch_add(at.cursor, at.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        StreamLookback::publish(v);
    }
    __device__ __forceinline__ void refuse() const { __hip_atomic_fetch_or(at.fail_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ void place(uint32_t *gbase, uint32_t tid, uint32_t before, uint32_t excl) const { at.place_run(gbase, tid, before, excl); }
    template <typename K, int ITEMS, uint32_t THREADS, bool FULL, typename DG>
    __device__ __forceinline__ void store(const uint32_t *g, const K (&key)[ITEMS], uint32_t (&dst)[ITEMS], K *kout, uint32_t valid, const DG &dg) const {
        at.template store<K, ITEMS, THREADS, FULL>(g, key, dst, kout, valid, dg);
    }
    template <int ITEMS, uint32_t THREADS, bool FULL>
    __device__ __forceinline__ void store_values(const uint32_t (&val)[ITEMS], const uint32_t (&dst)[ITEMS], uint32_t *vout, uint32_t valid) const {
        at.template store_values<ITEMS, THREADS, FULL>(val, dst, vout, valid);
    }
};

template <uint32_t SUBBITS, bool PAIRS>
__global__ __launch_bounds__(512, PAIRS ? 4 : 6) void pool_pass_b_kernel(const uint32_t *__restrict__ regions, const uint32_t *__restrict__ overflow,
                                                             uint32_t *__restrict__ slack, const MsdPlan *__restrict__ msd, PoolPlan *__restrict__ pool,
                                                             uint32_t n_virt, uint32_t key_base, uint32_t local_cap, uint32_t dump,
                                                             unsigned long long xcc_map, uint32_t stamp, uint32_t grouped, uint32_t par, PoolPayloads pv,
                                                             uint32_t top_bits) {
    __shared__ ChunkSmem<uint32_t, 16, 8, PAIRS> sm;
    // the list follows the XCC this workgroup RUNS on (pool_pass_a_kernel): all tiles of a top byte then meet behind the L2 that
    // holds its 64 cursors, whatever the dispatcher's rotation
    const uint32_t x = xcc_place(xcc_map), j = blockIdx.x >> 3;
    if (pool->ok_a == 0u) return;  // uniform (enqueued before the plan was known: it may have said no)
    if (x == 8u) {  // behind an L2 the probe never saw
        if (threadIdx.x == 0) __hip_atomic_fetch_or(&pool->fail[par], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // the claim: this sort's stamp into (list, tile)'s word -- asked for now, looked at when the tile's loads are under way
    uint32_t claimed = 0;
    if (threadIdx.x == 0) claimed = __hip_atomic_exchange(&pool->claim_b[j * 8u + x], stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (L2-local: a second claim comes from this XCC too)
    // which tile: one word pair of the plan kernel's map (its second word: the top byte and the tile's index in it)
    const uint2 entry = pool->tile_map[x][j < kPoolMaxTilesB ? j : 0u];
    if (j >= pool->tiles_b[x][32]) return;
    const uint32_t a = entry.y & 255u, tile_lo = (entry.y >> 8) * kPoolTile;
    const bool one_piece = entry.x != kPoolTileGeneral;  // uniform: a full tile inside one piece -- its loads wait for nothing else
    // The top byte's keys lie in 16 PIECES: slice s's primary region (piece 2 s), then its overflow region (2 s + 1).  A tile that
    // touches several: every wave, lane p = piece p -- its keys, its first (virtual) slot, where it starts in the top byte's run of keys
    // (the plan kernel's row of the top byte: 128 bytes, one load per wave).
    const uint32_t lane = threadIdx.x & 63u;
    uint2 pc = make_uint2(0, 0);
    if (!one_piece) pc = pool->pieces[a][lane & 15u];
    const uint32_t shift = pool->shift;
    constexpr uint32_t SUB = 1u << SUBBITS;
    const uint32_t d = threadIdx.x & 255u, b = a * SUB + min(d, SUB - 1u);
    SlackReserve at;
    at.cursor = &pool->sub_cursor[b];
    if (threadIdx.x < SUB) {  // (the threads that reserve: one per bucket of the top byte)
        at.start = pool->sub_start[b];
        at.cap = pool->sub_start[b + 1u] - at.start;
    }
    at.local_cap = local_cap;
    at.max_bucket = &pool->max_bucket;
    const uint32_t pend = pc.x, pslot = pc.y;
    const uint32_t before = __shfl_up(pend, 1);
    const uint32_t plo = lane == 0u ? 0u : before;  // the piece holds positions [plo, pend) of the top byte
    const uint32_t plen = lane < 16u ? pend - plo : 0u;
    const uint32_t keys_a = __builtin_amdgcn_readlane(pend, 15);
    const uint32_t valid = one_piece ? kPoolTile : min(kPoolTile, keys_a - tile_lo);
    const unsigned long long touch = __ballot(plen != 0u && plo < tile_lo + valid && pend > tile_lo);
    PieceSrc src;
    src.p0 = static_cast<uint32_t>(__builtin_ctzll(touch | (1ull << 63)));
    src.p1 = 64u - static_cast<uint32_t>(__builtin_clzll(touch | 1ull));
    src.lo = plo - tile_lo;
    src.len = plen;
    src.slot = pslot;
    src.regions = regions;
    src.overflow = overflow;
    src.n_virt = n_virt;
    const uint32_t slot0 = one_piece ? entry.x : __builtin_amdgcn_readlane(pslot, src.p0) + (tile_lo - __builtin_amdgcn_readlane(plo, src.p0));  // the tile's first key
    src.first_slot = slot0;
    const BitsDigit dg{shift + kMsdBits - top_bits - SUBBITS, SUB - 1u, key_base};  // (key_base is a multiple of 2^24 and the bits end at or below bit 24)
    at.dump = dump;
    at.pad_keys = d == SUB - 1u ? kPoolTile - valid : 0u;  // (the padding key, key_base - 1, carries the largest digit)
    // a sort: no key may have bits above the probed range; grouped keys (the caller's promise): every key of this tile carries top byte a
    at.above = grouped ? 0xFF000000u : (shift + kMsdBits < 32u ? ~0u << (shift + kMsdBits) : 0u);
    at.key_base = grouped ? key_base + (a << 24) : key_base;
    at.fail_word = &pool->fail[par];
    uint32_t unused = 0;
    // (two workgroups of one group of eight on ONE XCC: the tile has been taken twice and another not at all)
    if (threadIdx.x == 0 && claimed == stamp) __hip_atomic_fetch_or(&pool->fail[par], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool stream_in = static_cast<size_t>(pool->top_base[256]) * sizeof(uint32_t) >= kStreamInBytes;
    const uint32_t *kin = slot0 < n_virt ? regions + slot0 : overflow + (slot0 - n_virt);
    if constexpr (PAIRS) {
        // payloads: the tile's place in a bucket is its rank there (SlackLookback) -- the chain of top byte a: the rows of its tiles
        // in list x, 8 rows apart (tile i of the top byte is tile j of the list)
        src.vregions = pv.values_partner;
        src.voverflow = pv.overflow_values;
        SlackLookback lb;
        lb.at = at;
        lb.col = pv.status + (static_cast<size_t>(j - (entry.y >> 8)) * 8u + x) * kBins + d;
        lb.stride = static_cast<size_t>(8) * kBins;
        lb.index = static_cast<int>(entry.y >> 8);
        lb.tag = 6u << kLbTagShift;
        lb.budget = pv.spin_budget;
        const uint32_t *vin = slot0 < n_virt ? pv.values_partner + slot0 : pv.overflow_values + (slot0 - n_virt);
        if (one_piece)
            scatter_chunk<uint32_t, 16, 8, true, RANK_ATOMIC, true, BitsDigit, SlackLookback>(sm, kin, vin, slack, pv.slack_values, valid, dg, unused, lb, NoPieces{}, stream_in);
        else if (valid == kPoolTile)
            scatter_chunk<uint32_t, 16, 8, true, RANK_ATOMIC, true, BitsDigit, SlackLookback, PieceSrc>(sm, nullptr, nullptr, slack, pv.slack_values, valid, dg, unused, lb, src);
        else
            scatter_chunk<uint32_t, 16, 8, true, RANK_ATOMIC, false, BitsDigit, SlackLookback, PieceSrc>(sm, nullptr, nullptr, slack, pv.slack_values, valid, dg, unused, lb, src);
    } else {
        if (one_piece)  // five tiles in six
            scatter_chunk<uint32_t, 16, 8, false, RANK_ATOMIC, true, BitsDigit, SlackReserve>(sm, kin, nullptr, slack, nullptr, valid, dg, unused, at, NoPieces{}, stream_in);
        else if (valid == kPoolTile)
            scatter_chunk<uint32_t, 16, 8, false, RANK_ATOMIC, true, BitsDigit, SlackReserve, PieceSrc>(sm, nullptr, nullptr, slack, nullptr, valid, dg, unused, at, src);
        else  // the top byte's ragged last tile
            scatter_chunk<uint32_t, 16, 8, false, RANK_ATOMIC, false, BitsDigit, SlackReserve, PieceSrc>(sm, nullptr, nullptr, slack, nullptr, valid, dg, unused, at, src);
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host side

hipError_t launch_pool_sample(hipStream_t stream, const uint32_t *keys, uint32_t n, uint32_t key_base, const PoolStreams &ps,
                              PoolPlan *pool, uint32_t overflow_capacity, uint32_t par, LaunchEvents ev, uint32_t top_bits) {
    if (n == 0 || ps.tiles_per_stream < kPoolSampleTiles) return hipErrorInvalidValue;  // (a sample workgroup's tiles span at most two slices)
    const uint32_t grid = (ps.tiles_total + kPoolSampleTiles - 1u) / kPoolSampleTiles;
    VRS_LAUNCH(pool_sample_kernel, dim3(grid), dim3(256), stream, ev, keys, n, key_base, ps, pool, top_bits);
    hipLaunchKernelGGL(pool_layout_kernel, dim3(1), dim3(256), 0, stream, ps, pool, overflow_capacity, par);
    return hipGetLastError();
}

hipError_t launch_pool_pass_a(hipStream_t stream, const uint32_t *keys_in, uint32_t *keys_out, uint32_t *overflow, uint32_t n,
                              uint32_t key_base, const PoolStreams &ps, PoolPlan *pool, MsdPlan *msd, unsigned long long xcc_map,
                              bool misplace, uint32_t overflow_capacity, uint32_t par, LaunchEvents ev, const PoolPayloads *pv, uint32_t top_bits) {
    if (pv)
        VRS_LAUNCH(pool_pass_a_kernel<true>, dim3(8u * ps.tiles_per_stream), dim3(512), stream, ev, keys_in, keys_out, overflow, n, key_base, ps, pool, msd,
                   xcc_map, 0, overflow_capacity, par, *pv, top_bits);
    else
        VRS_LAUNCH(pool_pass_a_kernel<false>, dim3(8u * ps.tiles_per_stream), dim3(512), stream, ev, keys_in, keys_out, overflow, n, key_base, ps, pool, msd,
                   xcc_map, misplace ? 1 : 0, overflow_capacity, par, PoolPayloads{}, top_bits);
    return hipGetLastError();
}

hipError_t launch_pool_plan(hipStream_t stream, MsdPlan *msd, PoolPlan *pool, uint32_t n, uint32_t tiles_b_cap, uint32_t slack_capacity,
                            const uint32_t *regions, const uint32_t *overflow, uint32_t key_base, const PoolStreams &ps, uint32_t sub_bits,
                            uint32_t par, const PoolGroups *groups, bool keep_rooms, uint32_t top_bits) {
    if (ps.tiles_per_stream > kPoolMaxTilesA || tiles_b_cap > kPoolMaxTilesB || (groups && keep_rooms)) return hipErrorInvalidValue;
    const dim3 grid(256), block(512);
    const uint32_t keep = keep_rooms ? 1u : 0u;
    if (groups) {
        if (sub_bits == 8u) hipLaunchKernelGGL((pool_plan_kernel<8, true>), grid, block, 0, stream, msd, pool, n, tiles_b_cap, slack_capacity, regions, overflow, key_base, ps, *groups, par, 0u, 8u);
        else if (sub_bits == 7u) hipLaunchKernelGGL((pool_plan_kernel<7, true>), grid, block, 0, stream, msd, pool, n, tiles_b_cap, slack_capacity, regions, overflow, key_base, ps, *groups, par, 0u, 8u);
        else hipLaunchKernelGGL((pool_plan_kernel<6, true>), grid, block, 0, stream, msd, pool, n, tiles_b_cap, slack_capacity, regions, overflow, key_base, ps, *groups, par, 0u, 8u);
    } else {
        const PoolGroups none{};
        if (sub_bits == 8u) hipLaunchKernelGGL((pool_plan_kernel<8, false>), grid, block, 0, stream, msd, pool, n, tiles_b_cap, slack_capacity, regions, overflow, key_base, ps, none, par, keep, top_bits);
        else if (sub_bits == 7u) hipLaunchKernelGGL((pool_plan_kernel<7, false>), grid, block, 0, stream, msd, pool, n, tiles_b_cap, slack_capacity, regions, overflow, key_base, ps, none, par, keep, top_bits);
        else hipLaunchKernelGGL((pool_plan_kernel<6, false>), grid, block, 0, stream, msd, pool, n, tiles_b_cap, slack_capacity, regions, overflow, key_base, ps, none, par, keep, top_bits);
    }
    return hipGetLastError();
}

hipError_t launch_pool_pass_b(hipStream_t stream, const uint32_t *regions, const uint32_t *overflow, uint32_t *slack, uint32_t n, MsdPlan *msd,
                              PoolPlan *pool, uint32_t tiles_b, uint32_t key_base, uint32_t local_cap, uint32_t slack_capacity,
                              unsigned long long xcc_map, uint32_t stamp, uint32_t sub_bits, uint32_t par, LaunchEvents ev, bool grouped,
                              const PoolPayloads *pv, uint32_t top_bits) {
    if (tiles_b == 0) return hipSuccess;
    if (tiles_b > kPoolMaxTilesB || stamp == 0u) return hipErrorInvalidValue;
    // (grouped keys lie in `regions` alone: no slot is an overflow slot)
    // (... unless a part of them lies in a second buffer, PoolGroups::own: those are the slots from n on)
    const uint32_t n_virt = (grouped && regions == overflow) ? 0xFFFFFFFFu : n, g = grouped ? 1u : 0u;
    if (pv) {  // pairs: six bits, a sort
        if ((sub_bits != 6u && sub_bits != 7u && sub_bits != 8u) || grouped) return hipErrorInvalidValue;
        if (sub_bits == 8u) {
            VRS_LAUNCH((pool_pass_b_kernel<8, true>), dim3(8u * tiles_b), dim3(512), stream, ev, regions, overflow, slack, msd, pool, n_virt, key_base, local_cap,
                       slack_capacity - kPoolTile, xcc_map, stamp, g, par, *pv, top_bits);
            return hipGetLastError();
        }
        if (sub_bits == 7u) {
            VRS_LAUNCH((pool_pass_b_kernel<7, true>), dim3(8u * tiles_b), dim3(512), stream, ev, regions, overflow, slack, msd, pool, n_virt, key_base, local_cap,
                       slack_capacity - kPoolTile, xcc_map, stamp, g, par, *pv, top_bits);
            return hipGetLastError();
        }
        VRS_LAUNCH((pool_pass_b_kernel<6, true>), dim3(8u * tiles_b), dim3(512), stream, ev, regions, overflow, slack, msd, pool, n_virt, key_base, local_cap,
                   slack_capacity - kPoolTile, xcc_map, stamp, g, par, *pv, top_bits);
        return hipGetLastError();
    }
#define VRS_POOL_B(S)                                                                                                                         \
    VRS_LAUNCH((pool_pass_b_kernel<S, false>), dim3(8u * tiles_b), dim3(512), stream, ev, regions, overflow, slack, msd, pool, n_virt, key_base, local_cap, \
               slack_capacity - kPoolTile, xcc_map, stamp, g, par, PoolPayloads{}, top_bits)
    if (sub_bits == 8u) VRS_POOL_B(8);
    else if (sub_bits == 7u) VRS_POOL_B(7);
    else VRS_POOL_B(6);
#undef VRS_POOL_B
    return hipGetLastError();
}

}  // namespace vrs
