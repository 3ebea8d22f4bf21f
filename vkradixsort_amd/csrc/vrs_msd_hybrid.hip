// vrs_msd_hybrid.hip -- K5b, the hybrid form's own kernels: the plan of the MSD partition (msd_plan_kernel), the 64-bit counting read,
// the second MSD pass (msd_pass_b_kernel) and the LDS-local sorts of the buckets (keys / pairs / 64-bit keys / one wave per
// bucket).  The counting read and the first MSD pass are K5's (vrs_one_call.hip).  No reference counterpart: VkRadixSort runs
// four LSD passes whatever the size (MultiRadixSort.cpp:50-61).
#include "vrs_device.hpp"
#include "vrs_local_sort.hpp"
#include "vrs_plan.hpp"

#include <algorithm>
#include <type_traits>

namespace vrs {

constexpr int kLocalThreads = 256, kLocalItems = 26;  // local sort of pairs: capacity 6656 per bucket
constexpr uint32_t kLocalCap = kLocalThreads * kLocalItems;

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
        const bool stream_in = static_cast<size_t>(n) * sizeof(uint64_t) >= kStreamInBytes;  // (a read-only pass over keys beyond the caches: vrs_device.hpp)
        for (; i0 + THREADS * UNROLL <= nvec; i0 += THREADS * UNROLL) {
            ulonglong2 q[UNROLL];
#pragma unroll
            for (uint32_t r = 0; r < UNROLL; ++r) q[r] = stream_in ? load_stream16(v + i0 + r * THREADS + tid) : v[i0 + r * THREADS + tid];
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
                                                            uint32_t sub_bits, uint32_t *drift) {
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
    report_drift(drift, xcc_map);
    const uint32_t first = msd->base[a << sub_bits], last = msd->base[(a + 1u) << sub_bits];
    const uint32_t done = i * kTile;
    const uint32_t begin = first + done;
    const uint32_t valid = min(kTile, last - begin);
    const bool stream_in = static_cast<size_t>(msd->base[kMsdBuckets]) * sizeof(K) >= kStreamInBytes;  // (all keys of the sort: the pass's input)
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
            scatter_chunk<K, ITEMS, 8, PAIRS, RANK, true>(sm, keys_in + begin, vin, keys_out, values_out, valid, dg, unused, lb, NoPieces{}, stream_in);
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
            scatter_chunk<K, ITEMS, 8, PAIRS, RANK, true>(sm, keys_in + begin, vin, keys_out, values_out, valid, dg, unused, lb, NoPieces{}, stream_in);
        else
            scatter_chunk<K, ITEMS, 8, PAIRS, RANK, false>(sm, keys_in + begin, vin, keys_out, values_out, valid, dg, unused, lb);
    }
}

// (StatusClear, clear_status_share: vrs_local_sort.hpp)
// ... and it re-arms the reservation counters of the MSD passes (MsdPlan::cursor_* / back_*): workgroup b = bucket b clears the
// second pass's counters of its bucket, the first 2 * kStreams workgroups one row each of the first pass's.
// Workgroup -> bucket of the local sorts: the LAST bucket first.  The second MSD pass writes the buckets in ascending order, so the
// highest ones are what the memory-side cache (256 MB) still holds when the local sort starts; read in ascending order they would be
// evicted by the time their turn comes (10^8 keys: the pool form's sort 214 -> 207 us, the counted form's 164 -> 160 and the counting read
// behind it 138 -> 131; below 3e7 keys everything fits the cache either way; labs/r04_pool_form.txt).
__device__ __forceinline__ uint32_t local_sort_bucket_of_block() {
    return gridDim.x - 1u - blockIdx.x;
}

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

// (local_pass, local_sort_bucket: vrs_local_sort.hpp -- the pool form's pairs use them too)
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
    // Two copies of the rest.  The common one is inlined and carries no trace of the guard; the guarded one is a CALL that loads
    // the bucket again -- kept out of line so that its register demand cannot push the common path into scratch spills (spills
    // are HBM traffic: with both inlined the kernel moved 1032 instead of 800 MB per launch at 10^8 keys).
    if (guards == 0u) lean_sort_body<THREADS, VEC, false>(k, abase, mis, n, s_keys, s_hist2, s_tmp, false, false, mis);
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
    lean_sort_body<THREADS, VEC, true>(k, abase, mis, n, s_keys, s_hist2, s_tmp, (guards & 1u) != 0u, (guards & 2u) != 0u, mis);
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
    const uint32_t bkt = local_sort_bucket_of_block();
    const uint32_t begin = msd->base[bkt], n = msd->base[bkt + 1] - begin;
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
// (wave_phase, wave_scan512, wave_sort_body: vrs_local_sort.hpp)

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
    if (skew == 0u) wave_sort_body<VEC, false>(k, abase, mis, n, s_keys, tbl, false, false, mis);
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
    wave_sort_body<VEC, true>(k, abase, mis, n, s_keys, tbl, (skew & 1u) != 0u, (skew & 2u) != 0u, mis);
}

constexpr uint32_t kWaveCap = 64u * 4u * kLeanMaxVec - 3u;  // 1789 keys
__global__ __launch_bounds__(64, 4) void msd_local_sort_wave_kernel(uint32_t *__restrict__ keys, const MsdPlan *__restrict__ msd, StatusClear sc,
                                                                  uint32_t *__restrict__ cursors) {
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[64 * 4 * kLeanMaxVec + 4];
    __shared__ __attribute__((aligned(16))) uint32_t s_tbl[kLeanRow];
    if (msd->ok == 0u) return;
    rearm_reservation(cursors, 64);
    clear_status_share(sc, 64);
    const uint32_t bkt = local_sort_bucket_of_block();
    const uint32_t begin = msd->base[bkt], n = msd->base[bkt + 1] - begin;
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
    const uint32_t bkt = local_sort_bucket_of_block();
    const uint32_t begin = msd->base[bkt], n = msd->base[bkt + 1] - begin;
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
    const uint32_t bkt = local_sort_bucket_of_block();
    const uint32_t begin = msd->base[bkt], n = msd->base[bkt + 1] - begin;
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


// 64-bit keys WITH uint32 payloads (vrs_sort_pairs_u64; the reference's SORT_64_BIT stub, MultiRadixSort.h:10-18, has neither): the same
// passes as local_sort_bucket_u64, every one of them STABLE -- the two MSD passes in front are (look-back), and std::stable_sort is
// what the result is compared with -- and the payloads follow their keys through LDS.  512 threads x 8 elements (4096 per bucket: 64 KB
// of LDS, two workgroups per CU) or x 13 (6656: 96 KB, one per CU).
template <int THREADS, int ITEMS>
__device__ __forceinline__ void local_sort_bucket_pairs_u64(uint64_t *bucket, uint32_t *bvals, uint32_t n, uint32_t passes, uint64_t *s_keys,
                                                            uint32_t *s_vals, uint32_t *s_hist, uint32_t *s_tmp) {
    constexpr int BITS = 9;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint64_t key[ITEMS];
    uint32_t val[ITEMS];
    const uint32_t seg = wave * (ITEMS * 64) + lane;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t idx = seg + i * 64;
        key[i] = bucket[idx < n ? idx : n - 1u];
        val[i] = bvals[idx < n ? idx : n - 1u];
    }
    if (passes == 0u) return;  // one distinct key per bucket: the order the stable MSD passes left is the answer
    bool sorted = false;
    if (passes > 4u) {
        // more than 36 low bits: the TOP four digits first, then a look at the neighbours (local_sort_bucket_u64) -- equal keys are
        // in their arrival order either way, every pass being stable
        for (uint32_t p_ = passes - 4u; p_ < passes; ++p_)
            local_pass<THREADS, ITEMS, BITS, true, true, uint64_t>(key, val, s_keys, s_vals, s_hist, s_tmp, BITS * p_, n);
        int bad = 0;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t idx = seg + i * 64;
            if (idx + 1u < n) bad |= key[i] > s_keys[idx + 1u] ? 1 : 0;  // s_keys still holds what the last pass left
        }
        sorted = __syncthreads_or(bad) == 0;
    }
    if (!sorted)
        for (uint32_t p_ = 0; p_ < passes; ++p_)
            local_pass<THREADS, ITEMS, BITS, true, true, uint64_t>(key, val, s_keys, s_vals, s_hist, s_tmp, BITS * p_, n);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t idx = seg + i * 64;
        if (idx < n) {
            bucket[idx] = key[i];
            bvals[idx] = val[i];
        }
    }
}

constexpr int kLocalPairsU64Threads = 512, kLocalPairsU64Small = 8, kLocalPairsU64Big = 13;
constexpr uint32_t kLocalCapPairsU64Small = kLocalPairsU64Threads * kLocalPairsU64Small;  // 4096
constexpr uint32_t kLocalCapPairsU64 = kLocalPairsU64Threads * kLocalPairsU64Big;         // 6656
template <int ITEMS_MAX>
__global__ __launch_bounds__(kLocalPairsU64Threads, 2) void msd_local_sort_pairs_u64_kernel(uint64_t *__restrict__ keys, uint32_t *__restrict__ values,
                                                                                           const MsdPlan *__restrict__ msd, StatusClear sc,
                                                                                           uint32_t *__restrict__ cursors) {
    constexpr int THREADS = kLocalPairsU64Threads, WAVES = THREADS / 64;
    constexpr uint32_t CAP = THREADS * ITEMS_MAX;
    __shared__ uint64_t s_keys[CAP];
    __shared__ uint32_t s_vals[CAP];
    __shared__ uint32_t s_hist[WAVES << 9];
    __shared__ uint32_t s_tmp[1 + WAVES];
    if (msd->ok == 0u) return;
    rearm_reservation(cursors, THREADS);
    clear_status_share(sc, THREADS);
    const uint32_t bkt = local_sort_bucket_of_block();
    const uint32_t begin = msd->base[bkt], n = msd->base[bkt + 1] - begin;
    if (n == 0 || n > CAP) return;
    const uint32_t passes = (msd->shift + 8u) / 9u;
    uint64_t *bucket = keys + begin;
    uint32_t *bvals = values + begin;
    const uint32_t used = (n + THREADS - 1u) / THREADS;
    if (used <= 2) local_sort_bucket_pairs_u64<THREADS, 2>(bucket, bvals, n, passes, s_keys, s_vals, s_hist, s_tmp);
    else if (used <= 4) local_sort_bucket_pairs_u64<THREADS, 4>(bucket, bvals, n, passes, s_keys, s_vals, s_hist, s_tmp);
    else if (used <= 6) local_sort_bucket_pairs_u64<THREADS, 6>(bucket, bvals, n, passes, s_keys, s_vals, s_hist, s_tmp);
    else if (used <= 8 || ITEMS_MAX <= 8) local_sort_bucket_pairs_u64<THREADS, 8>(bucket, bvals, n, passes, s_keys, s_vals, s_hist, s_tmp);
    else if (used <= 10) local_sort_bucket_pairs_u64<THREADS, ITEMS_MAX >= 10 ? 10 : ITEMS_MAX>(bucket, bvals, n, passes, s_keys, s_vals, s_hist, s_tmp);
    else local_sort_bucket_pairs_u64<THREADS, ITEMS_MAX>(bucket, bvals, n, passes, s_keys, s_vals, s_hist, s_tmp);
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
                             uint32_t sub_bits, bool reserve, uint32_t *drift) {
    if (tiles_b == 0) return hipSuccess;
    if (sub_bits < 6u || sub_bits > 8u) return hipErrorInvalidValue;
    const dim3 grid(8 * tiles_b), block(512);
#define VRS_PASS_B(K, ITEMS, RANK, PAIRS, RESERVE)                                                                         \
    VRS_LAUNCH((msd_pass_b_kernel<K, ITEMS, RANK, PAIRS, RESERVE>), grid, block, stream, ev, static_cast<const K *>(keys_in), \
               static_cast<K *>(keys_out), values_in, values_out, msd, status, xcc_map, spin_budget, key_base, sub_bits, drift)
    // (the hybrid form runs only with the LDS-atomic ranking; bare keys may take their places by reservation)
    if (key_bytes == 8 && values_in != nullptr) {  // (payloads: stable, by look-back)
        if (atomic_rank) VRS_PASS_B(uint64_t, 8, RANK_ATOMIC, true, false); else VRS_PASS_B(uint64_t, 8, RANK_BALLOT, true, false);
    } else if (key_bytes == 8) {
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
                                     uint32_t *clear_status, size_t clear_words, uint32_t *values) {
    if (max_bucket > (values ? kLocalCapPairsU64 : kLocalCapBig)) return hipErrorInvalidValue;  // the plan would have refused
    const StatusClear sc{reinterpret_cast<uint4 *>(clear_status), static_cast<uint32_t>(clear_words / 4)};
    if (values != nullptr) {
        if (max_bucket > kLocalCapPairsU64Small)
            VRS_LAUNCH(msd_local_sort_pairs_u64_kernel<kLocalPairsU64Big>, dim3(kMsdBuckets), dim3(kLocalPairsU64Threads), stream, ev, static_cast<uint64_t *>(keys), values, msd, sc, &msd->cursor_a[0][0]);
        else
            VRS_LAUNCH(msd_local_sort_pairs_u64_kernel<kLocalPairsU64Small>, dim3(kMsdBuckets), dim3(kLocalPairsU64Threads), stream, ev, static_cast<uint64_t *>(keys), values, msd, sc, &msd->cursor_a[0][0]);
        return hipGetLastError();
    }
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
uint32_t msd_local_capacity_pairs_u64(bool small) { return small ? kLocalCapPairsU64Small : kLocalCapPairsU64; }

}  // namespace vrs
