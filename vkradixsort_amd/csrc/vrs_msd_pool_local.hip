// vrs_msd_pool_local.hip -- the pool form's last kernel (vrs_msd_pool.hip has the form as a whole): one workgroup -- or one WAVE -- per bucket
// reads the bucket from its region of the slack buffer (ONE contiguous, 16-byte aligned piece), sorts it inside LDS (lean_sort_body /
// wave_sort_body / local_pass, vrs_local_sort.hpp) and streams it to its final place in the caller's buffer; every workgroup derives the
// second verdict from the same two words, workgroup 0 tells the host.  The reference has no counterpart (its four passes are global).
#include "vrs_local_sort.hpp"

#include <cstdlib>


namespace vrs {

namespace {

// ---------------------------------------------------------------------------------------------
// Local sort: workgroup w = bucket 16383 - w.  The bucket lies in ONE piece at the start of its slack region (a 16-byte boundary):
// read like lean_sort_bucket (vrs_msd_hybrid.hip) reads a bucket of the counted form, sorted by lean_sort_body, written -- unlike
// there -- somewhere else: to the bucket's final place in the caller's buffer, whose misalignment is the OUTPUT's alone.
template <int THREADS, int VEC>
__device__ __forceinline__ void slack_load(uint32_t (&k)[4 * VEC], const uint32_t *src, uint32_t n) {
    const uint32_t nvec = (n + 3u) / 4u;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        uint32_t v = j * THREADS + threadIdx.x;
        if (j == VEC - 1) v = v < nvec ? v : nvec - 1u;  // only the last row can reach behind the bucket
        const uint4 t = reinterpret_cast<const uint4 *>(src)[v];
        k[4 * j] = t.x;
        k[4 * j + 1] = t.y;
        k[4 * j + 2] = t.z;
        k[4 * j + 3] = t.w;
    }
}

template <int THREADS, int VEC>
__device__ __attribute__((noinline)) void slack_sort_guarded(const uint32_t *src, uint32_t *abase, uint32_t mis, uint32_t n, uint32_t *s_keys,
                                                            uint32_t *s_hist2, uint32_t *s_tmp, uint32_t guards) {
    // (out of line, loading the bucket again: this copy's registers must not cost the common path its occupancy -- lean_sort_bucket)
    uint32_t k[4 * VEC];
    slack_load<THREADS, VEC>(k, src, n);
    lean_sort_body<THREADS, VEC, true, true, true>(k, abase, mis, n, s_keys, s_hist2, s_tmp, (guards & 1u) != 0u, (guards & 2u) != 0u, 0u);
}

template <int THREADS, int VEC>
__device__ __forceinline__ void slack_sort_bucket(const uint32_t *src, uint32_t *abase, uint32_t mis, uint32_t n, uint32_t *s_keys, uint32_t *s_hist2,
                                                  uint32_t *s_tmp) {
    constexpr int WAVES = THREADS / 64;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t k[4 * VEC];
    slack_load<THREADS, VEC>(k, src, n);
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
    __syncthreads();
    uint32_t guards = 0;
#pragma unroll
    for (int v = 0; v < WAVES; ++v) guards |= s_tmp[16 + v];
    guards = __builtin_amdgcn_readfirstlane(guards);
    if (guards == 0u) lean_sort_body<THREADS, VEC, false, true, true>(k, abase, mis, n, s_keys, s_hist2, s_tmp, false, false, 0u);
    else slack_sort_guarded<THREADS, VEC>(src, abase, mis, n, s_keys, s_hist2, s_tmp, guards);
}

// What every workgroup of a local sort does first, whatever its shape: workgroup w = bucket (buckets - 1 - w) -- the LAST bucket
// first: the second pass wrote the top bytes in ascending order, the highest are what the memory-side cache still holds (round 4:
// 215 -> 208 us; first bucket first measured 186-205 instead of 177-181 here).  False: nothing to sort (the verdict said no, the
// bucket is empty).
template <int THREADS, uint32_t CAPACITY, uint32_t SUBBITS>
__device__ __forceinline__ bool pool_bucket(const uint32_t *__restrict__ slack, uint32_t *__restrict__ keys_out, MsdPlan *__restrict__ msd,
                                            const PoolPlan *__restrict__ pool, uint32_t *__restrict__ cursors, OnesweepPlanHead *__restrict__ dev_head,
                                            OnesweepPlanHead *host_head, uint32_t stamp, uint32_t *host_log, uint32_t retry, uint32_t par,
                                            const uint32_t *&src, uint32_t *&abase, uint32_t &mis, uint32_t &n, const StatusClear sc = {nullptr, 0u}) {
    constexpr uint32_t SUB = 1u << SUBBITS, PER = SUB / 64u;
    const uint32_t b = gridDim.x - 1u - blockIdx.x, a = b >> SUBBITS, c = b & (SUB - 1u);  // (the grid: the top bytes that exist x SUB)
    const uint32_t lane = threadIdx.x & 63u;
    // The bucket's region, its top byte's start and the counters of the top byte's buckets (PER per lane, every wave the same
    // 256 or 512 bytes) are asked for BEFORE the verdict is looked at (all exist whatever it says): a workgroup lives for a few memory latencies.
    uint32_t cnt[PER];
#pragma unroll
    for (uint32_t q = 0; q < PER; ++q) cnt[q] = pool->sub_cursor[(a << SUBBITS) + 64u * q + lane];
    const uint32_t start = pool->sub_start[b], top = pool->top_base[a];
    // Verdict 2, by every workgroup from the same two words (final when this kernel starts): verdict 1 said yes and no pass flagged
    // the sort (a region out of room, a bucket above this kernel's capacity, a key outside the probed range).  Workgroup 0 tells the host.
    // A bucket beyond THIS kernel's shape (fail bit 1; the shape was chosen from n alone) is no refusal of the form: the bucket lies
    // whole in its region, this kernel leaves, and the host -- told the bucket's size -- enqueues a larger shape (retry: that second one).
    const uint32_t flags = pool->fail[par], mx = pool->max_bucket;
    const uint32_t ok = (pool->ok_a != 0u && (flags & (retry ? 1u : 3u)) == 0u && (retry == 0u || mx <= CAPACITY - 3u)) ? 1u : 0u;
    const uint32_t again = (ok == 0u && retry == 0u && pool->ok_a != 0u && flags == 2u) ? mx : 0u;  // != 0: a larger local sort finishes the sort
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        msd->ok = ok;
        dev_head->msd_ok = ok;
        dev_head->msd_max_bucket = again;
        dev_head->lsd_missing = 1u;
        if (host_head) {
            __hip_atomic_store(&host_head->lsd_missing, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&host_head->msd_ok, ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&host_head->msd_max_bucket, again, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            // (a finish among several enqueued before any is asked about: its decision also goes to the log, vrs_msd_finish_status_at)
            if (host_log) __hip_atomic_store(&host_log[stamp & (kMsdLogWords - 1u)], (stamp << 1) | ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
            __hip_atomic_store(&host_head->ready, stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (ok == 0u) return false;  // (enqueued before the verdicts were known, and one said no)
    clear_status_share(sc, THREADS);  // (pairs: the look-back words of the two passes, clear for the next sort -- every workgroup its share)
    // the first pass's cursors, zero for the next sort (the counted form's local sort does the same: rearm_reservation)
    if (blockIdx.x < 2u * kStreams)
        for (uint32_t q = threadIdx.x; q < 256u; q += THREADS) cursors[blockIdx.x * 256u + q] = 0;
    // keys of the top byte's buckets before this one (every wave sums the counters below c), and this bucket's own
    uint32_t before = 0;
    n = 0;
#pragma unroll
    for (uint32_t q = 0; q < PER; ++q) {
        before += 64u * q + lane < c ? cnt[q] : 0u;
        const uint32_t v = __builtin_amdgcn_readlane(cnt[q], c & 63u);
        n = (c >> 6) == q ? v : n;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) before += __shfl_xor(before, o);
    const uint32_t begin = top + __builtin_amdgcn_readfirstlane(before);
    mis = static_cast<uint32_t>((reinterpret_cast<uintptr_t>(keys_out + begin) >> 2) & 3u);
    if (n == 0 || mis + n > CAPACITY) return false;  // uniform; above the capacity cannot happen (the second pass would have flagged it)
    abase = keys_out + begin - mis;
    src = slack + start;
    return true;
}

// Shapes: THREADS x 4 MAXVEC slots -- 256 x 16 (buckets up to 4093 keys: 28 KB of LDS, five workgroups per CU), 256 x 28 (7165 keys,
// four per CU), 512 x 28 (14333 keys, two per CU); and ONE WAVE per bucket (below) for buckets up to 1789 keys.  A workgroup lives for
// two memory round trips (its bucket's words, its keys) on top of the sort itself -- about 5 us of an 11 us life at 6100 keys.
template <int THREADS, int MAXVEC, int WGS, uint32_t SUBBITS>
__global__ __launch_bounds__(THREADS, WGS *(THREADS / 64) / 4) void pool_local_sort_kernel(const uint32_t *__restrict__ slack, uint32_t *__restrict__ keys_out,
                                                                                          MsdPlan *__restrict__ msd, const PoolPlan *__restrict__ pool,
                                                                                          uint32_t *__restrict__ cursors, OnesweepPlanHead *__restrict__ dev_head,
                                                                                          OnesweepPlanHead *host_head, uint32_t stamp, uint32_t *host_log, uint32_t retry, uint32_t par) {
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[THREADS * 4 * MAXVEC + 4];
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[(THREADS / 64 + 1) * kLeanRow];
    __shared__ uint32_t s_tmp[32];
    const uint32_t *src;
    uint32_t *abase, mis, n;
    if (!pool_bucket<THREADS, THREADS * 4u * MAXVEC, SUBBITS>(slack, keys_out, msd, pool, cursors, dev_head, host_head, stamp, host_log, retry, par, src, abase, mis, n)) return;
    const uint32_t rows = (mis + n + 4u * THREADS - 1u) / (4u * THREADS);  // rows of THREADS vectors the bucket touches where it is written
    if constexpr (MAXVEC == 4) {
        switch (rows) {
            case 1: slack_sort_bucket<THREADS, 1>(src, abase, mis, n, s_keys, s_hist, s_tmp); break;
            case 2: slack_sort_bucket<THREADS, 2>(src, abase, mis, n, s_keys, s_hist, s_tmp); break;
            case 3: slack_sort_bucket<THREADS, 3>(src, abase, mis, n, s_keys, s_hist, s_tmp); break;
            default: slack_sort_bucket<THREADS, 4>(src, abase, mis, n, s_keys, s_hist, s_tmp); break;
        }
    } else {
        switch (rows) {
            case 1: slack_sort_bucket<THREADS, 1>(src, abase, mis, n, s_keys, s_hist, s_tmp); break;
            case 2: slack_sort_bucket<THREADS, 2>(src, abase, mis, n, s_keys, s_hist, s_tmp); break;
            case 3: slack_sort_bucket<THREADS, 3>(src, abase, mis, n, s_keys, s_hist, s_tmp); break;
            case 4: slack_sort_bucket<THREADS, 4>(src, abase, mis, n, s_keys, s_hist, s_tmp); break;
            case 5: slack_sort_bucket<THREADS, 5>(src, abase, mis, n, s_keys, s_hist, s_tmp); break;
            case 6: slack_sort_bucket<THREADS, 6>(src, abase, mis, n, s_keys, s_hist, s_tmp); break;
            default: slack_sort_bucket<THREADS, 7>(src, abase, mis, n, s_keys, s_hist, s_tmp); break;
        }
    }
}

// Key + payload pairs: the bucket's keys and payloads from the two slack buffers (the same region in both), two STABLE 9-bit passes
// inside LDS (local_pass, vrs_local_sort.hpp: the counted form's local sort of pairs), written to the bucket's final place in the
// caller's two buffers.  512 threads x up to 13 pairs (two workgroups per CU), or 1024 x 13 for buckets of up to 13312.
template <int THREADS, uint32_t SUBBITS, bool PACKED>
__global__ __launch_bounds__(THREADS, (PACKED && THREADS == 512) ? 6 : 4) void pool_local_sort_pairs_kernel(const uint32_t *__restrict__ slack, uint32_t *__restrict__ keys_out,
                                                                          MsdPlan *__restrict__ msd, const PoolPlan *__restrict__ pool,
                                                                          uint32_t *__restrict__ cursors, OnesweepPlanHead *__restrict__ dev_head,
                                                                          OnesweepPlanHead *host_head, uint32_t stamp, uint32_t *host_log, uint32_t retry,
                                                                          uint32_t par, PoolPayloads pv) {
    constexpr int WAVES = THREADS / 64;
    constexpr uint32_t CAP = THREADS * kPoolPairItems;
    __shared__ uint32_t s_keys[CAP];
    __shared__ uint32_t s_vals[PACKED ? 1 : CAP];  // (packed: the payloads take the words' array once the words are sorted)
    __shared__ uint32_t s_hist[WAVES << 9];
    __shared__ uint32_t s_tmp[1 + WAVES];
    const uint32_t *src;
    uint32_t *abase, mis, n;
    const StatusClear sc{reinterpret_cast<uint4 *>(pv.status), static_cast<uint32_t>(pv.status_words / 4u)};
    if (!pool_bucket<THREADS, CAP + 3u, SUBBITS>(slack, keys_out, msd, pool, cursors, dev_head, host_head, stamp, host_log, retry, par, src, abase, mis, n, sc)) return;
    if (n > CAP) return;  // (cannot happen: the second pass flags a bucket above the capacity it was told)
    uint32_t *bucket = abase + mis, *bvals = pv.values_home + (bucket - keys_out);
    const uint32_t *svals = pv.slack_values + (src - slack);
    const uint32_t used = (n + THREADS - 1u) / THREADS;
    // the bits a bucket's keys differ in: below the first pass's digit (the top byte's shift + what the cut left of the byte) and the second pass's bits
    const uint32_t lowbits = pool->shift + kMsdBits - pv.top_bits - SUBBITS;
    constexpr int IDXB = THREADS == 512 ? 13 : 14;
    if constexpr (PACKED) {
        if (lowbits > 18u) return;  // (cannot happen: the top byte's shift is at most 24, and the two passes took 14 bits or more of the range)
        if (used <= 4) local_sort_packed_pairs_to<THREADS, 4, IDXB, true>(src, svals, bucket, bvals, n, lowbits, s_keys, s_hist, s_tmp);
        else if (used <= 8) local_sort_packed_pairs_to<THREADS, 8, IDXB, true>(src, svals, bucket, bvals, n, lowbits, s_keys, s_hist, s_tmp);
        else if (used <= 10) local_sort_packed_pairs_to<THREADS, 10, IDXB, true>(src, svals, bucket, bvals, n, lowbits, s_keys, s_hist, s_tmp);
        else if (used <= 12) local_sort_packed_pairs_to<THREADS, 12, IDXB, true>(src, svals, bucket, bvals, n, lowbits, s_keys, s_hist, s_tmp);
        else local_sort_packed_pairs_to<THREADS, kPoolPairItems, IDXB, true>(src, svals, bucket, bvals, n, lowbits, s_keys, s_hist, s_tmp);
        return;
    }
    if (used <= 2) local_sort_bucket_to<THREADS, 2, true, true>(src, svals, bucket, bvals, n, s_keys, s_vals, s_hist, s_tmp);
    else if (used <= 4) local_sort_bucket_to<THREADS, 4, true, true>(src, svals, bucket, bvals, n, s_keys, s_vals, s_hist, s_tmp);
    else if (used <= 6) local_sort_bucket_to<THREADS, 6, true, true>(src, svals, bucket, bvals, n, s_keys, s_vals, s_hist, s_tmp);
    else if (used <= 8) local_sort_bucket_to<THREADS, 8, true, true>(src, svals, bucket, bvals, n, s_keys, s_vals, s_hist, s_tmp);
    else if (used <= 10) local_sort_bucket_to<THREADS, 10, true, true>(src, svals, bucket, bvals, n, s_keys, s_vals, s_hist, s_tmp);
    else if (used <= 12) local_sort_bucket_to<THREADS, 12, true, true>(src, svals, bucket, bvals, n, s_keys, s_vals, s_hist, s_tmp);
    else local_sort_bucket_to<THREADS, kPoolPairItems, true, true>(src, svals, bucket, bvals, n, s_keys, s_vals, s_hist, s_tmp);
}

// Small buckets (up to 1789 keys: uniform inputs below about 2.6e7 keys): ONE WAVE per bucket, no workgroup barrier anywhere
// (msd_local_sort_wave_kernel's idea, vrs_msd_hybrid.hip: 16 independent buckets per CU instead of workgroups whose fixed work is most of
// their life) -- with it the pool form is worth taking from about 10^7 keys on.
template <int VEC>
__device__ __forceinline__ void slack_wave_load(uint32_t (&k)[4 * VEC], const uint32_t *src, uint32_t n) {
    const uint32_t lane = threadIdx.x & 63u, nvec = (n + 3u) / 4u;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        uint32_t v = j * 64 + lane;
        if (j == VEC - 1) v = v < nvec ? v : nvec - 1u;
        const uint4 t = reinterpret_cast<const uint4 *>(src)[v];
        k[4 * j] = t.x;
        k[4 * j + 1] = t.y;
        k[4 * j + 2] = t.z;
        k[4 * j + 3] = t.w;
    }
}
template <int VEC>
__device__ __attribute__((noinline)) void slack_wave_sort_guarded(const uint32_t *src, uint32_t *abase, uint32_t mis, uint32_t n, uint32_t *s_keys, uint32_t *tbl,
                                                                 uint32_t skew) {
    uint32_t k[4 * VEC];
    slack_wave_load<VEC>(k, src, n);
    wave_sort_body<VEC, true, true, true>(k, abase, mis, n, s_keys, tbl, (skew & 1u) != 0u, (skew & 2u) != 0u, 0u);
}
template <int VEC>
__device__ __forceinline__ void slack_wave_sort_bucket(const uint32_t *src, uint32_t *abase, uint32_t mis, uint32_t n, uint32_t *s_keys, uint32_t *tbl) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t k[4 * VEC];
    slack_wave_load<VEC>(k, src, n);
    // the table zeroed: 576 words, two 16-byte stores per lane + one more from the first 16 lanes
    reinterpret_cast<uint4 *>(tbl)[2 * lane] = make_uint4(0, 0, 0, 0);
    reinterpret_cast<uint4 *>(tbl)[2 * lane + 1] = make_uint4(0, 0, 0, 0);
    if (lane < 16u) reinterpret_cast<uint4 *>(tbl)[128 + lane] = make_uint4(0, 0, 0, 0);
    uint32_t skew = 0;  // does an instruction of the first row put half its lanes on one counter?  bit 0: pass 1, bit 1: pass 2
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t a1 = k[c] & 511u, a2 = (k[c] >> 9) & 511u;
        skew |= __popcll(__ballot(a1 == __builtin_amdgcn_readfirstlane(a1))) >= 32 ? 1u : 0u;
        skew |= __popcll(__ballot(a2 == __builtin_amdgcn_readfirstlane(a2))) >= 32 ? 2u : 0u;
    }
    skew = __builtin_amdgcn_readfirstlane(skew);
    wave_phase();
    if (skew == 0u) wave_sort_body<VEC, false, true, true>(k, abase, mis, n, s_keys, tbl, false, false, 0u);
    else slack_wave_sort_guarded<VEC>(src, abase, mis, n, s_keys, tbl, skew);
}
template <uint32_t SUBBITS, int MAXVEC>  // MAXVEC 7: buckets up to 1789 keys (9.5 KB of LDS, 16 buckets per CU at a time); four rows (1021 keys, 25 per CU) measured the same
__global__ __launch_bounds__(64, 4) void pool_local_sort_wave_kernel(const uint32_t *__restrict__ slack, uint32_t *__restrict__ keys_out, MsdPlan *__restrict__ msd,
                                                                   const PoolPlan *__restrict__ pool, uint32_t *__restrict__ cursors,
                                                                   OnesweepPlanHead *__restrict__ dev_head, OnesweepPlanHead *host_head, uint32_t stamp,
                                                                   uint32_t *host_log, uint32_t retry, uint32_t par) {
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[64 * 4 * MAXVEC + 4];
    __shared__ __attribute__((aligned(16))) uint32_t s_tbl[kLeanRow];
    const uint32_t *src;
    uint32_t *abase, mis, n;
    if (!pool_bucket<64, 64u * 4u * MAXVEC, SUBBITS>(slack, keys_out, msd, pool, cursors, dev_head, host_head, stamp, host_log, retry, par, src, abase, mis, n)) return;
    const uint32_t rows = (mis + n + 255u) / 256u;
    if constexpr (MAXVEC == 4) {
        switch (rows) {
            case 1: slack_wave_sort_bucket<1>(src, abase, mis, n, s_keys, s_tbl); break;
            case 2: slack_wave_sort_bucket<2>(src, abase, mis, n, s_keys, s_tbl); break;
            case 3: slack_wave_sort_bucket<3>(src, abase, mis, n, s_keys, s_tbl); break;
            default: slack_wave_sort_bucket<4>(src, abase, mis, n, s_keys, s_tbl); break;
        }
    } else {
        switch (rows) {
            case 1: slack_wave_sort_bucket<1>(src, abase, mis, n, s_keys, s_tbl); break;
            case 2: slack_wave_sort_bucket<2>(src, abase, mis, n, s_keys, s_tbl); break;
            case 3: slack_wave_sort_bucket<3>(src, abase, mis, n, s_keys, s_tbl); break;
            case 4: slack_wave_sort_bucket<4>(src, abase, mis, n, s_keys, s_tbl); break;
            case 5: slack_wave_sort_bucket<5>(src, abase, mis, n, s_keys, s_tbl); break;
            case 6: slack_wave_sort_bucket<6>(src, abase, mis, n, s_keys, s_tbl); break;
            default: slack_wave_sort_bucket<7>(src, abase, mis, n, s_keys, s_tbl); break;
        }
    }
}

}  // namespace

hipError_t launch_pool_local_sort(hipStream_t stream, const uint32_t *slack, uint32_t *keys_out, uint32_t n, MsdPlan *msd, const PoolPlan *pool,
                                  PoolShape shape, OnesweepPlanHead *dev_head, OnesweepPlanHead *host_head, uint32_t stamp, uint32_t par,
                                  LaunchEvents ev, uint32_t top_bytes, uint32_t *host_log, bool retry, const PoolPayloads *pv) {
    const uint32_t again = retry ? 1u : 0u;
    uint32_t *cursors = &msd->cursor_a[0][0];
    if (top_bytes == 0u || top_bytes > 256u || (top_bytes << shape.sub_bits) > kPoolMaxBuckets) return hipErrorInvalidValue;
    if (pv || shape.local >= 4u) {  // pairs
        if (!pv || (shape.local != 4u && shape.local != 5u)) return hipErrorInvalidValue;
        const uint32_t buckets = top_bytes << shape.sub_bits;
        // the packed form (one word per pair in LDS, three workgroups per CU) for buckets of up to ten rows of the 512-thread workgroup on average:
        // 5-9 % faster from 2.6e7 to 8e7 pairs, level or 1 % behind at 12-13 rows (10^8, 2e8 pairs) -- profiles/labs/r06_pairs_packed.txt
        const bool packed = pv->packed == 1 || (pv->packed < 0 && shape.local == 4u && n / buckets <= kPoolPackedPairsMean);
#define VRS_POOL_PAIRS(T, S)                                                                                                                          \
    do {                                                                                                                                              \
        if (packed)                                                                                                                                   \
            VRS_LAUNCH((pool_local_sort_pairs_kernel<T, S, true>), dim3(buckets), dim3(T), stream, ev, slack, keys_out, msd, pool, cursors, dev_head, \
                       host_head, stamp, host_log, again, par, *pv);                                                                                  \
        else                                                                                                                                          \
            VRS_LAUNCH((pool_local_sort_pairs_kernel<T, S, false>), dim3(buckets), dim3(T), stream, ev, slack, keys_out, msd, pool, cursors, dev_head, \
                       host_head, stamp, host_log, again, par, *pv);                                                                                  \
    } while (0)
        if (shape.sub_bits == 8u) {
            if (shape.local == 4u) VRS_POOL_PAIRS(512, 8);
            else VRS_POOL_PAIRS(1024, 8);
        } else if (shape.sub_bits == 7u) {
            if (shape.local == 4u) VRS_POOL_PAIRS(512, 7);
            else VRS_POOL_PAIRS(1024, 7);
        } else {
            if (shape.local == 4u) VRS_POOL_PAIRS(512, 6);
            else VRS_POOL_PAIRS(1024, 6);
        }
#undef VRS_POOL_PAIRS
        return hipGetLastError();
    }
#define VRS_POOL_LOCAL(T, V, W, S)                                                                                                            \
    VRS_LAUNCH((pool_local_sort_kernel<T, V, W, S>), dim3(top_bytes << S), dim3(T), stream, ev, slack, keys_out, msd, pool, cursors, dev_head, host_head, \
               stamp, host_log, again, par)
#define VRS_POOL_LOCAL_S(S)                                    \
    do {                                                       \
        if (shape.local == 3u)                                 \
            VRS_LAUNCH((pool_local_sort_wave_kernel<S, 7>), dim3(top_bytes << S), dim3(64), stream, ev, slack, keys_out, msd, pool, cursors, dev_head, host_head, \
                       stamp, host_log, again, par);           \
        else if (shape.local == 0u) VRS_POOL_LOCAL(256, 4, 5, S);   \
        else if (shape.local == 1u) VRS_POOL_LOCAL(256, 7, 4, S); \
        else VRS_POOL_LOCAL(512, 7, 2, S);                     \
    } while (0)
    if (shape.sub_bits == 8u) VRS_POOL_LOCAL_S(8);
    else if (shape.sub_bits == 7u) VRS_POOL_LOCAL_S(7);
    else VRS_POOL_LOCAL_S(6);
#undef VRS_POOL_LOCAL_S
#undef VRS_POOL_LOCAL
    return hipGetLastError();
}

}  // namespace vrs
