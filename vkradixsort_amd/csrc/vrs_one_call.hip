// vrs_one_call.hip -- K5, the one-call sort's own kernels (vrs_sort_keys_u32 / _u64 / vrs_sort_pairs_u32 from 2^21 keys on): ONE counting
// read of all digits (digit_tables_kernel), the plan (plan_kernel), and the stable scatter pass with decoupled look-back or -- first
// MSD pass over bare keys -- with reserved places (onesweep_scatter_kernel).  The reference counts once per pass
// (multi_radixsort_histograms.comp:42-50) and prefixes inside the scatter shader (multi_radixsort.comp:56-77).
#include "vrs_device.hpp"
#include "vrs_plan.hpp"

#include <algorithm>
#include <type_traits>

namespace vrs {

// ---------------------------------------------------------------------------------------------
// K5: the one-call sort (vrs_sort_keys_u32 / _u64 / vrs_sort_pairs_u32) for large N: 36 instead of 48 bytes per key.
//
// The contract path reads the keys once per pass just to count them, because the [W][256] table is part of the
// reference's interface.  When the library owns all four passes it can count ONCE, before the first pass, and let
// every scatter pass find its offsets by decoupled look-back.  A single look-back chain over all tiles does not
// fit this chip (tiles must stay in XCD-contiguous order for the L2s to merge their partial lines, and 500-750
// resident tiles finish 12 ns apart while a hand-off between workgroups takes 1-3 us under load), so the tiles of a
// pass are cut into kStreams independent STREAMS -- one per XCD -- whose starting offsets are known before the pass
// starts.  A stream is a run of neighbouring GROUPS; a key's group is a function of the key alone:
//   pass 0   group g = the g-th slice of the input (whole tiles of 8192 uint32 / 4096 uint64 keys);
//   pass p>0 group g = the keys whose digit p-1 lies in [g * 256/G, (g+1) * 256/G): after pass p-1 they are the
//            contiguous range [P_{p-1}[g * 256/G], P_{p-1}[(g+1) * 256/G]) of its output (P = exclusive digit prefix),
//            whatever their order inside.
// digit_tables_kernel counts, in one read of the keys, H[p][g][d] = #keys of group g of pass p with digit p == d;
// plan_kernel merges the G groups of each pass into kStreams streams of nearly equal length and turns H into their
// ranges and seeds (P_p[d] + the keys with digit d in the groups before the stream's first group);
// onesweep_scatter_kernel walks stream s in tile order on XCD s % 8 (tiles of one stream are neighbours in that L2)
// and looks back only along its own stream.
// Streams follow the data: a pass whose streams cannot be balanced (one group holds far more than 1/kStreams of the
// keys: keys that are all multiples of 256, say) is marked in the plan and run through the contract path instead.

// ---- K5b, hybrid form of the one-call sort for uint32 keys (28 instead of 36 bytes per key): an MSD partition by the top
// kMsdBits bits in two look-back scatter passes (8 + 6 bits), then every bucket (about N / 16384 keys) is sorted by its
// low 18 bits inside ONE workgroup's LDS and written back once.
// (kMsdBits, kMsdSub, ... : vrs_device.hpp)

// fused form of the counting read: the last workgroup to finish also makes the plan (plan == nullptr: separate kernel)
struct FusedPlanArgs {
    OnesweepPlan *plan;
    OnesweepPlanHead *host_head;
    uint32_t *done;  // ticket counter, zero between launches
    uint32_t stamp, tile, tile_cap, blind_cap;
    StreamCuts cuts0;
};

// LDS row of one group's 256 counters, padded by one word: keys that share the counted digit but not the group
// (sorted input) would otherwise hit one LDS bank from every lane
constexpr int kTableRow = kBins + 1;
// The pass-0 table has only 256 counters, hit by every key of the workgroup: it is kept in COPIES copies, lane l
// adding to copy l % COPIES (word d0 * COPIES + copy, so the lanes of a half wave spread over COPIES banks whatever
// their digits; 32 copies = conflict-free).
template <int GROUPS, int COPIES>
struct TableIndex {
    static constexpr int kShift = GROUPS == 32 ? 3 : GROUPS == 16 ? 4 : 5;  // log2(256 / GROUPS)
    // word of pass-0 digit / of the joint (group of digit p-1, digit p) counter
    static __device__ __forceinline__ uint32_t t0(uint32_t w, uint32_t lane) { return (w & 255u) * COPIES + (lane % COPIES); }
    static __device__ __forceinline__ uint32_t t1(uint32_t w) { return ((w & 255u) >> kShift) * kTableRow + ((w >> 8) & 255u); }
    static __device__ __forceinline__ uint32_t t2(uint32_t w) { return (((w >> 8) & 255u) >> kShift) * kTableRow + ((w >> 16) & 255u); }
    static __device__ __forceinline__ uint32_t t3(uint32_t w) { return (((w >> 16) & 255u) >> kShift) * kTableRow + (w >> 24); }
};

// V counters per lane (one 16-byte vector of keys).  Same-address lanes of one LDS atomic are served one after the
// other, so input with few distinct counters per wave (constant bytes, sorted or clustered keys) would crawl.  When
// (nearly) every lane's keys share a counter -- the signature of such input -- the wave adds once per RUN of equal
// counters across its lanes instead of once per key; uniform-random keys fail the vote at once and take the plain path.
// clustered: the vote, taken by the caller on the first vector of a step (it is a speed heuristic only: both forms
// count every key exactly once).
template <int V>
__device__ __forceinline__ bool table_vote(const uint32_t (&idx)[V]) {
    bool same = true;
#pragma unroll
    for (int j = 1; j < V; ++j) same = same && idx[j] == idx[0];
    return __popcll(__ballot(same)) >= 48;  // wave-uniform
}
template <int V>
__device__ __forceinline__ void table_add(uint32_t *t, const uint32_t (&idx)[V], uint32_t lane, bool clustered) {
    if (clustered) {  // wave-uniform
        // run-length aggregation across the lanes: the first lane of every run of equal counters adds the whole run.
        // (Equal counters in different runs just add twice: always correct, best on sorted / clustered input.)
        bool same = true;
#pragma unroll
        for (int j = 1; j < V; ++j) same = same && idx[j] == idx[0];
        const uint64_t uniform_lanes = __ballot(same);
        const uint32_t mine = same ? idx[0] : 0xFFFFFFFFu;  // lanes that straddle two counters break the runs
        const uint32_t prev = __shfl_up(mine, 1);
        const bool head = same && (lane == 0u || prev != mine);
        const uint64_t breaks = __ballot(head) | ~uniform_lanes;
        const uint64_t after = lane == 63u ? 0ull : breaks >> (lane + 1u);
        const uint32_t run = after ? static_cast<uint32_t>(__ffsll(static_cast<long long>(after))) : 64u - lane;
        if (head) atomicAdd(&t[idx[0]], static_cast<uint32_t>(V) * run);
        if (!same) {
#pragma unroll
            for (int j = 0; j < V; ++j) atomicAdd(&t[idx[j]], 1u);
        }
    } else {
#pragma unroll
        for (int j = 0; j < V; ++j) atomicAdd(&t[idx[j]], 1u);
    }
}

// the 32-bit word of a key that holds the four digits of this group of passes (bits [base_shift, base_shift + 32))
__device__ __forceinline__ uint32_t digit_word(uint32_t key, uint32_t) { return key; }
__device__ __forceinline__ uint32_t digit_word(uint64_t key, uint32_t base_shift) {
    return static_cast<uint32_t>(key >> base_shift);
}

// hm: the 16384-bin histogram of the key's top 14 bits (hybrid form, K5b), or nullptr; t0: nullptr = no LSD tables
template <typename TI>
__device__ __forceinline__ void digit_tables_count(uint32_t *t0, uint32_t *t1, uint32_t *t2, uint32_t *t3, uint32_t *hm,
                                                   uint32_t msd_shift, uint32_t msd_base, uint32_t &msd_over, uint32_t w) {
    if (t0) {  // workgroup-uniform: nullptr when only the bucket histogram is counted (hybrid form, fast count)
        atomicAdd(&t0[TI::t0(w, lane_id())], 1u);
        atomicAdd(&t1[TI::t1(w)], 1u);
        atomicAdd(&t2[TI::t2(w)], 1u);
        atomicAdd(&t3[TI::t3(w)], 1u);
    }
    if (hm) {
        const uint32_t b = (w - msd_base) >> msd_shift;
        msd_over |= (b >> kMsdBits) | (w < msd_base ? 1u : 0u);  // a key above the probed range (or below the promised floor): the plan will refuse the hybrid form
        atomicAdd(&hm[min(b, kMsdBuckets - 1u)], 1u);
    }
}

// one 16-byte vector of keys per lane: 4 uint32 or 2 uint64.  vote: bit t = table t takes the run-length form
template <typename K, typename TI, bool VOTE, bool MSD>
__device__ __forceinline__ void digit_tables_count_vec(uint32_t *t0, uint32_t *t1, uint32_t *t2, uint32_t *t3, uint32_t *hm,
                                                       uint32_t msd_shift, uint32_t msd_base, uint32_t &msd_over,
                                                       const typename KeyVec<K>::type &q, uint32_t base_shift,
                                                       uint32_t lane, uint32_t &vote) {
    constexpr int V = KeyVec<K>::kKeys;
    if constexpr (MSD) {
        if (t0 == nullptr) {  // only the bucket histogram (workgroup-uniform; hm != nullptr then)
            uint32_t im[V];
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const uint32_t w = digit_word(KeyVec<K>::get(q, j), base_shift);
                const uint32_t b = (w - msd_base) >> msd_shift;
                msd_over |= (b >> kMsdBits) | (w < msd_base ? 1u : 0u);
                im[j] = min(b, kMsdBuckets - 1u);
            }
            if constexpr (VOTE) vote = table_vote<V>(im) ? 16u : 0u;
            table_add<V>(hm, im, lane, (vote & 16u) != 0u);
            return;
        }
    }
    uint32_t i0[V], i1[V], i2[V], i3[V], im[MSD ? V : 1];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const uint32_t w = digit_word(KeyVec<K>::get(q, j), base_shift);
        i0[j] = TI::t0(w, lane);
        i1[j] = TI::t1(w);
        i2[j] = TI::t2(w);
        i3[j] = TI::t3(w);
        if constexpr (MSD) {
            const uint32_t b = (w - msd_base) >> msd_shift;
            msd_over |= (b >> kMsdBits) | (w < msd_base ? 1u : 0u);
            im[j] = min(b, kMsdBuckets - 1u);
        }
    }
    if constexpr (VOTE) {
        vote = (table_vote<V>(i0) ? 1u : 0u) | (table_vote<V>(i1) ? 2u : 0u) | (table_vote<V>(i2) ? 4u : 0u) |
               (table_vote<V>(i3) ? 8u : 0u);
        if constexpr (MSD) {
            if (hm) vote |= table_vote<V>(im) ? 16u : 0u;
        }
    }
    table_add<V>(t0, i0, lane, (vote & 1u) != 0u);
    table_add<V>(t1, i1, lane, (vote & 2u) != 0u);
    table_add<V>(t2, i2, lane, (vote & 4u) != 0u);
    table_add<V>(t3, i3, lane, (vote & 8u) != 0u);
    if constexpr (MSD) {
        if (hm) table_add<V>(hm, im, lane, (vote & 16u) != 0u);  // workgroup-uniform: nullptr when the key range is too narrow
    }
}

template <int GROUPS>
__global__ __launch_bounds__(4 * kBins) void plan_kernel(uint32_t *__restrict__ tables, OnesweepPlan *__restrict__ plan,
                                                        OnesweepPlanHead *host_head, uint32_t stamp, uint32_t n,
                                                        uint32_t group_len, uint32_t tile, uint32_t tile_cap,
                                                        uint32_t blind_cap, StreamCuts cuts0) {
    plan_body<GROUPS>(tables, plan, host_head, stamp, n, group_len, tile, tile_cap, blind_cap, cuts0);
}


// grid = GROUPS * slices workgroups; workgroup (s, g) counts the g-th part of pass-0 group s and zeroes its share
// of the look-back status words.  group_len (the length of a pass-0 group) is a multiple of 4 * slices.
// The loads run one step ahead of the counting, vector by vector (a vector's register is refilled for the next step
// as soon as it has been consumed), so UNROLL 16-byte loads per lane are in flight all the time.  64-bit keys are sorted
// in two groups of four passes, each with its own counting read: base_shift = 0, then 32.
// MSD (hybrid form, K5b; uint32 keys, GROUPS == 8): the same read also fills a 16384-bin histogram of the top 14 bits
// (msd_hist) and, per pass-0 group, the 256 top-byte counts the MSD pass needs as its streams' seeds (msd_slices).
template <typename K, int GROUPS, int THREADS, int COPIES, int UNROLL, int OCC, bool MSD = false>
__global__ __launch_bounds__(THREADS, OCC) void digit_tables_kernel(const K *__restrict__ keys, uint32_t n,
                                                                    uint32_t base_shift, uint32_t group_len,
                                                                    uint32_t slices, uint32_t *__restrict__ tables,
                                                                    uint4 *__restrict__ status, uint32_t status_vecs,
                                                                    FusedPlanArgs fp, uint32_t *__restrict__ msd_hist,
                                                                    uint32_t *__restrict__ msd_slices, uint32_t msd_only,
                                                                    uint32_t msd_base, uint32_t msd_force_shift) {
    using Vec = typename KeyVec<K>::type;
    using TI = TableIndex<GROUPS, COPIES>;
    constexpr uint32_t V = KeyVec<K>::kKeys;
    __shared__ uint32_t t0_[kBins * COPIES];
    __shared__ uint32_t t[3][GROUPS * kTableRow];
    uint32_t *t0 = t0_;
    __shared__ uint32_t s_msd[MSD ? kMsdBuckets : 1];
    uint32_t *hm = MSD ? s_msd : nullptr;
    // the hybrid form's buckets are the top 14 bits of the key RANGE: msd_hist[kMsdProbeWord] holds the shift a probe of
    // the input suggested (range_probe_kernel); a key above that range sets msd_hist[kMsdOverWord]
    uint32_t msd_shift = 0, msd_over = 0;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    if constexpr (MSD) {
        // The hybrid form buckets the keys by the top 14 bits of their RANGE (32-bit keys: bits 18-31; the reference's
        // 28-bit keys: bits 14-27; ...).  Every workgroup ORs the SAME strided sample of 4096 keys (16 KB: served by L2
        // after the first few) and derives the same bucket shift; every key above the sampled range is flagged below, so
        // a wrong guess costs the hybrid form, never the result.
        __shared__ uint32_t s_or;
        if (tid == 0) s_or = msd_force_shift ? 0xFFFFFFFFu >> (18u - min(msd_force_shift, 18u)) : 0u;  // forced: as if keys < 2^(shift + 14) had been seen
        __syncthreads();
        const uint32_t samples = msd_force_shift ? 0u : min(n, 4096u);
        const uint64_t stride = n / max(samples, 1u);  // >= 1
        uint32_t acc = 0;
        for (uint32_t i = tid; i < samples; i += THREADS) acc |= digit_word(keys[static_cast<uint64_t>(i) * stride], base_shift) - msd_base;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc |= __shfl_down(acc, o);
        if (lane == 0u && acc) atomicOr(&s_or, acc);
        __syncthreads();
        const uint32_t bits = s_or ? 32u - static_cast<uint32_t>(__clz(static_cast<int>(s_or))) : 0u;  // sampled keys < 2^bits
        msd_shift = bits > kMsdBits ? bits - kMsdBits : 0u;
        if (blockIdx.x == 0 && tid == 0) msd_hist[kMsdProbeWord] = msd_shift;  // for the plan
        // a key range below 27 bits is left to the LSD passes (the plan will say so): do not pay for the histogram
        if (msd_shift < kMsdMinShift) hm = nullptr;
        // fast count (msd_only): a range the hybrid form takes gets ONLY the bucket histogram -- 1 LDS add per key instead
        // of 5; should the plan then refuse (a bucket too large), the host counts again for the LSD passes
        else if (msd_only) t0 = nullptr;
    }
    if constexpr (MSD) {
        for (uint32_t c = tid; c < kMsdBuckets; c += THREADS) s_msd[c] = 0;
    }
    for (uint32_t c = tid; c < 3u * GROUPS * kTableRow; c += THREADS) (&t[0][0])[c] = 0;
    for (uint32_t c = tid; c < static_cast<uint32_t>(kBins * COPIES); c += THREADS) t0_[c] = 0;
    {
        const uint4 zero = make_uint4(0, 0, 0, 0);
        const uint32_t per = (status_vecs + gridDim.x - 1) / gridDim.x;
        const uint32_t z0 = blockIdx.x * per, z1 = min(z0 + per, status_vecs);
        for (uint32_t c = z0 + tid; c < z1; c += THREADS) status[c] = zero;
    }
    __syncthreads();
    const uint32_t s = blockIdx.x / slices, g = blockIdx.x % slices;
    const uint32_t part = group_len / slices;
    const uint64_t begin64 = static_cast<uint64_t>(s) * group_len + static_cast<uint64_t>(g) * part;
    if (begin64 < n) {
        const uint32_t begin = static_cast<uint32_t>(begin64);
        const uint32_t len = min(part, n - begin);
        // 16-byte loads need a 16-byte aligned address: peel `head` keys (the buffer may start anywhere in a larger
        // allocation; every slice starts a multiple of V keys after it)
        const uint32_t mis = static_cast<uint32_t>((reinterpret_cast<uintptr_t>(keys + begin) / sizeof(K)) % V);
        const uint32_t head = min((V - mis) % V, len);
        if (tid < head) digit_tables_count<TI>(t0, t[0], t[1], t[2], hm, msd_shift, msd_base, msd_over, digit_word(keys[begin + tid], base_shift));
        const Vec *v = reinterpret_cast<const Vec *>(keys + begin + head);
        const uint32_t nvec = (len - head) / V;
        constexpr uint32_t kStep = THREADS * UNROLL;
        uint32_t i0 = 0;
        Vec cur[UNROLL];
        // The loads of a lane return in issue order, and the compiler's wait before vector r is consumed must hold for
        // every way into the loop: the scheduling barriers keep the issue order r = 0, 1, ... in the prologue and in the
        // loop alike, so that wait is "all but the UNROLL - 1 youngest loads" (s_waitcnt vmcnt(UNROLL - 1) before every
        // vector) instead of one "all but one" at the top of the step.  It measures the same (the kernel runs at the
        // HBM rate of its 400 MB read plus the write-back of the previous kernel's dirty lines),
        // but the loads are what the comment above says they are.
        // (nontemporal loads when the keys exceed the caches: the counting read then runs at the rate it has with nothing before it --
        // 98 instead of 133 us at 10^8 keys -- because it no longer has to push the previous kernel's dirty lines out in front of
        // itself; what it saves the next kernel pays in part: 0.603 -> 0.587 ms per sort.  Below about 3e7 keys plain loads are faster.)
        const auto main_loop = [&](auto nt) {
            constexpr bool NT = decltype(nt)::value;
            const auto load = [&](uint32_t at) { return NT ? load_stream16(v + at) : v[at]; };
            if (kStep <= nvec) {
#pragma unroll
                for (int r = 0; r < UNROLL; ++r) {
                    cur[r] = load(r * THREADS + tid);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            for (; i0 + kStep <= nvec; i0 += kStep) {
                // the refill is unconditional (the last step re-reads its own vectors, which nobody consumes): a
                // conditional load would force the waits to cover the path on which it was not issued
                const uint32_t refill = i0 + 2u * kStep <= nvec ? i0 + kStep : i0;  // workgroup-uniform
                uint32_t vote = 0;
#pragma unroll
                for (int r = 0; r < UNROLL; ++r) {
                    const Vec x = cur[r];
                    cur[r] = load(refill + r * THREADS + tid);
                    __builtin_amdgcn_sched_barrier(0);
                    if (r == 0)
                        digit_tables_count_vec<K, TI, true, MSD>(t0, t[0], t[1], t[2], hm, msd_shift, msd_base, msd_over, x, base_shift, lane, vote);
                    else
                        digit_tables_count_vec<K, TI, false, MSD>(t0, t[0], t[1], t[2], hm, msd_shift, msd_base, msd_over, x, base_shift, lane, vote);
                }
            }
        };
        if (static_cast<size_t>(n) * sizeof(K) >= kStreamInBytes) main_loop(std::true_type{});
        else main_loop(std::false_type{});
        for (uint32_t i = i0 + tid; i < nvec; i += THREADS) {
            const Vec q = v[i];
#pragma unroll
            for (int j = 0; j < static_cast<int>(V); ++j)
                digit_tables_count<TI>(t0, t[0], t[1], t[2], hm, msd_shift, msd_base, msd_over, digit_word(KeyVec<K>::get(q, j), base_shift));
        }
        const uint32_t tail = head + nvec * V + tid;  // at most V - 1 keys
        if (tail < len) digit_tables_count<TI>(t0, t[0], t[1], t[2], hm, msd_shift, msd_base, msd_over, digit_word(keys[begin + tail], base_shift));
    }
    __syncthreads();
    if (t0 != nullptr) {
    for (uint32_t d = tid; d < static_cast<uint32_t>(kBins); d += THREADS) {
        uint32_t sum = 0;
#pragma unroll
        for (int r = 0; r < COPIES; ++r) sum += t0_[d * COPIES + ((r + d) % COPIES)];  // skewed: no bank conflicts
        if (sum)
            __hip_atomic_fetch_add(&tables[static_cast<size_t>(s) * kBins + d], sum, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
    }
    for (uint32_t c = tid; c < 3u * GROUPS * kBins; c += THREADS) {  // c = (pass - 1, group, digit)
        const uint32_t x = (&t[0][0])[(c >> 8) * kTableRow + (c & 255u)];
        if (x) __hip_atomic_fetch_add(&tables[GROUPS * kBins + c], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    }
    if constexpr (MSD) {
      if (hm != nullptr) {
        if (__ballot(msd_over != 0u) != 0ull && lane == 0u)
            __hip_atomic_fetch_or(&msd_hist[kMsdOverWord], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // top-14-bit histogram; the top-byte counts of this workgroup's pass-0 group are the sums of 64 sub-bins each
        for (uint32_t c = tid; c < kMsdBuckets; c += THREADS) {
            const uint32_t x = s_msd[c];
            if (x) __hip_atomic_fetch_add(&msd_hist[c], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < kBins) {
            uint32_t sum = 0;
            for (uint32_t j = 0; j < kMsdBuckets / kBins; ++j) sum += s_msd[tid * (kMsdBuckets / kBins) + ((j + tid) % (kMsdBuckets / kBins))];
            if (sum)
                __hip_atomic_fetch_add(&msd_slices[static_cast<size_t>(s) * kBins + tid], sum, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if constexpr (THREADS == 4 * kBins) {
        if (fp.plan != nullptr) {  // fused form: the workgroup that finishes LAST turns the tables into the plan
            __shared__ uint32_t s_last;
            // every lane's atomics above must have been performed before this workgroup's ticket is drawn
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const uint32_t ticket = __hip_atomic_fetch_add(fp.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last = ticket == gridDim.x - 1u ? 1u : 0u;
                if (s_last) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    __hip_atomic_store(fp.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch
                }
            }
            __syncthreads();
            if (s_last)
                plan_body<GROUPS>(tables, fp.plan, fp.host_head, fp.stamp, n, group_len, fp.tile, fp.tile_cap, fp.blind_cap, fp.cuts0);
        }
    }
}

// grid = kStreams * grid_tiles workgroups; block b -> XCD b % 8 -> stream b%8 + 8*((b/8) % (kStreams/8)),
// tile (b/8) / (kStreams/8): every tile's predecessors in its stream sit in lower-numbered blocks of the same XCD.
// The stream's range comes from the plan in device memory (three scalar loads): the host launches the pass before it
// has seen the plan, with room for the longest stream the plan may accept (tile_cap tiles); surplus workgroups leave.
template <typename K, int ITEMS, int WAVES, bool PAIRS, int RANK, int OCC, bool RESERVE = false>
__global__ __launch_bounds__(WAVES * 64, OCC) void onesweep_scatter_kernel(const K *__restrict__ keys_in,
                                                                       K *__restrict__ keys_out,
                                                                       const uint32_t *__restrict__ values_in,
                                                                       uint32_t *__restrict__ values_out,
                                                                       const OnesweepPlan *__restrict__ plan,
                                                                       uint32_t pass, int forced, uint32_t shift,
                                                                       uint32_t *__restrict__ status,
                                                                       unsigned long long xcc_map, int misplace,
                                                                       uint32_t spin_budget, int hold_tile, uint32_t key_base,
                                                                       MsdPlan *__restrict__ reserve, uint32_t *drift) {
    constexpr uint32_t kTile = ITEMS * WAVES * 64;  // the tile the plan counted with (onesweep_tile_keys)
    __shared__ ChunkSmem<K, ITEMS, WAVES, PAIRS> sm;
    const uint32_t k = blockIdx.x >> 3, i = k / (kStreams / 8);
    // misplace (test hook): odd tiles of every stream run on the neighbouring XCD, so the look-back has to work
    // through the write-through copies instead of one L2
    const uint32_t s = ((blockIdx.x + (misplace ? (i & 1u) : 0u)) & 7u) + 8u * (k % (kStreams / 8));
    // the host enqueued this pass before it knew the plan: a pass at or after the first one that needs another form
    // (identity, unbalanced streams) leaves at once and the host enqueues it again, `forced`, in the right order
    // (the plan keeps a second copy of the streams in which such a pass has no tiles: ONE scalar load decides)
    const StreamDesc sd = forced ? plan->head.stream[pass][s] : plan->head.blind[pass][s];
    if (i >= sd.tiles) return;  // uniform per workgroup
    // forced == 2 (vrs_msd_partition_*: the first MSD pass whatever the plan thinks of THIS shard's buckets): still not without
    // counts -- a key range below 27 bits or a key outside the probed range left the bucket histogram empty or wrong, every seed
    // would be void and a reservation could run out of its range
    if (forced == 2 && plan->head.msd_counted == 0u) return;
    report_drift(drift, xcc_map);
    const uint32_t done = i * kTile;
    const uint32_t begin = sd.start + done;
    const bool stream_in = static_cast<size_t>(sd.len) * sizeof(K) * kStreams >= kStreamInBytes;  // (streams are about equal: the pass's input)
    const uint32_t valid = min(kTile, sd.len - done);
    RadixDigit<K> dg;
    dg.shift = shift == kShiftFromPlan ? plan->head.msd_shift_a : shift;  // first MSD pass of the hybrid form: set by msd_plan_kernel
    dg.base = shift == kShiftFromPlan ? static_cast<K>(key_base) : static_cast<K>(0);
    // byte x of xcc_map = XCC of the blocks with blockIdx % 8 == x (probed); my stream's tiles sit in blocks = s (mod 8)
    const bool foreign = xcc_id() != static_cast<uint32_t>((xcc_map >> (8u * (s & 7u))) & 0xFFu);
    uint32_t unused = 0;
    const uint32_t *vin = PAIRS ? values_in + begin : nullptr;
    if constexpr (RESERVE) {
        // first MSD pass over bare keys: the tile reserves its place in (stream, top byte)'s range instead of looking back
        StreamReserve lb;
        const uint32_t d = threadIdx.x & 255u;
        lb.foreign = foreign;
        lb.cursor = &reserve->cursor_a[s][d];
        lb.back = &reserve->back_a[s][d];
        lb.pad_keys = d == dg(dg.template pad<K>()) ? kTile - valid : 0u;
        lb.seed = plan->group_seed[pass][sd.first_group][d];
        if (foreign) {  // the range of (stream, digit) ends where the next stream's begins
            const uint32_t next_group = s + 1u < static_cast<uint32_t>(kStreams) ? plan->head.stream[pass][s + 1u].first_group : 8u;
            lb.region_len = plan->group_seed[pass][next_group][d] - lb.seed;
        }
        if (valid == kTile)
            scatter_chunk<K, ITEMS, WAVES, PAIRS, RANK, true>(sm, keys_in + begin, vin, keys_out, values_out, valid, dg, unused, lb, NoPieces{}, stream_in);
        else
            scatter_chunk<K, ITEMS, WAVES, PAIRS, RANK, false>(sm, keys_in + begin, vin, keys_out, values_out, valid, dg, unused, lb);
    } else {
        StreamLookback lb;
        lb.foreign = foreign;
        lb.hold = hold_tile >= 0 && i == static_cast<uint32_t>(hold_tile);
        lb.stream_keys = keys_in + sd.start;
        lb.done = done;
        if (lb.foreign) {
            // the earlier tiles of the stream are all full: count their digits from the keys themselves
            uint32_t *cnt = sm.whist[0];
            if (threadIdx.x < kBins) cnt[threadIdx.x] = 0;
            __syncthreads();
            recount_keys(cnt, keys_in + sd.start, done, dg);
            __syncthreads();
            if (threadIdx.x < kBins) lb.recounted = cnt[threadIdx.x];
            __syncthreads();
        }
        lb.col = status + static_cast<size_t>(s) * kBins + (threadIdx.x & 255u);
        lb.stride = static_cast<size_t>(kStreams) * kBins;
        lb.index = static_cast<int>(i);
        lb.tag = (pass + 1u) << kLbTagShift;
        lb.budget = spin_budget;
        lb.seed = threadIdx.x < kBins ? plan->group_seed[pass][sd.first_group][threadIdx.x] : 0u;
        if (valid == kTile)
            scatter_chunk<K, ITEMS, WAVES, PAIRS, RANK, true>(sm, keys_in + begin, vin, keys_out, values_out, valid, dg, unused, lb, NoPieces{}, stream_in);
        else
            scatter_chunk<K, ITEMS, WAVES, PAIRS, RANK, false>(sm, keys_in + begin, vin, keys_out, values_out, valid, dg, unused, lb);
    }
    VRS_MARK_FLUSH();
}


uint32_t onesweep_tile_keys(int key_bytes) { return key_bytes == 8 ? 4096u * kLbWaves / 8u : kLbItems * 64u * kLbWaves; }


template <typename K, int GROUPS, int THREADS, int COPIES, int UNROLL, int OCC, bool MSD = false>
static void launch_digit_tables_variant(hipStream_t stream, const void *keys, uint32_t n, uint32_t base_shift,
                                        uint32_t group_len, uint32_t *tables, uint32_t *status, size_t status_words,
                                        int compute_units, LaunchEvents ev, const FusedPlanArgs &fp,
                                        uint32_t *msd_counts = nullptr, uint32_t msd_only = 0, uint32_t msd_base = 0,
                                        uint32_t msd_force_shift = 0) {
    // one workgroup per (pass-0 group, slice): a power-of-two number of slices that fills the chip once
    const uint32_t wgs = static_cast<uint32_t>(compute_units) * (OCC * 256 / THREADS);
    uint32_t slices = floor_pow2(wgs / GROUPS > 0 ? wgs / GROUPS : 1u);
    if (MSD) {
        // every workgroup of the hybrid form's counting read zeroes and flushes 16384 counters whatever it counts: below about
        // 3e7 keys fewer, longer-running workgroups are cheaper (10^7 keys: 64 workgroups, 1 M instead of 4 M counter flushes)
        const uint32_t by_size = floor_pow2(std::max<uint32_t>(n / (GROUPS * 131072u), 1u));
        slices = std::min(slices, by_size);
    }
    const dim3 grid(GROUPS * slices), block(THREADS);
    const uint32_t vecs = static_cast<uint32_t>(status_words / 4);
    VRS_LAUNCH((digit_tables_kernel<K, GROUPS, THREADS, COPIES, UNROLL, OCC, MSD>), grid, block, stream, ev,
               static_cast<const K *>(keys), n, base_shift, group_len, slices, tables, reinterpret_cast<uint4 *>(status), vecs,
               fp, msd_counts, msd_counts ? msd_counts + kMsdBuckets : nullptr, msd_only, msd_base, msd_force_shift);
}

hipError_t launch_digit_tables_msd(hipStream_t stream, const void *keys, uint32_t n, uint32_t group_len, uint32_t *tables,
                                   uint32_t *status, size_t status_words, int compute_units, uint32_t *msd_counts,
                                   bool msd_only, LaunchEvents ev, uint32_t key_base, uint32_t force_shift) {
    launch_digit_tables_variant<uint32_t, 8, 1024, 32, kDtUnroll, 4, true>(stream, keys, n, 0, group_len, tables, status,
                                                                               status_words, compute_units, ev,
                                                                               FusedPlanArgs{}, msd_counts, msd_only ? 1u : 0u, key_base, force_shift);
    return hipGetLastError();
}

hipError_t launch_digit_tables(hipStream_t stream, const void *keys, uint32_t n, int key_bytes, uint32_t base_shift,
                               uint32_t group_len, uint32_t groups, uint32_t *tables, uint32_t *status,
                               size_t status_words, int compute_units, LaunchEvents ev, const FusedPlan *fused) {
    FusedPlanArgs fp{};
    if (fused) {
        fp.plan = fused->plan;
        fp.host_head = fused->host_head;
        fp.done = fused->done;
        fp.stamp = fused->stamp;
        fp.tile = fused->tile;
        fp.tile_cap = fused->tile_cap;
        fp.blind_cap = fused->blind_cap;
        fp.cuts0 = fused->cuts0;
    }
#define VRS_DT(K, G, T, C, U, O) \
    launch_digit_tables_variant<K, G, T, C, U, O>(stream, keys, n, base_shift, group_len, tables, status, status_words, compute_units, ev, fp)
    // (THREADS, COPIES, UNROLL, OCC): one 1024-thread workgroup per CU (32 groups: 131 KiB of LDS counters, 8 groups: 57)
    if (key_bytes == 8) {
        if (groups == 32) VRS_DT(uint64_t, 32, 1024, 32, kDtUnroll, 4);
        else if (groups == 16) VRS_DT(uint64_t, 16, 1024, 32, kDtUnroll, 4);
        else if (groups == 8) VRS_DT(uint64_t, 8, 1024, 32, kDtUnroll, 4);
        else return hipErrorInvalidValue;
    } else {
        if (groups == 32) VRS_DT(uint32_t, 32, 1024, 32, kDtUnroll, 4);
        else if (groups == 16) VRS_DT(uint32_t, 16, 1024, 32, kDtUnroll, 4);
        else if (groups == 8) VRS_DT(uint32_t, 8, 1024, 32, kDtUnroll, 4);
        else return hipErrorInvalidValue;
    }
#undef VRS_DT
    return hipGetLastError();
}

StreamCuts pass0_stream_cuts(uint32_t n, uint32_t group_len, uint32_t groups) {
    const auto start_of = [&](uint32_t g) -> uint32_t {
        const uint64_t a = static_cast<uint64_t>(g) * group_len;
        return static_cast<uint32_t>(a < n ? a : n);
    };
    StreamCuts c;
    c.first_group[0] = 0;
    for (uint32_t k = 1; k < static_cast<uint32_t>(kStreams); ++k) c.first_group[k] = balanced_cut(start_of, n, k, groups);
    c.first_group[kStreams] = groups;
    return c;
}

hipError_t launch_plan(hipStream_t stream, uint32_t *tables, OnesweepPlan *plan, OnesweepPlanHead *host_head,
                       uint32_t stamp, uint32_t n, uint32_t group_len, uint32_t groups, uint32_t tile, uint32_t tile_cap,
                       uint32_t blind_cap, const StreamCuts &cuts0) {
    const dim3 grid(1), block(4 * kBins);
    if (groups == 32)
        hipLaunchKernelGGL(plan_kernel<32>, grid, block, 0, stream, tables, plan, host_head, stamp, n, group_len, tile, tile_cap, blind_cap, cuts0);
    else if (groups == 16)
        hipLaunchKernelGGL(plan_kernel<16>, grid, block, 0, stream, tables, plan, host_head, stamp, n, group_len, tile, tile_cap, blind_cap, cuts0);
    else if (groups == 8)
        hipLaunchKernelGGL(plan_kernel<8>, grid, block, 0, stream, tables, plan, host_head, stamp, n, group_len, tile, tile_cap, blind_cap, cuts0);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_onesweep_scatter(hipStream_t stream, const void *keys_in, void *keys_out, const uint32_t *values_in,
                                   uint32_t *values_out, const OnesweepPlan *plan, uint32_t pass, uint32_t shift,
                                   uint32_t *status, uint32_t grid_tiles, int forced, bool atomic_rank,
                                   unsigned long long xcc_map, int key_bytes, uint32_t spin_budget, int hold_tile,
                                   LaunchEvents ev, bool misplace, uint32_t key_base, MsdPlan *reserve, uint32_t *drift) {
    const int mis = misplace ? 1 : 0, force = forced;
    if (grid_tiles == 0) return hipSuccess;
    const dim3 grid(kStreams * grid_tiles), block(64 * kLbWaves);
    const bool pairs = values_in != nullptr;
#define VRS_ONESWEEP_R(K, ITEMS, PAIRS, RANK, RESERVE)                                                                \
    VRS_LAUNCH((onesweep_scatter_kernel<K, ITEMS, kLbWaves, PAIRS, RANK, 4, RESERVE>), grid, block, stream, ev,   \
               static_cast<const K *>(keys_in), static_cast<K *>(keys_out), values_in, values_out, plan, pass, force,  \
               shift, status, xcc_map, mis, spin_budget, hold_tile, key_base, reserve, drift)
#define VRS_ONESWEEP(K, ITEMS, PAIRS, RANK) VRS_ONESWEEP_R(K, ITEMS, PAIRS, RANK, false)
    // the first MSD pass of the hybrid form over bare keys (LDS-atomic ranking) may take its places by reservation
    if (reserve != nullptr && !pairs && atomic_rank && shift == kShiftFromPlan) {
        if (key_bytes == 8) VRS_ONESWEEP_R(uint64_t, 8, false, RANK_ATOMIC, true);
        else VRS_ONESWEEP_R(uint32_t, kLbItems, false, RANK_ATOMIC, true);
    } else if (key_bytes == 8 && pairs) {  // uint64 keys + uint32 payloads: 4096-pair tiles (32 KB of keys + 16 KB of payloads in LDS)
        if (atomic_rank) VRS_ONESWEEP(uint64_t, 8, true, RANK_ATOMIC); else VRS_ONESWEEP(uint64_t, 8, true, RANK_BALLOT);
    } else if (key_bytes == 8) {
        if (atomic_rank) VRS_ONESWEEP(uint64_t, 8, false, RANK_ATOMIC); else VRS_ONESWEEP(uint64_t, 8, false, RANK_BALLOT);
    } else if (pairs) {
        if (atomic_rank) VRS_ONESWEEP(uint32_t, kLbItems, true, RANK_ATOMIC); else VRS_ONESWEEP(uint32_t, kLbItems, true, RANK_BALLOT);
    } else {
        if (atomic_rank) VRS_ONESWEEP(uint32_t, kLbItems, false, RANK_ATOMIC); else VRS_ONESWEEP(uint32_t, kLbItems, false, RANK_BALLOT);
    }
#undef VRS_ONESWEEP
#undef VRS_ONESWEEP_R
    return hipGetLastError();
}

}  // namespace vrs
