// vrs_plan.hpp -- the plan of the one-call sort's four LSD passes as a device function of one 1024-thread workgroup: the standalone
// plan kernel and the fused tail of the counting read (vrs_one_call.hip), and the first part of the hybrid form's plan kernel
// (vrs_msd_hybrid.hip).
#pragma once
#include "vrs_device.hpp"

namespace vrs {

// one workgroup; thread (p, d).  Merges the GROUPS groups of every pass into kStreams streams of (nearly) equal
// length -- cuts only between groups, so a stream is still a contiguous range of the pass's input and its seed is a
// prefix over whole groups -- and leaves `tables` zeroed for the next sort.  The head goes to device memory (the
// scatter workgroups read their stream from it) and, with system-scope stores, to the pinned host copy (stamp last).
// A device function of one 1024-thread workgroup: the standalone plan kernel, or the tail of the counting read's LAST
// workgroup (fused form: no kernel of its own).  The tables are read with agent-scope loads -- in the fused form they
// were written by other workgroups' atomics in the same launch.
template <int GROUPS>
__device__ __forceinline__ void plan_body(uint32_t *__restrict__ tables, OnesweepPlan *__restrict__ plan,
                                          OnesweepPlanHead *host_head, uint32_t stamp, uint32_t n, uint32_t group_len,
                                          uint32_t tile, uint32_t tile_cap, uint32_t blind_cap, const StreamCuts &cuts0) {
    constexpr uint32_t kGroupDigits = kBins / GROUPS;  // digit values of pass p-1 per group of pass p
    __shared__ uint32_t s_prefix[4][kBins + 1];
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_max[4], s_const[4];
    __shared__ uint32_t s_cut[4][kStreams + 1];  // first group of every stream
    __shared__ OnesweepPlanHead s_head;
    const uint32_t tid = threadIdx.x, p = tid >> 8, d = tid & 255u, lane = tid & 63u, wave = tid >> 6;
    uint32_t before[GROUPS];
    uint32_t total = 0;
#pragma unroll
    for (int g = 0; g < GROUPS; ++g)
        before[g] = __hip_atomic_load(&tables[(static_cast<size_t>(p) * GROUPS + g) * kBins + d], __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
        const uint32_t c = before[g];
        before[g] = total;
        total += c;
        tables[(static_cast<size_t>(p) * GROUPS + g) * kBins + d] = 0;
    }
    uint32_t incl = total;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t x = __shfl_up(incl, o);
        if (lane >= static_cast<uint32_t>(o)) incl += x;
    }
    if (lane == 63u) s_wave[wave] = incl;
    if (tid < 4) {
        s_max[tid] = 0;
        s_const[tid] = 0;
    }
    __syncthreads();
    if (total == n) s_const[p] = 1;  // one digit value holds every key
    uint32_t base = 0;
    for (uint32_t j = p * 4u; j < wave; ++j) base += s_wave[j];
    const uint32_t digit_start = base + incl - total;
    s_prefix[p][d] = digit_start;
    if (d == 255u) s_prefix[p][kBins] = n;
    __syncthreads();
    // where group g of pass q starts in the pass's input
    const auto group_start = [&](uint32_t q, uint32_t g) -> uint32_t {
        if (q == 0) {
            const uint64_t a = static_cast<uint64_t>(g) * group_len;
            return static_cast<uint32_t>(a < n ? a : n);
        }
        return s_prefix[q - 1][g * kGroupDigits];  // g == GROUPS -> n
    };
    if (tid < 4u * kStreams) {  // thread (q, k): the cut between streams k-1 and k of pass q
        const uint32_t q = tid / kStreams, k = tid % kStreams;
        uint32_t cut = 0;
        if (q == 0)  // slices of the input: the host made these cuts (it sizes pass 0's grid from them)
            cut = cuts0.first_group[k];
        else if (k > 0)
            cut = balanced_cut([&](uint32_t g) { return s_prefix[q - 1][g * kGroupDigits]; }, n, k, GROUPS);  // [GROUPS] -> n
        s_cut[q][k] = cut;
        if (k == 0) s_cut[q][kStreams] = GROUPS;
    }
    // where digit d of every group starts in the pass's output; a stream's seed is the row of its first group
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) plan->group_seed[p][g][d] = digit_start + before[g];
    plan->group_seed[p][GROUPS][d] = digit_start + total;
    __syncthreads();
    if (tid < 4u * kStreams) {
        const uint32_t q = tid / kStreams, s = tid % kStreams;
        const uint32_t start = group_start(q, s_cut[q][s]), end = group_start(q, s_cut[q][s + 1]);
        const uint32_t tiles = (end - start + tile - 1u) / tile;
        s_head.stream[q][s] = StreamDesc{start, end - start, s_cut[q][s], tiles};
        atomicMax(&s_max[q], tiles);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t first = 4;
        for (int q = 3; q >= 0; --q) {
            // pass 0's grid is sized by the host from its own cuts; passes 1-3 were enqueued with blind_cap rows
            const uint32_t mode = s_const[q]                       ? kPassIdentity
                                  : s_max[q] > tile_cap            ? kPassUnbalanced
                                  : (q > 0 && s_max[q] > blind_cap) ? kPassLookbackWide
                                                                    : kPassLookback;
            s_head.max_tiles[q] = s_max[q];
            s_head.mode[q] = mode;
            if (mode != kPassLookback) first = static_cast<uint32_t>(q);
        }
        s_head.first_abnormal = first;
        s_head.msd_ok = 0;  // the hybrid form's fields: msd_plan_kernel fills them in when it runs
        s_head.msd_tiles_b = 0;
        s_head.msd_max_bucket = 0;
        s_head.msd_shift_a = 0;
        s_head.lsd_missing = 0;
        s_head.ready = 0;
    }
    __syncthreads();
    if (tid < 4u * kStreams) {  // what a speculatively enqueued pass sees: no tiles from the first abnormal pass on
        const uint32_t q = tid / kStreams, k = tid % kStreams;
        StreamDesc d = s_head.stream[q][k];
        if (q >= s_head.first_abnormal) d.tiles = 0;
        s_head.blind[q][k] = d;
    }
    __syncthreads();
    constexpr uint32_t kHeadWords = sizeof(OnesweepPlanHead) / sizeof(uint32_t);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&s_head);
    static_assert(offsetof(OnesweepPlanHead, drift) == sizeof(OnesweepPlanHead) - 8 && offsetof(OnesweepPlanHead, ready) == sizeof(OnesweepPlanHead) - 4,
                  "the last two words are not the plan's: `drift` (kernels add to the host copy) and `ready` (the stamp)");
    for (uint32_t i = tid; i < kHeadWords - 2u; i += 4 * kBins)  // every word but `drift` and `ready` (the last two)
        reinterpret_cast<uint32_t *>(&plan->head)[i] = src[i];
    if (host_head && tid < 64u) {  // ONE wave writes the host copy, so one wave's fence orders it before the stamp
        for (uint32_t i = tid; i < kHeadWords - 2u; i += 64u)
            __hip_atomic_store(reinterpret_cast<uint32_t *>(host_head) + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        // stamp == 0: another kernel (msd_plan_kernel) completes the head and stamps it
        if (tid == 0 && stamp != 0u) __hip_atomic_store(&host_head->ready, stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace vrs
