// vrs_sort_form.hpp -- WHICH FORM a one-call sort takes (vrs_sort_keys_u32 / _u64 / vrs_sort_pairs_u32 / _u64), as one pure function of
// the size, the kind of sort and what the context is set to and remembers: the dispatcher (vrs_capi_sort.hip: sort_all_passes,
// one_read_enqueue) asks it, and so does the host-only entry point vrs_sort_form_for, through which tests/test_capi_cpu.py walks the
// whole decision table without a device.  No HIP here.
//   single    ONE launch of the single-workgroup kernel (the reference's guidance for small inputs, README.md:18-21)
//   contract  the reference's two stages, pass by pass (4 or 8 x [histogram, prefix, scatter]: 12 B per key and pass)
//   lsd       one counting read + one look-back scatter pass per key byte (36 B/key for uint32 keys)
//   counted   the counted hybrid form: counting read, two MSD passes, LDS-local sort (28 B/key)
//   pool      the hybrid form without its counting read (24 B/key; pairs: its stable variant, 48 B/pair)
// A form a verdict on the device refuses (a key range below 27 bits, a misjudged sample, a bucket no workgroup holds) starts over one
// form down with `no_pool` / `no_hybrid` set: the same function answers for the retry.
#pragma once
#include "vrs_kernels.h"

namespace vrs {

enum SortFormId : int { kFormNone = 0, kFormSingle = 1, kFormContract = 2, kFormLsd = 3, kFormCounted = 4, kFormPool = 5 };

struct SortKnobs {
    // settings (vrs_set_tuning) and what the context found out about the device
    uint32_t single_max_keys = 4096;      // VRS_TUNE_SINGLE_MAX_KEYS
    uint32_t one_call_min_keys = 1u << 13;  // VRS_TUNE_ONE_CALL_MIN_KEYS (0 = never: always the contract stages)
    uint32_t hybrid_min_keys = 0;         // VRS_TUNE_HYBRID_MIN_KEYS (0 = the measured crossovers: 1.3e7 keys, 2.5e7 pairs, 2e7 64-bit keys)
    uint32_t pool_min_keys = 1u << 22;    // VRS_TUNE_MSD_POOL_MIN_KEYS
    int hybrid = 1;                       // VRS_TUNE_HYBRID
    int pool = 1;                         // VRS_TUNE_MSD_POOL: 0 never, 1 adaptive, 2 always
    int pool_pairs = 1;                   // VRS_TUNE_MSD_POOL_PAIRS
    int reserve = 1;                      // VRS_TUNE_MSD_RESERVE
    uint32_t groups = 0;                  // VRS_TUNE_DIGIT_TABLE_GROUPS (0 = by size)
    bool xcc_map_valid = true;            // the placement probe found "block b on the XCC of b % 8"
    bool atomic_rank = true;              // the lane-order self-test passed and the atomic ranking is in effect
    // what the context remembers of earlier sorts
    uint32_t pool_skip = 0, pool_skip_n = 0;  // sorts left that skip the pool form after a refusal of a sort of pool_skip_n keys
    bool wide_refused = false;            // 64-bit keys: the last attempt at the hybrid form was refused ...
    uint32_t wide_skipped = 0;            //   ... sorts since (every 16th tries again)
    // this sort is a retry after a refusal
    bool no_pool = false, no_hybrid = false;
};

struct SortDecision {
    SortFormId form = kFormNone;
    bool msd_capable = false;   // the hybrid form (counted or pool) may take this sort
    bool pool_candidate = false, pool = false;
    uint32_t pool_skip = 0, wide_skipped = 0;  // the context's memory after this decision (the adaptive skips count down)
};

inline SortDecision sort_form_for(uint32_t n, int key_bytes, bool pairs, const SortKnobs &k) {
    SortDecision d;
    d.pool_skip = k.pool_skip;
    d.wide_skipped = k.wide_skipped;
    if (n == 0) return d;
    const bool wide = key_bytes == 8;
    if (!wide && !pairs && n <= k.single_max_keys) {
        d.form = kFormSingle;
        return d;
    }
    // (the look-back status words carry 28-bit stream counts: 2^30 keys and more run the contract stages)
    if (!(k.xcc_map_valid && k.one_call_min_keys != 0 && n >= k.one_call_min_keys && n < (1u << 30))) {
        d.form = kFormContract;
        return d;
    }
    // Hybrid form: uint32 keys, with or without uint32 payloads, and 64-bit keys, from hybrid_min on -- below, the fixed costs of the
    // 16384-bin counting read and of a launch per bucket outweigh the saved pass (profiles/labs/r03_hybrid_by_size.txt, r02_hybrid_pairs.txt,
    // r02_hybrid_u64.txt); above about 2.3e8 uniform keys the largest bucket no longer fits a workgroup's LDS and the plan says no.
    const uint32_t set = k.hybrid_min_keys;
    const uint32_t hybrid_min = set == 0u ? (wide ? 20000000u : pairs ? 25000000u : 13000000u) : (wide ? set / 2u : pairs ? set / 8u * 5u : set);
    const bool reserves_bare = k.reserve != 0;
    // bare uint32 keys the pool form may take: from ITS threshold on (below the counted form's: its first half costs a sample, not a counting read)
    const bool skip_applies = !(k.pool == 2 || k.pool_skip == 0 || n / 2u > k.pool_skip_n || n < k.pool_skip_n / 2u);
    const bool pool_size = !pairs && !wide && !k.no_pool && k.pool != 0 && reserves_bare && n >= k.pool_min_keys && n <= kPoolMaxKeys && !skip_applies;
    bool wide_try = wide;
    if (wide_try && !k.no_hybrid && k.wide_refused && (++d.wide_skipped % 16u) != 0u) wide_try = false;  // after a refusal only every 16th 64-bit sort tries again
    const uint64_t local_cap = wide && pairs ? msd_local_capacity_pairs_u64(false) : msd_local_capacity(pairs || wide);
    d.msd_capable = !k.no_hybrid && (!wide || wide_try) && k.hybrid != 0 && k.atomic_rank && (n >= hybrid_min || pool_size) && n >= (1u << 22) &&
                    static_cast<uint64_t>(n) <= 2ull * kMsdBucketCount * local_cap && (k.groups == 0 || k.groups == 8);
    // Pool form: a refusal costs the first pass, so the default is adaptive -- after one, the next 15 such sorts of the context take the counted
    // form (a sort of another size class, beyond a factor of two, is another workload and starts afresh).  Sizes: a bucket must fit the local
    // sort's larger shape (uniform keys up to kPoolMaxKeys; pairs up to pool_max_pairs()).
    d.pool_candidate = d.msd_capable && !wide && !k.no_pool && k.pool != 0 && n >= k.pool_min_keys && n <= kPoolMaxKeys &&
                       (pairs ? k.pool_pairs != 0 && n <= pool_max_pairs() : reserves_bare);
    if (d.pool_candidate && d.pool_skip && (n / 2u > k.pool_skip_n || n < k.pool_skip_n / 2u)) d.pool_skip = 0;
    d.pool = d.pool_candidate && (k.pool == 2 || d.pool_skip == 0);
    if (d.pool_candidate && !d.pool) --d.pool_skip;
    d.form = d.msd_capable ? (d.pool ? kFormPool : kFormCounted) : kFormLsd;
    return d;
}

}  // namespace vrs
