// vrs_capi_msd.hip -- the C ABI, part 5 of 5: the two halves of the hybrid form as calls of their own (vrs_msd_partition_* /
// vrs_msd_finish_*): what a rank of the multi-GPU step runs before and after the exchange (vrs_dist.hip).
#include "vrs_host.hpp"

using namespace vrsh;

static_assert(VRS_MSD_COUNT_WORDS == vrs::kMsdCountWords && VRS_MSD_SHIFT_WORD == vrs::kMsdBucketCount + 8u * 256u, "the public layout of the count words is the kernels' own");

namespace vrsh {

// ---- the hybrid form in two halves, for callers that move the keys between its two MSD passes (vrs_dist_*: the exchange
// between the GPUs sits there).  Both halves only enqueue.
int msd_half_setup(vrs_context ctx, uint32_t n, vrs_context_t::OneRead *st, OneReadGeometry *g, uint32_t bucket_hint,
                          uint32_t pass_b_groups) {
    if (!ctx->xcc_map_valid || !ctx->atomic_rank_verified || !ctx->scatter.atomic_rank)
        return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "the hybrid form needs the look-back placement probe and the LDS-atomic ranking self-test to have passed on this device");
    if (n == 0 || n >= (1u << 30)) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "the hybrid form takes 1 .. 2^30 - 1 keys");
    *st = vrs_context_t::OneRead{};
    st->n = n;
    st->key_bytes = 4;
    st->msd_capable = true;
    st->blind_tail = true;
    st->fast_count = true;
    st->bucket_hint = bucket_hint;
    st->pass_b_groups = pass_b_groups;
    *g = one_read_geometry(ctx, *st);
    return one_read_scratch(ctx, *st, *g);
}
}  // namespace vrsh

extern "C" {

int vrs_msd_partition_u32(vrs_context ctx, vrs_buffer keys, vrs_buffer out, vrs_buffer counts_out, uint32_t n) {
    return vrs_msd_partition_signal_u32(ctx, keys, out, counts_out, n, nullptr);
}

int vrs_msd_partition_signal_u32(vrs_context ctx, vrs_buffer keys, vrs_buffer out, vrs_buffer counts_out, uint32_t n, void *counts_ready_event) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    int rc;
    const size_t bytes = static_cast<size_t>(n) * sizeof(uint32_t);
    if ((rc = check_buffer(ctx, keys, bytes, "keys"))) return rc;
    if ((rc = check_buffer(ctx, out, bytes, "out"))) return rc;
    if ((rc = check_buffer(ctx, counts_out, VRS_MSD_COUNT_WORDS * sizeof(uint32_t), "counts_out"))) return rc;
    if (keys->ptr == out->ptr) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "keys and out alias");
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = settle_pending(ctx))) return rc;
    vrs_context_t::OneRead st;
    OneReadGeometry g;
    if ((rc = msd_half_setup(ctx, n, &st, &g))) return rc;
    if ((rc = reservation_begin(ctx))) return rc;
    ctx->sub_cache.valid = false;
    // the counting read below clears the status words if anything has written them since they were last clear; the first MSD
    // pass then leaves them alone when it reserves, and writes them when it looks back
    const size_t partition_zero_words = ctx->os_status_clean ? 0 : ctx->os_status_rows * VRS_RADIX_SORT_BINS;
    ctx->os_status_clean = reserves(ctx, n, false);
    vrs::LaunchEvents ev;
    if ((rc = profile_events(ctx, VRS_KERNEL_DIGIT_TABLES, &ev))) return rc;
    if (++ctx->os_stamp == 0) ctx->os_stamp = 1;
    // counting read: only the bucket histogram and the slices' top-byte counts (a key range below 27 bits gets the LSD
    // tables instead -- the plan kernel clears them again -- and leaves the histogram empty: the caller sees the shift)
    VRS_HIP(ctx, vrs::launch_digit_tables_msd(ctx->stream, keys->ptr, n, g.group_len, ctx->os_tables, ctx->os_status,
                                              partition_zero_words, ctx->scatter.compute_units, ctx->os_msd_counts, true, ev));
    VRS_HIP(ctx, hipMemcpyAsync(counts_out->ptr, ctx->os_msd_counts, VRS_MSD_COUNT_WORDS * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    // the counts are all a caller needs to start talking to its peers: the first pass below runs meanwhile
    if (counts_ready_event) VRS_HIP(ctx, hipEventRecord(static_cast<hipEvent_t>(counts_ready_event), ctx->stream));
    VRS_HIP(ctx, vrs::launch_msd_plan(ctx->stream, ctx->os_msd_counts, ctx->os_msd_plan, ctx->os_plan_a, ctx->os_plan,
                                      ctx->os_host_head_dev, ctx->os_stamp, n, g.T, g.tiles_b_cap, g.local_cap, ctx->os_tables,
                                      g.group_len, g.tile_cap, g.blind_cap, g.cuts0, 1u, 18u));
    // the first MSD pass, whatever the plan thinks of this shard's buckets (forced: the streams, not their armed copies) -- but not
    // without counts (2: a key range below 27 bits, or a key outside the probed range: the workgroups leave at once, `out` is not
    // written, and the caller, who sees the shift and the flag in counts_out, takes another shape)
    if ((rc = profile_events(ctx, VRS_KERNEL_LOOKBACK_SCATTER, &ev))) return rc;
    VRS_HIP(ctx, vrs::launch_onesweep_scatter(ctx->stream, keys->ptr, out->ptr, nullptr, nullptr, ctx->os_plan_a, 0, vrs::kShiftFromPlan,
                                              ctx->os_status, g.tiles0, 2, ctx->scatter.atomic_rank, ctx->xcc_map, 4,
                                              ctx->os_spin_budget, -1, ev, false, 0u, reserves(ctx, n, false) ? ctx->os_msd_plan : nullptr, drift_word(ctx)));
    return VRS_OK;
}

int vrs_msd_finish_u32(vrs_context ctx, vrs_buffer grouped, vrs_buffer out, vrs_buffer counts, uint32_t n, uint32_t bucket_hint) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    int rc;
    const size_t bytes = static_cast<size_t>(n) * sizeof(uint32_t);
    if ((rc = check_buffer(ctx, grouped, bytes, "grouped"))) return rc;
    if ((rc = check_buffer(ctx, out, bytes, "out"))) return rc;
    if ((rc = check_buffer(ctx, counts, VRS_MSD_COUNT_WORDS * sizeof(uint32_t), "counts"))) return rc;
    if (grouped->ptr == out->ptr) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "grouped and out alias");
    if (reinterpret_cast<uintptr_t>(counts->ptr) % 16u) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "counts must be 16-byte aligned");
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = settle_pending(ctx))) return rc;
    vrs_context_t::OneRead st;
    OneReadGeometry g;
    if ((rc = msd_half_setup(ctx, n, &st, &g, bucket_hint))) return rc;
    if ((rc = reservation_begin(ctx))) return rc;
    ctx->sub_cache.valid = false;
    if (++ctx->os_stamp == 0) ctx->os_stamp = 1;
    ctx->os_msd_half_stamp = ctx->os_stamp;
    // the look-back rows of the second pass must read "never written": the counting read of a whole sort clears them, here
    // nothing else does -- unless the last kernel that touched them was a local sort that cleared them (the previous round's)
    if (!ctx->os_status_clean)
        VRS_HIP(ctx, hipMemsetAsync(ctx->os_status, 0, ctx->os_status_rows * VRS_RADIX_SORT_BINS * sizeof(uint32_t), ctx->stream));
    ctx->os_status_clean = false;
    // the plan reads the caller's table in place (and leaves its histogram zeroed, like the context's own)
    VRS_HIP(ctx, vrs::launch_msd_plan(ctx->stream, static_cast<uint32_t *>(counts->ptr), ctx->os_msd_plan, ctx->os_plan_a, ctx->os_plan,
                                      ctx->os_host_head_dev, ctx->os_stamp, n, g.T, g.tiles_b_cap, g.local_cap, ctx->os_tables,
                                      g.group_len, g.tile_cap, g.blind_cap, g.cuts0, 1u, 18u,
                                      reinterpret_cast<uint32_t *>(ctx->os_host_head_dev + 1)));
    st.kptr[0] = out->ptr;      // "home": the second pass writes here, the local sort works here
    st.kptr[1] = grouped->ptr;  // the partner holds the first pass's output
    st.cur_at_start = 0;
    return one_read_hybrid_tail(ctx, st, g, g.tiles_b_cap, g.local_cap, true);
}

int vrs_msd_finish_grouped_u32(vrs_context ctx, vrs_buffer grouped, vrs_buffer out, uint32_t n, uint32_t first_top_byte,
                               uint32_t top_bytes) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    int rc;
    const size_t bytes = static_cast<size_t>(n) * sizeof(uint32_t);
    if ((rc = check_buffer(ctx, grouped, bytes, "grouped"))) return rc;
    if ((rc = check_buffer(ctx, out, bytes, "out"))) return rc;
    if (grouped->ptr == out->ptr) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "grouped and out alias");
    if (top_bytes == 0 || first_top_byte > 255u || first_top_byte + top_bytes > 256u)
        return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "top bytes [first, first + count) must lie in [0, 256)");
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = settle_pending(ctx))) return rc;
    // the 14-bit bucket index = (top byte - first) in its high bits, the next sub_bits of the key below: as many as the
    // second pass can sort by (8) while all groups fit the 16384 buckets
    uint32_t group_bits = 0;
    while ((1u << group_bits) < top_bytes) ++group_bits;
    const uint32_t sub_bits = std::min(8u, 14u - group_bits), shift = 24u - sub_bits;
    const uint32_t key_base = first_top_byte << 24;
    vrs_context_t::OneRead st;
    OneReadGeometry g;
    const uint32_t hint = static_cast<uint32_t>(std::min<uint64_t>((static_cast<uint64_t>(n) * 9u / 8u) / (static_cast<uint64_t>(top_bytes) << sub_bits) + 64u, 0xFFFFFFFFu));
    if ((rc = msd_half_setup(ctx, n, &st, &g, hint, top_bytes))) return rc;
    if ((rc = reservation_begin(ctx))) return rc;
    st.key_base = key_base;
    st.sub_bits = sub_bits;
    ctx->sub_cache.valid = false;
    if (++ctx->os_stamp == 0) ctx->os_stamp = 1;
    ctx->os_msd_half_stamp = ctx->os_stamp;
    // the counting read clears the look-back rows unless the last kernel that touched them left them clear
    const size_t zero_words = ctx->os_status_clean ? 0 : ctx->os_status_rows * VRS_RADIX_SORT_BINS;
    ctx->os_status_clean = false;
    vrs::LaunchEvents ev;
    if ((rc = profile_events(ctx, VRS_KERNEL_DIGIT_TABLES, &ev))) return rc;
    VRS_HIP(ctx, vrs::launch_digit_tables_msd(ctx->stream, grouped->ptr, n, g.group_len, ctx->os_tables, ctx->os_status, zero_words,
                                              ctx->scatter.compute_units, ctx->os_msd_counts, true, ev, key_base, shift));
    VRS_HIP(ctx, vrs::launch_msd_plan(ctx->stream, ctx->os_msd_counts, ctx->os_msd_plan, ctx->os_plan_a, ctx->os_plan,
                                      ctx->os_host_head_dev, ctx->os_stamp, n, g.T, g.tiles_b_cap, g.local_cap, ctx->os_tables,
                                      g.group_len, g.tile_cap, g.blind_cap, g.cuts0, 1u, 18u,
                                      reinterpret_cast<uint32_t *>(ctx->os_host_head_dev + 1), sub_bits));
    st.kptr[0] = out->ptr;
    st.kptr[1] = grouped->ptr;
    st.cur_at_start = 0;
    return one_read_hybrid_tail(ctx, st, g, g.tiles_b_cap, g.local_cap, true);
}

int vrs_msd_finish_grouped_counts_u32(vrs_context ctx, vrs_buffer grouped, vrs_buffer out, uint32_t n, uint32_t first_top_byte,
                                      uint32_t top_bytes, const uint32_t *counts) {
    return vrs_msd_finish_grouped_split_u32(ctx, grouped, nullptr, 0, out, n, first_top_byte, top_bytes, counts, nullptr);
}
}  // extern "C"

namespace vrsh {

// the own parts into the holes of the grouped buffer (neighbouring ones merged: device copies are launch-bound below a megabyte)
int fill_own_holes(vrs_context ctx, vrs_buffer grouped, vrs_buffer own, uint64_t own_offset, uint32_t top_bytes, const uint32_t *counts,
                          const uint32_t *own_counts) {
    uint32_t *dst = static_cast<uint32_t *>(grouped->ptr);
    const uint32_t *src = static_cast<const uint32_t *>(own->ptr) + own_offset;
    uint64_t at = 0, from = 0, run_dst = 0, run_src = 0, run_len = 0;
    for (uint32_t a = 0; a <= top_bytes; ++a) {
        const uint64_t hole = a < top_bytes ? at + counts[a] - own_counts[a] : 0, len = a < top_bytes ? own_counts[a] : 0;
        if (a < top_bytes && len && run_len && run_dst + run_len == hole && run_src + run_len == from) {
            run_len += len;
        } else {
            if (run_len) VRS_HIP(ctx, hipMemcpyAsync(dst + run_dst, src + run_src, run_len * 4, hipMemcpyDeviceToDevice, ctx->stream));
            run_dst = hole;
            run_src = from;
            run_len = len;
        }
        if (a < top_bytes) {
            at += counts[a];
            from += len;
        }
    }
    return VRS_OK;
}
}  // namespace vrsh

extern "C" {

int vrs_msd_finish_grouped_split_u32(vrs_context ctx, vrs_buffer grouped, vrs_buffer own, uint64_t own_offset, vrs_buffer out, uint32_t n,
                                     uint32_t first_top_byte, uint32_t top_bytes, const uint32_t *counts, const uint32_t *own_counts) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    if (!counts) {
        if (own || own_counts) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "own keys elsewhere need the top bytes' counts");
        return vrs_msd_finish_grouped_u32(ctx, grouped, out, n, first_top_byte, top_bytes);
    }
    int rc;
    const size_t bytes = static_cast<size_t>(n) * sizeof(uint32_t);
    if ((rc = check_buffer(ctx, grouped, bytes, "grouped"))) return rc;
    if ((rc = check_buffer(ctx, out, bytes, "out"))) return rc;
    if (grouped->ptr == out->ptr) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "grouped and out alias");
    if (top_bytes == 0 || first_top_byte > 255u || first_top_byte + top_bytes > 256u)
        return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "top bytes [first, first + count) must lie in [0, 256)");
    if ((own != nullptr) != (own_counts != nullptr)) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "own and own_counts go together");
    vrs::PoolGroups groups{};
    uint64_t sum = 0, own_sum = 0;
    for (uint32_t a = 0; a < top_bytes; ++a) {
        groups.count[a] = counts[a];
        sum += counts[a];
        if (own_counts) {
            if (own_counts[a] > counts[a]) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "a top byte has more own keys than keys");
            groups.own[a] = own_counts[a];
            own_sum += own_counts[a];
        }
    }
    groups.top_bytes = top_bytes;
    if (sum != n) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "the top bytes' counts do not add up to num_elements");
    if (own_sum == 0) own = nullptr;  // (nothing lies elsewhere)
    if (own) {
        if ((rc = check_buffer(ctx, own, (own_offset + own_sum) * sizeof(uint32_t), "own"))) return rc;
        if (own->ptr == out->ptr || own->ptr == grouped->ptr) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "own aliases grouped or out");
        if (own_offset + own_sum > 0x7FFFFFFFull) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "own keys beyond slot 2^31");
        groups.own_first = static_cast<uint32_t>(own_offset);
    }
    // where the form cannot run, the counted finish takes over -- over keys in ONE piece: the own parts are copied into their holes first
    const auto counted = [&]() -> int {
        if (own) {
            VRS_HIP(ctx, hipSetDevice(ctx->device));
            if (const int e = settle_pending(ctx)) return e;
            if (const int e = fill_own_holes(ctx, grouped, own, own_offset, top_bytes, counts, own_counts)) return e;
        }
        return vrs_msd_finish_grouped_u32(ctx, grouped, out, n, first_top_byte, top_bytes);
    };
    // The pool form's second half: the plan samples the grouped keys (nothing is read to be counted), the second pass scatters into
    // the buckets' slack regions, the local sort finishes.  Where it cannot run -- the form switched off, no shape for these buckets,
    // fewer keys than its fixed costs are worth -- the counted finish takes over.
    const vrs::PoolShape shape = vrs::pool_grouped_shape(n, top_bytes);
    if (ctx->os_pool == 0 || !reserves(ctx, n, false) || shape.sub_bits == 0u || n < (1u << 20) || n >= (1u << 30) || !ctx->xcc_map_valid ||
        !ctx->atomic_rank_verified || !ctx->scatter.atomic_rank)
        return counted();
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = settle_pending(ctx))) return rc;
    vrs_context_t::OneRead st;
    OneReadGeometry g;
    if ((rc = msd_half_setup(ctx, n, &st, &g, 0, top_bytes))) return rc;  // (the plan head, its host copy and the log of decisions)
    // rows of workgroups of the second pass: the busiest XCD's tiles (XCD x walks top bytes x, x + 8, ...) -- known exactly here
    uint32_t tiles_b = 0;
    for (uint32_t x = 0; x < 8u; ++x) {
        uint32_t t = 0;
        for (uint32_t a = x; a < top_bytes; a += 8u) t += (groups.count[a] + vrs::kPoolTile - 1u) / vrs::kPoolTile;
        tiles_b = std::max(tiles_b, t);
    }
    if (tiles_b > vrs::kPoolMaxTilesB) return counted();
    const uint32_t slack = vrs::pool_slack_capacity(n, shape.sub_bits, top_bytes);
    if ((rc = pool_scratch(ctx, std::max(ctx->os_pool_overflow_cap, 32u), slack))) return rc == kPoolNoMemory ? counted() : rc;  // (no room for the slack buffer: the counted finish needs none)
    ctx->sub_cache.valid = false;
    if (++ctx->os_stamp == 0) ctx->os_stamp = 1;
    ctx->os_msd_half_stamp = ctx->os_stamp;
    ctx->os_pool_layout_valid = false;  // (the plan of grouped keys rewrites words a kept layout rests on: PoolPlan::shift)
    const uint32_t par = (++ctx->os_pool_epoch) & 1u;
    const uint32_t key_base = first_top_byte << 24;
    const uint32_t *keys_in = static_cast<const uint32_t *>(grouped->ptr);
    const uint32_t *keys_own = own ? static_cast<const uint32_t *>(own->ptr) : keys_in;  // (virtual slots from n on)
    vrs::LaunchEvents ev;
    VRS_HIP(ctx, vrs::launch_pool_plan(ctx->stream, ctx->os_msd_plan, ctx->os_pool_plan, n, tiles_b, ctx->os_pool_slack_cap, keys_in, keys_own, key_base,
                                       vrs::pool_streams(n), shape.sub_bits, par, &groups));
    if ((rc = profile_events(ctx, VRS_KERNEL_POOL_PASS_B, &ev))) return rc;
    VRS_HIP(ctx, vrs::launch_pool_pass_b(ctx->stream, keys_in, keys_own, ctx->os_pool_slack, n, ctx->os_msd_plan, ctx->os_pool_plan, tiles_b, key_base,
                                         vrs::pool_local_capacity(shape.local), ctx->os_pool_slack_cap, ctx->xcc_map, ctx->os_stamp, shape.sub_bits, par, ev, true));
    if ((rc = profile_events(ctx, VRS_KERNEL_LOCAL_SORT, &ev))) return rc;
    VRS_HIP(ctx, vrs::launch_pool_local_sort(ctx->stream, ctx->os_pool_slack, static_cast<uint32_t *>(out->ptr), n, ctx->os_msd_plan, ctx->os_pool_plan, shape,
                                             &ctx->os_plan->head, ctx->os_host_head_dev, ctx->os_stamp, par, ev, top_bytes,
                                             reinterpret_cast<uint32_t *>(ctx->os_host_head_dev + 1)));
    return VRS_OK;
}

int vrs_msd_finish_status(vrs_context ctx, int *took) {
    if (!ctx || !took) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "context or took is NULL");
    *took = 0;
    if (ctx->os_msd_half_stamp == 0) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "no vrs_msd_finish_u32 to ask about");
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    const int rc = wait_for_plan(ctx, ctx->os_msd_half_stamp);
    if (rc) return rc;
    *took = ctx->os_host_head->msd_ok ? 1 : 0;
    return VRS_OK;
}

int vrs_msd_finish_ticket(vrs_context ctx, uint32_t *ticket) {
    if (!ctx || !ticket) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "context or ticket is NULL");
    if (ctx->os_msd_half_stamp == 0) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "no vrs_msd_finish_u32 to ask about");
    *ticket = ctx->os_msd_half_stamp;
    return VRS_OK;
}

int vrs_msd_finish_status_at(vrs_context ctx, uint32_t ticket, int *took) {
    if (!ctx || !took) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "context or took is NULL");
    *took = 0;
    if (ticket == 0 || !ctx->os_host_head) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "not a ticket of vrs_msd_finish_ticket");
    // stamps count up: a ticket ahead of the last finish this context enqueued was never handed out (nothing would ever write its word)
    if (ctx->os_msd_half_stamp == 0 || static_cast<int32_t>(ticket - ctx->os_msd_half_stamp) > 0)
        return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "not a ticket of vrs_msd_finish_ticket");
    // The log word of a ticket is (stamp % 32): only plans of vrs_msd_finish_* write the log, so the word keeps this ticket's
    // decision until ANOTHER finish plan whose stamp is congruent to it is made -- however many plans of other kinds (partitions,
    // ranged sorts, recounts) come in between.  The word itself says whose decision it holds.
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    volatile uint32_t *word = reinterpret_cast<volatile uint32_t *>(ctx->os_host_head + 1) + (ticket & (vrs::kMsdLogWords - 1u));
    const uint32_t want = ticket << 1;
    bool overwritten = false, never = false;
    const int rc = wait_for_host_word(ctx, [&] {
        const uint32_t w = __atomic_load_n(word, __ATOMIC_ACQUIRE);
        if ((w & ~1u) == want) return true;
        // a later plan's decision in this word (stamps count up; a word that is zero or older has not been written yet)
        overwritten = w != 0u && static_cast<int32_t>((w >> 1) - (ticket & 0x7FFFFFFFu)) > 0;
        return overwritten;
    }, &never);
    if (never) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "not a ticket of vrs_msd_finish_ticket, or one too old: the stream is idle and the log does not hold its decision");
    if (rc) return rc;
    if (overwritten) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "ticket too old: a later vrs_msd_finish plan has taken its place in the log");
    *took = static_cast<int>(*word & 1u);
    return VRS_OK;
}
}  // extern "C"
