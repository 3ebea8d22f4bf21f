// vrs_capi_pool.hip -- the C ABI, part 4 of 5: the host side of the pool form (vrs_msd_pool.hip): its scratch (allocated, grown, given
// back; a device without room for it is no error of a sort), the enqueue of a whole pool sort, its shape and statistics.
#include "vrs_host.hpp"

using namespace vrsh;

namespace vrsh {

// the pool form's scratch: its plan (once), the first pass's overflow regions and the slack buffer (grown when a sort needs more).
// kPoolNoMemory: the device has no room for it (about 1.5 n slots, twice that for pairs) -- not an error of the SORT: whatever was
// allocated is released and the caller takes a form that needs no such scratch (the counted form, the LSD passes).
void pool_scratch_release(vrs_context ctx, bool payloads_only) {
    const auto drop = [](uint32_t *&p, uint32_t &cap) {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    };
    drop(ctx->os_pool_overflow_vals, ctx->os_pool_vals_overflow_cap);
    drop(ctx->os_pool_slack_vals, ctx->os_pool_vals_slack_cap);
    if (payloads_only) return;
    drop(ctx->os_pool_overflow, ctx->os_pool_overflow_cap);
    drop(ctx->os_pool_slack, ctx->os_pool_slack_cap);
    ctx->os_pool_layout_valid = false;  // (a kept layout speaks of slots of the buffers that just went)
}
int pool_scratch(vrs_context ctx, uint32_t room, uint32_t slack, bool pairs) {
    const auto alloc = [&](uint32_t *&p, uint32_t &cap, uint32_t slots) -> hipError_t {
        if (ctx->os_pool_fail_alloc) {  // test hook (VRS_TUNE_DEBUG_POOL_NO_MEMORY): as if the device were full
            --ctx->os_pool_fail_alloc;
            return hipErrorOutOfMemory;
        }
        const hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), static_cast<size_t>(slots) * sizeof(uint32_t));
        if (e == hipSuccess) cap = slots;
        else p = nullptr;
        return e;
    };
    const auto no_room = [&](hipError_t e, bool payloads_only) -> int {
        (void)hipGetLastError();  // (the failed hipMalloc is no sticky error of the stream's work)
        pool_scratch_release(ctx, payloads_only);
        ctx->os_pool_no_memory++;
        if (e == hipErrorOutOfMemory) return kPoolNoMemory;
        return fail_hip(ctx, "pool form scratch allocation", e);
    };
    if (!ctx->os_pool_plan) {
        vrs::PoolPlan *pp = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&pp), sizeof(vrs::PoolPlan));
        if (e == hipSuccess) e = hipMemsetAsync(pp, 0, sizeof(vrs::PoolPlan), ctx->stream);  // sample counts, flags: zero between sorts
        if (e != hipSuccess) {
            if (pp) (void)hipFree(pp);
            return no_room(e, false);
        }
        ctx->os_pool_plan = pp;
    }
    if (room > ctx->os_pool_overflow_cap || slack > ctx->os_pool_slack_cap) {
        room = std::max(room, ctx->os_pool_overflow_cap);  // (both are made anew: neither may shrink)
        slack = std::max(slack, ctx->os_pool_slack_cap);
        if (ctx->os_pool_overflow || ctx->os_pool_slack) {
            VRS_HIP(ctx, hipStreamSynchronize(ctx->stream));
            pool_scratch_release(ctx);  // (the payloads' twins with them: they are made to the keys' sizes)
        }
        hipError_t e = alloc(ctx->os_pool_overflow, ctx->os_pool_overflow_cap, room);
        if (e == hipSuccess) e = alloc(ctx->os_pool_slack, ctx->os_pool_slack_cap, slack);
        if (e != hipSuccess) return no_room(e, false);
    }
    if (pairs && (ctx->os_pool_vals_overflow_cap < ctx->os_pool_overflow_cap || ctx->os_pool_vals_slack_cap < ctx->os_pool_slack_cap)) {
        // the payloads' twins: the keys' sizes, so that one slot number serves both
        if (ctx->os_pool_overflow_vals || ctx->os_pool_slack_vals) {
            VRS_HIP(ctx, hipStreamSynchronize(ctx->stream));
            pool_scratch_release(ctx, true);
        }
        hipError_t e = alloc(ctx->os_pool_overflow_vals, ctx->os_pool_vals_overflow_cap, ctx->os_pool_overflow_cap);
        if (e == hipSuccess) e = alloc(ctx->os_pool_slack_vals, ctx->os_pool_vals_slack_cap, ctx->os_pool_slack_cap);
        if (e != hipSuccess) return no_room(e, true);
    }
    return VRS_OK;
}

int one_read_enqueue_pool(vrs_context ctx, const OneReadGeometry &g) {
    (void)g;
    vrs_context_t::OneRead &st = ctx->one_read;
    const uint32_t n = st.n;
    int rc;
    const bool pairs = st.vptr[0] != nullptr;
    const vrs::PoolCut cut = vrs::pool_cut(n, pairs, ctx->os_pool_top_bits, pairs ? 0 : ctx->os_pool_sub_bits);
    const vrs::PoolShape shape{cut.sub_bits, cut.local};
    const uint32_t top_bits = cut.top_bits, top_bytes = 1u << top_bits;
    st.pool_top_bits = top_bits;
    st.pool_sub_bits = shape.sub_bits;
    st.pool_local = shape.local;
    st.pool_retried = false;
    const uint32_t room = vrs::pool_overflow_capacity(n), slack = vrs::pool_slack_capacity(n, shape.sub_bits, top_bytes);
    if ((rc = pool_scratch(ctx, room, slack, pairs))) return rc;
    if ((rc = reservation_begin(ctx))) return rc;  // the first pass's cursors: zero
    // Pairs: the passes are stable -- decoupled look-back through the one-call sort's status words, which must be clear when the
    // first pass starts (the local sort of a taken sort leaves them so) and are written from here on
    vrs::PoolPayloads pv{};
    if (pairs) {
        if (!ctx->os_status_clean)
            VRS_HIP(ctx, hipMemsetAsync(ctx->os_status, 0, ctx->os_status_rows * VRS_RADIX_SORT_BINS * sizeof(uint32_t), ctx->stream));
        ctx->os_status_clean = false;
        pv.values_home = static_cast<uint32_t *>(st.vptr[st.cur]);
        pv.values_partner = static_cast<uint32_t *>(st.vptr[st.cur ^ 1u]);
        pv.overflow_values = ctx->os_pool_overflow_vals;
        pv.slack_values = ctx->os_pool_slack_vals;
        pv.status = ctx->os_status;
        pv.status_words = ctx->os_status_rows * VRS_RADIX_SORT_BINS;
        pv.spin_budget = ctx->os_spin_budget;
        pv.hold_tile = ctx->os_hold_tile;
        pv.top_bits = top_bits;
        pv.packed = ctx->os_pool_pairs_packed;
    }
    const vrs::PoolPayloads *pvp = pairs ? &pv : nullptr;
    const vrs::PoolStreams ps = vrs::pool_streams(n);
    const uint32_t c = st.cur;
    uint32_t *home = static_cast<uint32_t *>(st.kptr[c]), *partner = static_cast<uint32_t *>(st.kptr[c ^ 1u]);
    vrs::LaunchEvents ev;
    st.cur_at_start = c;
    st.blind_passes = 0;
    if (++ctx->os_stamp == 0) ctx->os_stamp = 1;
    st.stamp = ctx->os_stamp;
    // Everything is enqueued here, before any verdict is known (the workgroups of what a verdict refuses leave at once): the
    // second verdict falls only when the second pass has run, and a host that enqueued the local sort after it would leave the
    // GPU idle for a round trip.  The form's shape -- bits of the second pass, the local sort's workgroup -- is chosen from n alone
    // (pool_shape: uniform keys, buckets of n / 16384 or n / 32768 + a few per cent); a bucket above the local sort's capacity
    // makes the second pass flag the sort.
    const uint32_t tiles_b = vrs::pool_tiles_b_cap(n);
    const uint32_t par = (++ctx->os_pool_epoch) & 1u;
    st.pool_par = par;
    st.pool_reused = ctx->os_pool_reuse && ctx->os_pool_layout_valid && ctx->os_pool_layout_n == n && ctx->os_pool_layout_base == st.key_base &&
                     (ctx->os_pool_layout_sub_bits >> 8) == top_bits;
    // Back-off: a workload whose distribution changes from sort to sort at equal n (sorted, then random; alternating key ranges) finds
    // every kept layout stale -- two passes, a host round trip and the whole sort again, each time.  After two stale layouts in a row the
    // next 16 sorts that could have started in a kept layout sample for themselves; then one tries again.
    if (st.pool_reused && ctx->os_pool_reuse_pause) {
        --ctx->os_pool_reuse_pause;
        st.pool_reused = false;
    }
    if (!st.pool_reused) {
        if ((rc = profile_events(ctx, VRS_KERNEL_POOL_SAMPLE, &ev))) return rc;
        VRS_HIP(ctx, vrs::launch_pool_sample(ctx->stream, home, n, st.key_base, ps, ctx->os_pool_plan, room, par, ev, top_bits));
        ctx->os_pool_layout_valid = false;  // (until this sort is known to have been taken)
    } else {
        ctx->os_pool_layout_reuses++;
    }
    st.ev_lb_before = ctx->events_used[VRS_KERNEL_LOOKBACK_SCATTER];
    st.ev_ls_before = ctx->events_used[VRS_KERNEL_LOCAL_SORT];
    if ((rc = profile_events(ctx, VRS_KERNEL_POOL_PASS_A, &ev))) return rc;
    VRS_HIP(ctx, vrs::launch_pool_pass_a(ctx->stream, home, partner, ctx->os_pool_overflow, n, st.key_base, ps, ctx->os_pool_plan, ctx->os_msd_plan,
                                         ctx->xcc_map, ctx->os_misplace, room, par, ev, pvp, top_bits));
    const bool keep_rooms = st.pool_reused && ctx->os_pool_reuse_rooms && (ctx->os_pool_layout_sub_bits & 255u) == shape.sub_bits;
    VRS_HIP(ctx, vrs::launch_pool_plan(ctx->stream, ctx->os_msd_plan, ctx->os_pool_plan, n, tiles_b, ctx->os_pool_slack_cap, partner, ctx->os_pool_overflow, st.key_base, ps, shape.sub_bits, par,
                                       nullptr, keep_rooms, top_bits));
    if ((rc = profile_events(ctx, VRS_KERNEL_POOL_PASS_B, &ev))) return rc;
    VRS_HIP(ctx, vrs::launch_pool_pass_b(ctx->stream, partner, ctx->os_pool_overflow, ctx->os_pool_slack, n, ctx->os_msd_plan, ctx->os_pool_plan, tiles_b,
                                         st.key_base, vrs::pool_local_capacity(shape.local), ctx->os_pool_slack_cap, ctx->xcc_map, st.stamp, shape.sub_bits, par, ev,
                                         false, pvp, top_bits));
    if ((rc = profile_events(ctx, VRS_KERNEL_LOCAL_SORT, &ev))) return rc;
    VRS_HIP(ctx, vrs::launch_pool_local_sort(ctx->stream, ctx->os_pool_slack, home, n, ctx->os_msd_plan, ctx->os_pool_plan, shape, &ctx->os_plan->head,
                                             ctx->os_host_head_dev, st.stamp, par, ev, top_bytes, nullptr, false, pvp));
    ctx->os_cursors_open = false;  // the local sort re-arms the reservation counters (a refusal is handled by one_read_complete)
    st.active = true;
    return VRS_OK;
}
}  // namespace vrsh

extern "C" {

int vrs_one_call_pool_sorts(vrs_context ctx, uint64_t *pool_sorts, uint64_t *pool_refusals) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    if (pool_sorts) *pool_sorts = ctx->os_pool_sorts;
    if (pool_refusals) *pool_refusals = ctx->os_pool_refusals;
    return VRS_OK;
}

int vrs_pool_form_shape(uint32_t n, uint32_t *sub_bits, uint32_t *bucket_capacity, uint64_t *scratch_bytes) {
    const bool takes = n >= (1u << 22) && n <= vrs::kPoolMaxKeys;
    const vrs::PoolShape shape = takes ? vrs::pool_shape(n) : vrs::PoolShape{0u, 0u};
    if (sub_bits) *sub_bits = shape.sub_bits;
    if (bucket_capacity) *bucket_capacity = takes ? vrs::pool_local_capacity(shape.local) : 0u;
    if (scratch_bytes)
        *scratch_bytes = takes ? (static_cast<uint64_t>(vrs::pool_slack_capacity(n, shape.sub_bits)) + vrs::pool_overflow_capacity(n)) * sizeof(uint32_t) + sizeof(vrs::PoolPlan)
                               : 0u;
    return VRS_OK;
}

int vrs_pool_form_shape_ex(uint32_t n, int pairs, int top_bits_setting, uint32_t *first_pass_bits, uint32_t *second_pass_bits, uint32_t *bucket_capacity,
                           uint64_t *scratch_bytes) {
    if (top_bits_setting == 0) top_bits_setting = 7;  // the library's default cut (vrs_context_t::os_pool_top_bits)
    if (top_bits_setting < 6 || top_bits_setting > 8) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "top_bits: 0 (the default), 6, 7 or 8");
    const bool takes = n >= (1u << 22) && (pairs ? n <= vrs::pool_max_pairs() : n <= vrs::kPoolMaxKeys);
    const vrs::PoolCut cut = takes ? vrs::pool_cut(n, pairs != 0, top_bits_setting, 0) : vrs::PoolCut{0u, 0u, 0u};
    if (first_pass_bits) *first_pass_bits = cut.top_bits;
    if (second_pass_bits) *second_pass_bits = cut.sub_bits;
    if (bucket_capacity) *bucket_capacity = takes ? vrs::pool_local_capacity(cut.local) : 0u;
    if (scratch_bytes) {
        const uint64_t slots = takes ? static_cast<uint64_t>(vrs::pool_slack_capacity(n, cut.sub_bits, 1u << cut.top_bits)) + vrs::pool_overflow_capacity(n) : 0u;
        *scratch_bytes = takes ? slots * sizeof(uint32_t) * (pairs ? 2u : 1u) + sizeof(vrs::PoolPlan) : 0u;  // (pairs: the payloads' twins of both buffers)
    }
    return VRS_OK;
}

int vrs_context_trim_scratch(vrs_context ctx, uint64_t *released_bytes) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    if (const int rc = settle_pending(ctx)) return rc;
    VRS_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (a sort on the stream may still read the buffers)
    const uint64_t bytes = (static_cast<uint64_t>(ctx->os_pool_overflow_cap) + ctx->os_pool_slack_cap + ctx->os_pool_vals_overflow_cap + ctx->os_pool_vals_slack_cap) * sizeof(uint32_t);
    pool_scratch_release(ctx);
    if (released_bytes) *released_bytes = bytes;
    return VRS_OK;
}

int vrs_one_call_pool_no_memory(vrs_context ctx, uint64_t *sorts) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    if (sorts) *sorts = ctx->os_pool_no_memory;
    return VRS_OK;
}

int vrs_one_call_pool_retries(vrs_context ctx, uint64_t *retries) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    if (retries) *retries = ctx->os_pool_retries;
    return VRS_OK;
}

int vrs_one_call_pool_layouts(vrs_context ctx, uint64_t *reused, uint64_t *stale) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    if (reused) *reused = ctx->os_pool_layout_reuses;
    if (stale) *stale = ctx->os_pool_stale_layouts;
    return VRS_OK;
}
}  // extern "C"
