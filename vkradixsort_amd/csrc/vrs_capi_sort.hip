// vrs_capi_sort.hip -- the C ABI, part 3 of 5: the one-call sorts (vrs_sort_*): which form a sort takes (vrs_sort_form.hpp), the
// enqueue-only first half, the settle that finishes what the plan still asks for.  "Two halves": DESIGN.md section 3.
#include "vrs_host.hpp"
#include "vrs_sort_form.hpp"

#include <type_traits>

using namespace vrsh;

namespace vrsh {

// The plan kernel writes the head of the plan straight into pinned host memory and stamps it last; wait for the stamp.
// The plan is at most a counting read behind whatever the stream still has to run: a short spin (the usual case: it is
// there already, or microseconds away), then the thread yields between looks, sleeping a little longer each time, and asks
// the stream now and then so that a faulted queue surfaces as an error instead of an endless wait.  Bounded in time
// (VRS_TUNE_PLAN_WAIT_MS, default 60 s): a stream stuck behind work that never finishes returns VRS_ERROR_TIMEOUT.
int wait_for_host_word(vrs_context ctx, const std::function<bool()> &arrived, bool *never) {
    for (int spins = 0; spins < 20000; ++spins) {  // ~50-100 us
        if (arrived()) return VRS_OK;
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#elif defined(__aarch64__)
        asm volatile("yield" ::: "memory");
#endif
    }
    const auto t0 = std::chrono::steady_clock::now();
    unsigned nap_us = 1;
    for (uint64_t looks = 0;; ++looks) {
        if (arrived()) return VRS_OK;
        if ((looks & 63u) == 63u) {
            const hipError_t q = hipStreamQuery(ctx->stream);
            if (q == hipSuccess) {  // everything enqueued has run: the stamp must be there
                if (arrived()) return VRS_OK;
                if (never) *never = true;
                return fail(ctx, VRS_ERROR_HIP, "the one-call sort's plan never arrived on the host");
            }
            if (q != hipErrorNotReady) return fail_hip(ctx, "hipStreamQuery (waiting for the sort plan)", q);
            const auto waited = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
            if (ctx->os_plan_wait_ms != 0 && waited > static_cast<long long>(ctx->os_plan_wait_ms))
                return fail(ctx, VRS_ERROR_TIMEOUT,
                            "the one-call sort's plan did not arrive in time: the stream is held up by earlier work "
                            "(VRS_TUNE_PLAN_WAIT_MS; the sort itself is still queued -- vrs_sort_settle may be called again)");
        }
        if (nap_us <= 2) sched_yield(); else usleep(nap_us);
        if (nap_us < 200) nap_us *= 2;
    }
}

int wait_for_plan(vrs_context ctx, uint32_t stamp) {
    volatile uint32_t *ready = &ctx->os_host_head->ready;
    return wait_for_host_word(ctx, [&] { return __atomic_load_n(ready, __ATOMIC_ACQUIRE) == stamp; });
}

// ---- the one-call sort for large N (K5 / K5b), in two halves around the plan's arrival on the host.
// Half one (one_read_enqueue) puts one group of four passes on the stream without knowing the plan: the counting read, the
// plan kernel, and the scatter passes as speculative launches that read their streams from the plan in device memory.  Half
// two (one_read_complete), once the plan's head has arrived in pinned host memory, enqueues whatever the plan asks for beyond
// that -- usually nothing for the LSD form, the second MSD pass and the local sort for the hybrid form -- or the passes the
// plan marked abnormal (identity: left out; unbalanced streams: a contract pass; wide streams: launched again).
// vrs_sort_* run both halves (the host waits for the plan's head -- for the counting read, never for the sort -- while the
// first pass runs).  With VRS_TUNE_ASYNC_SORT = 1 they run only the first and return at once, whatever the stream still
// has queued; the second half runs in vrs_sort_settle (also called by every entry point that waits for the stream or
// starts another sort).  In that mode a sort the hybrid form may take is enqueued COMPLETELY -- second MSD pass and local
// sort included, with grids sized for the worst plan the form accepts; their workgroups leave at once should the plan
// refuse -- so that the usual case needs no second half at all.
vrs_buffer_t stack_view(vrs_context ctx, void *ptr, size_t bytes) {
    vrs_buffer_t b;
    b.ctx = ctx;
    b.device = ctx->device;
    b.ptr = ptr;
    b.size = bytes;
    b.owned = false;
    return b;
}

// everything here is a function of (n, key type, payload or not, the form) alone: both halves compute the same
OneReadGeometry one_read_geometry(vrs_context ctx, const vrs_context_t::OneRead &st) {
    constexpr uint32_t S = vrs::kStreams;
    OneReadGeometry g{};
    const uint32_t n = st.n;
    const bool wide = st.key_bytes == 8, pairs = st.vptr[0] != nullptr;
    // groups per pass: 32 let the streams follow skewed data more closely, but every workgroup of the counting read
    // flushes 3 * G * 256 counters -- a fixed cost that only large inputs amortise (10^7 keys: 20 vs 34 us for the
    // counting read, 3 * 10^7: 47 vs 61, 10^8: a tie; profiles/labs/r02_groups_and_fused_plan.txt); the hybrid form's
    // bucket histogram needs the 8-group tables to fit beside it in LDS
    g.G = st.msd_capable ? 8u : ctx->os_groups ? ctx->os_groups : (n < (1u << 26) ? 8u : 32u);
    g.T = vrs::onesweep_tile_keys(st.key_bytes);
    g.tiles_total = (n + g.T - 1) / g.T;
    const uint32_t group_tiles = (g.tiles_total + g.G - 1) / g.G;  // tiles per pass-0 group (slice of the input)
    g.group_len = group_tiles * g.T;                               // < 2^30 / 8 + 8192
    // pass 0's streams are neighbouring slices merged (the plan kernel gets the same cuts)
    g.cuts0 = vrs::pass0_stream_cuts(n, g.group_len, g.G);
    g.tiles0 = 0;  // tiles of the longest of them
    for (uint32_t k = 0; k < S; ++k) {
        const uint64_t a = std::min<uint64_t>(static_cast<uint64_t>(g.cuts0.first_group[k]) * g.group_len, n);
        const uint64_t b = std::min<uint64_t>(static_cast<uint64_t>(g.cuts0.first_group[k + 1]) * g.group_len, n);
        g.tiles0 = std::max<uint32_t>(g.tiles0, static_cast<uint32_t>((b - a + g.T - 1) / g.T));
    }
    const uint32_t even = (g.tiles_total + S - 1) / S;            // tiles of a perfectly even stream
    g.tile_cap = std::max(g.tiles0, even + even / 4 + 2);         // later passes: streams up to 25 % longer
    // Passes 1-3 are enqueued before the plan is known: their grids have room for streams a little longer than even ones
    // (uniform keys: the longest stream is within a tile or two of N / 8).  Surplus workgroups are not free (3 000 of
    // them cost 3-4 us per pass, profiles/labs/r02_blind_grid.txt), so the slack is small; a pass whose longest stream
    // needs more -- but no more than tile_cap -- leaves at once and is launched again with its exact grid.
    g.blind_cap = std::min(g.tile_cap, even + even / 64 + 2);
    // second MSD pass: every XCD walks 32 top-byte buckets, each rounded up to whole tiles.  Launched once the plan is known
    // it may be up to 25 % over the even share; launched blind (async mode) the grid IS the cap, so the slack is 6 %
    g.tiles_b_cap = st.blind_tail ? even + even / 16 + 40 : even + even / 4 + 40;
    if (st.pass_b_groups) {
        // a number of groups that is no multiple of 8 leaves some XCDs one group more than others: room for the fullest
        const uint32_t per_xcd = (st.pass_b_groups + 7u) / 8u;
        const uint32_t group_tiles_b = (g.tiles_total + st.pass_b_groups - 1) / st.pass_b_groups + 1;
        g.tiles_b_cap = std::max(g.tiles_b_cap, per_xcd * (group_tiles_b + group_tiles_b / 16u) + 40u);
    }
    // the local sort's capacity per bucket; launched blind, the workgroup shape of bare uint32 keys is chosen from N alone
    // (uniform keys: buckets of N / 16384 +- a few per cent)
    g.local_cap = wide && pairs ? vrs::msd_local_capacity_pairs_u64(false) : vrs::msd_local_capacity(pairs || wide);
    if (st.blind_tail) {
        // uniform keys: N / 16384 + a few per cent -- unless the caller knows better (a sub-range of a larger sort: vrs_msd_finish_u32)
        // (the fullest of 16384 buckets of uniform keys lies 4-4.5 deviations above the mean; one that does not fit after all is a
        // refusal, not an error)
        const double mean = static_cast<double>(n) / vrs::kMsdBucketCount;
        const uint64_t expect = st.bucket_hint ? st.bucket_hint : static_cast<uint64_t>(mean + 5.5 * std::sqrt(mean)) + 32u;
        if (pairs && wide) {
            if (expect <= vrs::msd_local_capacity_pairs_u64(true)) g.local_cap = vrs::msd_local_capacity_pairs_u64(true);
        } else if (pairs || wide) {
            if (expect <= vrs::msd_local_capacity_pairs_small()) g.local_cap = vrs::msd_local_capacity_pairs_small();
        } else if (expect <= vrs::msd_local_capacity_wave()) {
            g.local_cap = vrs::msd_local_capacity_wave();
        } else if (expect <= vrs::msd_local_capacity_small()) {
            g.local_cap = vrs::msd_local_capacity_small();
        }
    }
    g.rows = static_cast<size_t>(S) * std::max(g.tile_cap, st.msd_capable ? g.tiles_b_cap : 0u);  // status rows: one region for all passes (tagged words)
    if (st.pool && pairs)  // the stable pool form's two passes: a row per tile of the slices' lists / of the XCDs' lists
        g.rows = std::max(g.rows, static_cast<size_t>(8) * std::max(vrs::pool_streams(n).tiles_per_stream, vrs::pool_tiles_b_cap(n)));
    return g;
}

int one_read_scratch(vrs_context ctx, const vrs_context_t::OneRead &st, const OneReadGeometry &g) {
    if (!ctx->os_tables) {
        uint32_t *tables = nullptr;
        vrs::OnesweepPlan *plan = nullptr;
        vrs::OnesweepPlanHead *host = nullptr, *host_dev = nullptr;
        // one allocation: the digit tables and, behind them, the ticket word of the fused plan
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&tables), (vrs::kDigitTableWords + 64) * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemsetAsync(tables, 0, (vrs::kDigitTableWords + 64) * sizeof(uint32_t), ctx->stream);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&plan), sizeof(vrs::OnesweepPlan));
        if (e == hipSuccess)
            e = hipHostMalloc(reinterpret_cast<void **>(&host), sizeof(vrs::OnesweepPlanHead) + vrs::kMsdLogWords * sizeof(uint32_t),
                              hipHostMallocMapped | hipHostMallocCoherent);  // behind the head: the log of vrs_msd_finish_u32's decisions
        if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void **>(&host_dev), host, 0);
        if (e != hipSuccess) {  // all or nothing: a half-made set would be dereferenced by the next call
            if (host) (void)hipHostFree(host);
            if (plan) (void)hipFree(plan);
            if (tables) (void)hipFree(tables);
            return fail_hip(ctx, "one-call sort scratch allocation", e);
        }
        std::memset(host, 0, sizeof *host + vrs::kMsdLogWords * sizeof(uint32_t));
        ctx->os_tables = tables;
        ctx->os_ticket = tables + vrs::kDigitTableWords;
        ctx->os_plan = plan;
        ctx->os_host_head = host;
        ctx->os_host_head_dev = host_dev;
    }
    if (st.msd_capable && !ctx->os_msd_counts) {
        uint32_t *counts = nullptr;
        vrs::MsdPlan *mp = nullptr;
        vrs::OnesweepPlan *pa = nullptr;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&counts), vrs::kMsdCountWords * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemsetAsync(counts, 0, vrs::kMsdCountWords * sizeof(uint32_t), ctx->stream);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&mp), sizeof(vrs::MsdPlan));
        if (e == hipSuccess) e = hipMemsetAsync(mp, 0, sizeof(vrs::MsdPlan), ctx->stream);  // the reservation counters start at zero
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&pa), sizeof(vrs::OnesweepPlan));
        if (e != hipSuccess) {
            if (pa) (void)hipFree(pa);
            if (mp) (void)hipFree(mp);
            if (counts) (void)hipFree(counts);
            return fail_hip(ctx, "hybrid sort scratch allocation", e);
        }
        ctx->os_msd_counts = counts;
        ctx->os_msd_plan = mp;
        ctx->os_plan_a = pa;
    }
    if (g.rows > ctx->os_status_rows) {
        if (ctx->os_status) {
            VRS_HIP(ctx, hipStreamSynchronize(ctx->stream));
            VRS_HIP(ctx, hipFree(ctx->os_status));
            ctx->os_status = nullptr;
            ctx->os_status_rows = 0;
        }
        VRS_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->os_status), g.rows * VRS_RADIX_SORT_BINS * sizeof(uint32_t)));
        ctx->os_status_rows = g.rows;
        ctx->os_status_clean = false;
    }
    return VRS_OK;
}

int one_read_lookback_pass(vrs_context ctx, vrs_context_t::OneRead &st, uint32_t i, uint32_t shift, uint32_t grid_tiles, bool forced) {
    const bool pairs = st.vptr[0] != nullptr;
    void *kin = st.kptr[st.cur], *kout = st.kptr[st.cur ^ 1u];
    void *vin = pairs ? st.vptr[st.cur] : nullptr, *vout = pairs ? st.vptr[st.cur ^ 1u] : nullptr;
    st.cur ^= 1u;
    vrs::LaunchEvents ev;
    int r = profile_events(ctx, VRS_KERNEL_LOOKBACK_SCATTER, &ev);
    if (r) return r;
    VRS_HIP(ctx, vrs::launch_onesweep_scatter(ctx->stream, kin, kout, static_cast<const uint32_t *>(vin), static_cast<uint32_t *>(vout),
                                              ctx->os_plan, i, shift, ctx->os_status, grid_tiles, forced, ctx->scatter.atomic_rank,
                                              ctx->xcc_map, st.key_bytes, ctx->os_spin_budget, ctx->os_hold_tile, ev, ctx->os_misplace, 0, nullptr, drift_word(ctx)));
    return VRS_OK;
}

// Reservation counters (MsdPlan::cursor_* / back_*): zero when a reserving pass starts; the local sort leaves them so.  Every
// entry point that is about to enqueue a reserving pass calls this first.
bool reserves(vrs_context ctx, uint32_t n, bool pairs) {
    (void)n;  // (measured from 1.5e7 to 1e8 keys: 2 to 5 % of the sort at every size the hybrid form takes)
    return !pairs && ctx->os_reserve != 0;
}
int reservation_begin(vrs_context ctx) {
    if (!ctx->os_reserve || !ctx->os_msd_plan) return VRS_OK;
    if (ctx->os_cursors_open)
        VRS_HIP(ctx, hipMemsetAsync(reinterpret_cast<char *>(ctx->os_msd_plan) + offsetof(vrs::MsdPlan, cursor_a), 0, vrs::kMsdCursorBytes, ctx->stream));
    ctx->os_cursors_open = true;
    return VRS_OK;
}

// second MSD pass + local sort of the hybrid form: partner -> home, then the buckets in place
int one_read_hybrid_tail(vrs_context ctx, vrs_context_t::OneRead &st, const OneReadGeometry &g, uint32_t tiles_b, uint32_t max_bucket,
                                bool status_was_clean) {
    const bool pairs = st.vptr[0] != nullptr, wide = st.key_bytes == 8;
    const uint32_t home = st.cur_at_start;
    vrs::LaunchEvents ev;
    int rc;
    if ((rc = profile_events(ctx, VRS_KERNEL_LOOKBACK_SCATTER, &ev))) return rc;
    VRS_HIP(ctx, vrs::launch_msd_pass_b(ctx->stream, st.kptr[home ^ 1u], st.kptr[home],
                                        pairs ? static_cast<const uint32_t *>(st.vptr[home ^ 1u]) : nullptr,
                                        pairs ? static_cast<uint32_t *>(st.vptr[home]) : nullptr, ctx->os_msd_plan, ctx->os_status,
                                        tiles_b, ctx->scatter.atomic_rank, ctx->xcc_map, st.key_bytes, ctx->os_spin_budget, ev,
                                        st.key_base, st.sub_bits, reserves(ctx, st.n, pairs), drift_word(ctx)));
    if ((rc = profile_events(ctx, VRS_KERNEL_LOCAL_SORT, &ev))) return rc;
    // Launched with the plan known (it said yes), the local sort also clears the look-back status words -- it is LDS-bound and
    // has HBM time to spare, the next sort's counting read does not.  Launched blind it may leave at once: nothing is promised.
    // (Blind, but with every status word zero before the second pass -- vrs_msd_finish_u32 -- the promise holds again: either both
    // kernels run, and the local sort clears what the pass wrote, or both leave at once.)
    // Bare keys with reservation: neither MSD pass has touched the status words -- they are as clear as the counting read (or the
    // caller's memset) left them, and the local sort has nothing to do about them.
    const bool untouched = reserves(ctx, st.n, pairs);
    const bool clears = !untouched && (!st.blind_tail || status_was_clean);
    uint32_t *clear = clears ? ctx->os_status : nullptr;
    const size_t clear_words = clears ? ctx->os_status_rows * VRS_RADIX_SORT_BINS : 0;
    if (wide)
        VRS_HIP(ctx, vrs::launch_msd_local_sort_u64(ctx->stream, st.kptr[home], ctx->os_msd_plan, max_bucket, ev, clear, clear_words,
                                                    pairs ? static_cast<uint32_t *>(st.vptr[home]) : nullptr));
    else
        VRS_HIP(ctx, vrs::launch_msd_local_sort(ctx->stream, static_cast<uint32_t *>(st.kptr[home]),
                                                pairs ? static_cast<uint32_t *>(st.vptr[home]) : nullptr, ctx->os_msd_plan, max_bucket, ev,
                                                clear, clear_words));
    // (a whole sort enqueued blind may still be refused and run its LSD passes, which write the words, from one_read_complete:
    // it makes no claim)
    if (clear || (untouched && (!st.blind_tail || status_was_clean))) ctx->os_status_clean = true;
    ctx->os_cursors_open = false;  // the local sort is on the stream: it re-arms the reservation counters (or, the plan refusing, nothing touched them)
    (void)g;
    return VRS_OK;
}

// what the dispatcher's decision function looks at, read off the context
vrs::SortKnobs sort_knobs(vrs_context ctx) {
    vrs::SortKnobs k;
    k.single_max_keys = ctx->single_max_keys;
    k.one_call_min_keys = ctx->one_call_min_keys;
    k.hybrid_min_keys = ctx->os_hybrid_min_keys;
    k.pool_min_keys = ctx->os_pool_min_keys;
    k.hybrid = ctx->os_hybrid ? 1 : 0;
    k.pool = ctx->os_pool;
    k.pool_pairs = ctx->os_pool_pairs;
    k.reserve = ctx->os_reserve;
    k.groups = ctx->os_groups;
    k.xcc_map_valid = ctx->xcc_map_valid;
    k.atomic_rank = ctx->atomic_rank_verified && ctx->scatter.atomic_rank;
    k.pool_skip = ctx->os_pool_skip;
    k.pool_skip_n = ctx->os_pool_skip_n;
    k.wide_refused = ctx->os_wide_refused;
    k.wide_skipped = ctx->os_wide_skipped;
    return k;
}

int one_read_enqueue(vrs_context ctx) {
    vrs_context_t::OneRead &st = ctx->one_read;
    const uint32_t n = st.n;
    const int key_bytes = st.key_bytes;
    const bool wide = key_bytes == 8, pairs = st.vptr[0] != nullptr;
    if (st.group == 0) {
        // which form: vrs_sort_form.hpp (one pure function of the size, the kind of sort, the context's settings and what it remembers --
        // the same function answers vrs_sort_form_for, through which the CPU tests walk the decision table)
        vrs::SortKnobs knobs = sort_knobs(ctx);
        knobs.no_pool = st.no_pool;
        knobs.no_hybrid = st.no_hybrid;
        const vrs::SortDecision d = vrs::sort_form_for(n, key_bytes, pairs, knobs);
        ctx->os_wide_skipped = d.wide_skipped;
        ctx->os_pool_skip = d.pool_skip;
        st.msd_capable = d.msd_capable;
        st.pool = d.pool;
        // enqueued completely (enqueue-only calls): like the fast count it implies, only while the context's last hybrid-capable
        // sort of this kind took the form (or always: VRS_TUNE_HYBRID_FAST_COUNT = 2) -- a refusal of a blind tail costs a second
        // counting read, and data that was refused once is usually refused again
        st.blind_tail = st.msd_capable && st.deferred &&
                        (wide || ctx->os_fast_count == 2 || (ctx->os_fast_count == 1 && ctx->os_fast_count_armed[pairs ? 1 : 0]));
        // Fast count: the counting read of a hybrid-capable sort fills only the bucket histogram (1 LDS add per key instead
        // of 5).  If the plan then refuses the hybrid form, nothing has been moved and the sort starts over as an LSD sort
        // -- a second counting read.  Adaptive (default): fast only while the context's last hybrid-capable sort took the
        // hybrid form; after a refusal the next ones count everything again (a refusal then costs nothing extra) until one
        // is taken.  A sort enqueued completely (async mode) always counts fast: a refusal must find every key in place.
        st.fast_count = st.msd_capable && (wide || st.blind_tail || ctx->os_fast_count == 2 ||
                                           (ctx->os_fast_count == 1 && ctx->os_fast_count_armed[pairs ? 1 : 0]));
    }
    const bool msd = st.msd_capable && st.group == 0;
    const OneReadGeometry g = one_read_geometry(ctx, st);
    int rc = one_read_scratch(ctx, st, g);
    if (rc) return rc;
    if (msd && st.pool) {
        rc = one_read_enqueue_pool(ctx, g);
        if (rc != kPoolNoMemory) return rc;
        // no room on the device for the form's scratch: nothing was enqueued -- the same sort in a form that needs none (and the next
        // sorts of this size do not ask again at once: the adaptive skip, as after a refusal)
        st.no_pool = true;
        st.pool = false;
        if (ctx->os_pool == 1) {
            ctx->os_pool_skip = 15;
            ctx->os_pool_skip_n = n;
        }
        return one_read_enqueue(ctx);
    }
    vrs::LaunchEvents ev;
    // the digit tables must be all zero when a counting read starts; plan_kernel leaves them so.  Should anything fail
    // between the two launches, re-arm them for the next sort.
    struct TablesGuard {
        vrs_context ctx;
        bool armed = false;
        ~TablesGuard() {
            if (armed) (void)hipMemsetAsync(ctx->os_tables, 0, (vrs::kDigitTableWords + 64) * sizeof(uint32_t), ctx->stream);
            if (armed && ctx->os_msd_counts)
                (void)hipMemsetAsync(ctx->os_msd_counts, 0, vrs::kMsdCountWords * sizeof(uint32_t), ctx->stream);
        }
    } guard{ctx};
    const uint32_t group = st.group;
    // the previous hybrid sort's local sort left the status words cleared (see one_read_hybrid_tail): nothing to zero then
    // ("clean" speaks for the whole allocation -- a sort whose MSD passes reserve leaves the words alone and hands the claim on --
    // so a counting read that has to clear them clears all of them, not just the rows of this sort)
    const size_t zero_words = ctx->os_status_clean ? 0 : ctx->os_status_rows * VRS_RADIX_SORT_BINS;
    ctx->os_status_clean = false;  // this sort's passes write them
    if ((rc = profile_events(ctx, VRS_KERNEL_DIGIT_TABLES, &ev))) return rc;
    if (++ctx->os_stamp == 0) ctx->os_stamp = 1;
    st.stamp = ctx->os_stamp;
    guard.armed = true;
    const vrs::FusedPlan fused{ctx->os_plan, ctx->os_host_head_dev, ctx->os_ticket, st.stamp, g.T, g.tile_cap, g.blind_cap, g.cuts0};
    if (msd) {
        // hybrid: the same read (after probing the key range on a sample) also fills the histogram of the range's top 14
        // bits; ONE plan kernel makes the LSD plan as always, decides which form runs, arms exactly one of the two first
        // passes and stamps the head
        if (wide)
            VRS_HIP(ctx, vrs::launch_msd_count_u64(ctx->stream, st.kptr[st.cur], n, g.group_len, ctx->os_status,
                                                   zero_words, ctx->scatter.compute_units, ctx->os_msd_counts, ev));
        else
            VRS_HIP(ctx, vrs::launch_digit_tables_msd(ctx->stream, st.kptr[st.cur], n, g.group_len, ctx->os_tables, ctx->os_status,
                                                      zero_words, ctx->scatter.compute_units, ctx->os_msd_counts,
                                                      st.fast_count, ev, st.key_base));
        VRS_HIP(ctx, vrs::launch_msd_plan(ctx->stream, ctx->os_msd_counts, ctx->os_msd_plan, ctx->os_plan_a, ctx->os_plan,
                                          ctx->os_host_head_dev, st.stamp, n, g.T, g.tiles_b_cap, g.local_cap, ctx->os_tables,
                                          g.group_len, g.tile_cap, g.blind_cap, g.cuts0, wide ? 2u : st.fast_count ? 1u : 0u,
                                          wide ? 50u : 18u));
    } else {
        VRS_HIP(ctx, vrs::launch_digit_tables(ctx->stream, st.kptr[st.cur], n, key_bytes, 32u * group, g.group_len, g.G, ctx->os_tables,
                                              ctx->os_status, zero_words, ctx->scatter.compute_units, ev,
                                              ctx->os_fused_plan ? &fused : nullptr));
        if (!ctx->os_fused_plan)
            VRS_HIP(ctx, vrs::launch_plan(ctx->stream, ctx->os_tables, ctx->os_plan, ctx->os_host_head_dev, st.stamp, n, g.group_len,
                                          g.G, g.T, g.tile_cap, g.blind_cap, g.cuts0));
    }
    guard.armed = false;
    // speculative launches, before the plan is known here.  LSD form: all four passes (pass 0's streams are the host's
    // own cuts).  Hybrid-capable sort: the two candidate FIRST passes -- the first MSD pass and the LSD pass 0 (same
    // buffers; the plan arms exactly one, the other leaves at once; after a fast count the LSD pass 0 is not enqueued at
    // all) -- and, in async mode, the rest of the hybrid form as well.
    st.cur_at_start = st.cur;
    st.ev_lb_before = ctx->events_used[VRS_KERNEL_LOOKBACK_SCATTER];
    st.ev_ls_before = ctx->events_used[VRS_KERNEL_LOCAL_SORT];
    st.blind_passes = msd ? (st.fast_count ? 0u : 1u) : 4u;
    if (msd) {  // the first MSD pass goes first: it is the one that usually runs, the other then leaves behind it
        if (!pairs && (rc = reservation_begin(ctx))) return rc;
        if ((rc = profile_events(ctx, VRS_KERNEL_LOOKBACK_SCATTER, &ev))) return rc;
        const uint32_t c = st.cur_at_start;
        VRS_HIP(ctx, vrs::launch_onesweep_scatter(ctx->stream, st.kptr[c], st.kptr[c ^ 1u],
                                                  pairs ? static_cast<const uint32_t *>(st.vptr[c]) : nullptr,
                                                  pairs ? static_cast<uint32_t *>(st.vptr[c ^ 1u]) : nullptr, ctx->os_plan_a, 0,
                                                  vrs::kShiftFromPlan, ctx->os_status, g.tiles0, false, ctx->scatter.atomic_rank,
                                                  ctx->xcc_map, key_bytes, ctx->os_spin_budget, ctx->os_hold_tile, ev, ctx->os_misplace,
                                                  st.key_base, reserves(ctx, n, pairs) ? ctx->os_msd_plan : nullptr, drift_word(ctx)));
    }
    for (uint32_t i = 0; i < st.blind_passes; ++i)
        if ((rc = one_read_lookback_pass(ctx, st, i, 32u * group + 8u * i, i == 0 ? g.tiles0 : g.blind_cap, false))) return rc;
    if (msd && st.blind_tail && (rc = one_read_hybrid_tail(ctx, st, g, g.tiles_b_cap, g.local_cap))) return rc;
    st.active = true;
    return VRS_OK;
}

// the plan's head has arrived: finish the group; *done = the whole sort is on the stream
int one_read_complete(vrs_context ctx, bool *done) {
    vrs_context_t::OneRead &st = ctx->one_read;
    *done = false;
    const uint32_t n = st.n;
    const int key_bytes = st.key_bytes;
    const bool wide = key_bytes == 8, pairs = st.vptr[0] != nullptr;
    const bool msd = st.msd_capable && st.group == 0;
    const OneReadGeometry g = one_read_geometry(ctx, st);
    const vrs::OnesweepPlanHead &head = *ctx->os_host_head;
    const bool timed = (ctx->profile_mask & (1u << VRS_KERNEL_LOOKBACK_SCATTER)) != 0;
    const bool timed_ls = (ctx->profile_mask & (1u << VRS_KERNEL_LOCAL_SORT)) != 0;
    int rc;
    const auto finish = [&]() -> int {
        if (st.cur) {  // an odd number of passes ran
            VRS_HIP(ctx, hipMemcpyAsync(st.kptr[0], st.kptr[1], static_cast<size_t>(n) * key_bytes, hipMemcpyDeviceToDevice, ctx->stream));
            if (pairs)
                VRS_HIP(ctx, hipMemcpyAsync(st.vptr[0], st.vptr[1], static_cast<size_t>(n) * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
        }
        st.active = false;
        *done = true;
        return VRS_OK;
    };
    if (msd && st.pool) {
        if (head.msd_ok) {  // both verdicts said yes: the whole form is on the stream, the result lands in the caller's buffer
            st.cur = st.cur_at_start;
            ctx->os_hybrid_sorts++;
            ctx->os_pool_sorts++;
            if (pairs) {
                ctx->os_pool_pair_sorts++;
                ctx->os_status_clean = true;  // (the local sort cleared the look-back words behind the two passes)
            }
            if (st.pool_reused) ctx->os_pool_stale_run = 0;  // (a kept layout fitted)
            ctx->os_pool_layout_valid = true;  // its regions held: the next sort of this size may start in them
            ctx->os_pool_layout_n = n;
            ctx->os_pool_layout_base = st.key_base;
            ctx->os_pool_layout_sub_bits = st.pool_sub_bits | (st.pool_top_bits << 8);
            return finish();
        }
        if (head.msd_max_bucket != 0u && !st.pool_retried) {
            // Not refused, only misjudged: a bucket has more keys than the local sort that was enqueued blind takes (its shape came from
            // n alone; skewed keys).  Every bucket lies whole in its region: a local sort of a larger shape finishes the sort.
            uint32_t local = 99u;
            for (uint32_t cand : {0u, 1u, 2u, 5u})
                if (local == 99u && (cand == 5u) == pairs && head.msd_max_bucket <= vrs::pool_local_capacity(cand)) local = cand;
            if (local != 99u) {
                vrs::LaunchEvents ev;
                if ((rc = profile_events(ctx, VRS_KERNEL_LOCAL_SORT, &ev))) return rc;
                if (++ctx->os_stamp == 0) ctx->os_stamp = 1;
                st.stamp = ctx->os_stamp;
                vrs::PoolPayloads pv{};
                if (pairs) {
                    pv.values_home = static_cast<uint32_t *>(st.vptr[st.cur_at_start]);
                    pv.slack_values = ctx->os_pool_slack_vals;
                    pv.status = ctx->os_status;
                    pv.status_words = ctx->os_status_rows * VRS_RADIX_SORT_BINS;
                    pv.top_bits = st.pool_top_bits;
                    pv.packed = ctx->os_pool_pairs_packed;
                }
                st.pool_retried = true;
                st.pool_local = local;
                ctx->os_pool_retries++;
                VRS_HIP(ctx, vrs::launch_pool_local_sort(ctx->stream, ctx->os_pool_slack, static_cast<uint32_t *>(st.kptr[st.cur_at_start]), n, ctx->os_msd_plan,
                                                         ctx->os_pool_plan, vrs::PoolShape{st.pool_sub_bits, local}, &ctx->os_plan->head, ctx->os_host_head_dev,
                                                         st.stamp, st.pool_par, ev, 1u << st.pool_top_bits, nullptr, true, pairs ? &pv : nullptr));
                return VRS_OK;  // (still active: the settle waits for this one's word)
            }
        }
        // Refused: no key of the caller's buffer has moved (the passes wrote the partner and the context's scratch only; the
        // local sort left at once).  Hand the events of what left at once back (what ran stays on the books: the two
        // passes).  The reservation counters hold what the first pass reserved and no local sort re-armed them.
        if (timed_ls) ctx->events_used[VRS_KERNEL_LOCAL_SORT] = st.ev_ls_before;
        ctx->os_cursors_open = true;
        ctx->os_pool_layout_valid = false;
        if (st.pool_reused) {
            // the KEPT layout did not fit these keys (another distribution, another key range): no verdict on the form -- the same
            // sort again, sampled this time
            ctx->os_pool_stale_layouts++;
            if (++ctx->os_pool_stale_run >= 2u) {
                ctx->os_pool_stale_run = 0;
                ctx->os_pool_reuse_pause = 16;
            }
            st.group = 0;
            st.cur = st.cur_at_start;
            return one_read_enqueue(ctx);
        }
        ctx->os_pool_refusals++;
        if (ctx->os_pool == 1) {
            ctx->os_pool_skip = 15;
            ctx->os_pool_skip_n = n;
        }
        st.no_pool = true;
        st.pool = false;
        st.group = 0;
        st.cur = st.cur_at_start;
        return one_read_enqueue(ctx);
    }
    if (msd && !head.msd_ok && head.lsd_missing) {
        // fast count, and the plan refused the hybrid form: every speculative launch left at once, no key has moved.
        // Start over as an LSD sort (its own counting read).
        if (timed) ctx->events_used[VRS_KERNEL_LOOKBACK_SCATTER] = st.ev_lb_before;
        if (timed_ls) ctx->events_used[VRS_KERNEL_LOCAL_SORT] = st.ev_ls_before;
        if (wide) ctx->os_wide_refused = true; else ctx->os_fast_count_armed[pairs ? 1 : 0] = false;
        ctx->os_hybrid_recounts++;
        st.no_hybrid = true;
        st.group = 0;
        st.cur = st.cur_at_start;
        return one_read_enqueue(ctx);
    }
    if (msd && !wide) ctx->os_fast_count_armed[pairs ? 1 : 0] = head.msd_ok != 0u;
    if (msd && wide && head.msd_ok) ctx->os_wide_refused = false;
    if (msd && head.msd_ok) {
        // hybrid form: the first MSD pass is running (keys -> partner); second pass back, then the buckets in place
        if (timed) {  // the LSD pass 0 (if it was enqueued) left at once: hand its events back
            if (st.blind_passes && st.blind_tail)
                std::swap(ctx->events[VRS_KERNEL_LOOKBACK_SCATTER][st.ev_lb_before + 1], ctx->events[VRS_KERNEL_LOOKBACK_SCATTER][st.ev_lb_before + 2]);
            ctx->events_used[VRS_KERNEL_LOOKBACK_SCATTER] = st.ev_lb_before + (st.blind_tail ? 2 : 1);
        }
        if (!st.blind_tail && (rc = one_read_hybrid_tail(ctx, st, g, head.msd_tiles_b, head.msd_max_bucket))) return rc;
        // enqueued blind and taken: reserving passes have left the status words as the counting read cleared them
        if (st.blind_tail && reserves(ctx, n, pairs)) ctx->os_status_clean = true;
        st.cur = st.cur_at_start;
        ctx->os_hybrid_sorts++;
        return finish();  // the whole key is sorted (64-bit keys: no second group of passes)
    }
    if (msd) {  // refused, but the LSD plan exists: the first MSD pass (and a blind tail) left at once -- hand the events back, keep the LSD pass 0's
        if (timed) {
            if (st.blind_passes)
                std::swap(ctx->events[VRS_KERNEL_LOOKBACK_SCATTER][st.ev_lb_before], ctx->events[VRS_KERNEL_LOOKBACK_SCATTER][st.ev_lb_before + 1]);
            ctx->events_used[VRS_KERNEL_LOOKBACK_SCATTER] = st.ev_lb_before + st.blind_passes;
        }
        if (timed_ls) ctx->events_used[VRS_KERNEL_LOCAL_SORT] = st.ev_ls_before;
    }
    const uint32_t q = std::min<uint32_t>(head.first_abnormal, st.blind_passes);
    ctx->os_lookback_passes += q;
    if (q < 4) {
        // passes q..3 left at once on the device: take back their (untouched) buffers and timing events, enqueue them again
        st.cur = st.cur_at_start ^ (q & 1u);
        if (timed) ctx->events_used[VRS_KERNEL_LOOKBACK_SCATTER] = st.ev_lb_before + q;
        // the contract pass a group may fall back to walks launch tiles of 32 (uint32) / 16 (uint64) blocks
        const uint32_t B = launch_tile_blocks(key_bytes);
        vrs_push_constants pc{n, 0, vrs_workgroup_count(n, B), B};
        for (uint32_t i = q; i < 4; ++i) {
            const uint32_t shift = 32u * st.group + 8u * i;
            if (head.mode[i] == vrs::kPassIdentity) {
                ctx->os_skipped_passes++;
            } else if (head.mode[i] == vrs::kPassUnbalanced) {
                vrs_buffer_t kin = stack_view(ctx, st.kptr[st.cur], static_cast<size_t>(n) * key_bytes);
                vrs_buffer_t kout = stack_view(ctx, st.kptr[st.cur ^ 1u], static_cast<size_t>(n) * key_bytes);
                vrs_buffer_t vin = stack_view(ctx, pairs ? st.vptr[st.cur] : nullptr, static_cast<size_t>(n) * sizeof(uint32_t));
                vrs_buffer_t vout = stack_view(ctx, pairs ? st.vptr[st.cur ^ 1u] : nullptr, static_cast<size_t>(n) * sizeof(uint32_t));
                st.cur ^= 1u;
                ctx->os_fallback_passes++;
                if ((rc = ensure_sort_hist(ctx, pc.g_num_workgroups))) return rc;
                if ((rc = contract_pass(ctx, &kin, &kout, pairs ? &vin : nullptr, pairs ? &vout : nullptr, &pc, shift, key_bytes))) return rc;
            } else {
                ctx->os_lookback_passes++;
                if (i < st.blind_passes) ctx->os_relaunched_passes++;
                if ((rc = one_read_lookback_pass(ctx, st, i, shift, head.max_tiles[i], true))) return rc;
            }
        }
    }
    // (four look-back passes: the data is back where the group started)
    if (++st.group < static_cast<uint32_t>(key_bytes) / 4u) return one_read_enqueue(ctx);
    return finish();
}

// second half of a pending one-call sort (no-op without one); blocks until the plan(s) arrived and everything is enqueued
int one_read_settle(vrs_context ctx) {
    struct Settling {  // the second half itself goes through entry points that would settle
        vrs_context ctx;
        explicit Settling(vrs_context c) : ctx(c) { ctx->one_read_settling = true; }
        ~Settling() { ctx->one_read_settling = false; }
    } settling(ctx);
    while (ctx->one_read.active) {
        int rc = wait_for_plan(ctx, ctx->one_read.stamp);
        if (rc == VRS_ERROR_TIMEOUT) return rc;  // still pending: a later settle may succeed
        if (rc) {  // the plan never arrived / the stream faulted: nothing to resume, and the next sort must not find this one "pending"
            ctx->one_read.active = false;
            if (ctx->os_tables) (void)hipMemsetAsync(ctx->os_tables, 0, (vrs::kDigitTableWords + 64) * sizeof(uint32_t), ctx->stream);
            if (ctx->os_msd_counts) (void)hipMemsetAsync(ctx->os_msd_counts, 0, vrs::kMsdCountWords * sizeof(uint32_t), ctx->stream);
            ctx->os_cursors_open = true;
            return rc;
        }
        bool done = false;
        if ((rc = one_read_complete(ctx, &done))) {
            ctx->one_read.active = false;  // the sort failed half-way: nothing to resume
            return rc;
        }
    }
    return VRS_OK;
}
int settle_pending(vrs_context ctx) { return ctx->one_read.active && !ctx->one_read_settling ? one_read_settle(ctx) : VRS_OK; }

int sort_one_read(vrs_context ctx, vrs_buffer keys, vrs_buffer keys_tmp, vrs_buffer values, vrs_buffer values_tmp,
                         uint32_t n, int key_bytes, uint32_t key_base) {
    vrs_context_t::OneRead &st = ctx->one_read;
    st = vrs_context_t::OneRead{};
    st.key_base = key_bytes == 4 ? key_base & 0xFF000000u : 0u;
    st.kptr[0] = keys->ptr;
    st.kptr[1] = keys_tmp->ptr;
    st.vptr[0] = values ? values->ptr : nullptr;
    st.vptr[1] = values ? values_tmp->ptr : nullptr;
    st.n = n;
    st.key_bytes = key_bytes;
    st.deferred = ctx->os_async;
    int rc = reprobe_if_drifted(ctx);
    if (rc == VRS_OK) rc = one_read_enqueue(ctx);
    if (rc) {
        st.active = false;
        return rc;
    }
    return st.deferred ? VRS_OK : one_read_settle(ctx);
}

// One-call form: the four passes of MultiRadixSort::execute's hot loop (MultiRadixSort.cpp:50-61) with the
// library choosing NUM_BLOCKS_PER_WORKGROUP and owning the histogram table.
int sort_all_passes(vrs_context ctx, vrs_buffer keys, vrs_buffer keys_tmp, vrs_buffer values,
                           vrs_buffer values_tmp, uint32_t n, int key_bytes, uint32_t key_base) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    if (n == 0) return VRS_OK;
    const uint32_t B = launch_tile_blocks(key_bytes);
    vrs_push_constants pc{n, 0, vrs_workgroup_count(n, B), B};
    int rc;
    {
        const size_t bytes = static_cast<size_t>(n) * key_bytes;
        if ((rc = check_buffer(ctx, keys, bytes, "keys"))) return rc;
        if ((rc = check_buffer(ctx, keys_tmp, bytes, "keys_tmp"))) return rc;
        if (keys->ptr == keys_tmp->ptr) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "keys and keys_tmp alias");
        if (values) {
            const size_t vbytes = static_cast<size_t>(n) * sizeof(uint32_t);
            if ((rc = check_buffer(ctx, values, vbytes, "values"))) return rc;
            if ((rc = check_buffer(ctx, values_tmp, vbytes, "values_tmp"))) return rc;
            if (values->ptr == values_tmp->ptr) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "values and values_tmp alias");
        }
    }
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = one_read_settle(ctx))) return rc;  // an earlier async sort may still owe its second half
    ctx->sub_cache.valid = false;  // the keys are rewritten in place
    // small N: the whole sort in ONE launch of the single-workgroup kernel instead of twelve launch-bound ones (the
    // reference's own guidance: its single_radixsort is the faster path for small inputs, README.md:18-21)
    const vrs::SortFormId form = vrs::sort_form_for(n, key_bytes, values != nullptr, sort_knobs(ctx)).form;  // (single / contract / one of the one-read forms)
    if (form == vrs::kFormSingle) {
        vrs::LaunchEvents ev;
        if ((rc = profile_events(ctx, VRS_KERNEL_SINGLE, &ev))) return rc;
        VRS_HIP(ctx, vrs::launch_single(ctx->stream, static_cast<uint32_t *>(keys->ptr), static_cast<uint32_t *>(keys_tmp->ptr),
                                        n, ev));
        return VRS_OK;
    }
    if (form != vrs::kFormContract)  // (which of the one-read forms: decided again, with the memory's side effects, by one_read_enqueue)
        return sort_one_read(ctx, keys, keys_tmp, values, values_tmp, n, key_bytes, key_base);
    if ((rc = ensure_sort_hist(ctx, pc.g_num_workgroups))) return rc;
    for (uint32_t i = 0; i < static_cast<uint32_t>(key_bytes); ++i) {  // one pass per key byte: 4 or 8 (even either way)
        const bool odd = (i & 1u) != 0;
        if ((rc = contract_pass(ctx, odd ? keys_tmp : keys, odd ? keys : keys_tmp, values ? (odd ? values_tmp : values) : nullptr,
                                values ? (odd ? values : values_tmp) : nullptr, &pc, 8 * i, key_bytes)))
            return rc;
    }
    return VRS_OK;
}
}  // namespace vrsh

extern "C" {

int vrs_sort_form_for(uint32_t num_elements, int key_bytes, int pairs, const int64_t *knobs, int knob_count, int *form, int64_t *memory_out) {
    if (!form || (key_bytes != 4 && key_bytes != 8) || knob_count < 0 || knob_count > VRS_FORM_KNOB_COUNT || (knob_count && !knobs))
        return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "form is NULL, key_bytes is not 4 or 8, or the knobs do not match their count");
    vrs::SortKnobs k;  // (a fresh context on a device whose probes passed)
    const auto set = [&](int id, auto &field) {
        if (id < knob_count && knobs[id] >= 0) field = static_cast<std::remove_reference_t<decltype(field)>>(knobs[id]);
    };
    set(VRS_FORM_KNOB_SINGLE_MAX_KEYS, k.single_max_keys);
    set(VRS_FORM_KNOB_ONE_CALL_MIN_KEYS, k.one_call_min_keys);
    set(VRS_FORM_KNOB_HYBRID_MIN_KEYS, k.hybrid_min_keys);
    set(VRS_FORM_KNOB_POOL_MIN_KEYS, k.pool_min_keys);
    set(VRS_FORM_KNOB_HYBRID, k.hybrid);
    set(VRS_FORM_KNOB_POOL, k.pool);
    set(VRS_FORM_KNOB_POOL_PAIRS, k.pool_pairs);
    set(VRS_FORM_KNOB_RESERVE, k.reserve);
    set(VRS_FORM_KNOB_GROUPS, k.groups);
    set(VRS_FORM_KNOB_XCC_MAP_VALID, k.xcc_map_valid);
    set(VRS_FORM_KNOB_ATOMIC_RANK, k.atomic_rank);
    set(VRS_FORM_KNOB_POOL_SKIP, k.pool_skip);
    set(VRS_FORM_KNOB_POOL_SKIP_N, k.pool_skip_n);
    set(VRS_FORM_KNOB_WIDE_REFUSED, k.wide_refused);
    set(VRS_FORM_KNOB_WIDE_SKIPPED, k.wide_skipped);
    set(VRS_FORM_KNOB_NO_POOL, k.no_pool);
    set(VRS_FORM_KNOB_NO_HYBRID, k.no_hybrid);
    const vrs::SortDecision d = vrs::sort_form_for(num_elements, key_bytes, pairs != 0, k);
    *form = d.form;
    if (memory_out) {
        memory_out[0] = d.pool_skip;
        memory_out[1] = d.wide_skipped;
    }
    return VRS_OK;
}

int vrs_sort_settle(vrs_context ctx) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    if (!ctx->one_read.active) return VRS_OK;
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    return one_read_settle(ctx);
}

int vrs_sort_pending(vrs_context ctx) { return ctx && ctx->one_read.active ? 1 : 0; }

int vrs_sort_keys_u64(vrs_context ctx, vrs_buffer keys, vrs_buffer keys_tmp, uint32_t num_elements) {
    return sort_all_passes(ctx, keys, keys_tmp, nullptr, nullptr, num_elements, 8);
}

int vrs_sort_keys_u32(vrs_context ctx, vrs_buffer keys, vrs_buffer keys_tmp, uint32_t num_elements) {
    return sort_all_passes(ctx, keys, keys_tmp, nullptr, nullptr, num_elements);
}

int vrs_sort_keys_u32_ranged(vrs_context ctx, vrs_buffer keys, vrs_buffer keys_tmp, uint32_t num_elements, uint32_t key_floor) {
    return sort_all_passes(ctx, keys, keys_tmp, nullptr, nullptr, num_elements, 4, key_floor);
}

int vrs_sort_pairs_u32(vrs_context ctx, vrs_buffer keys, vrs_buffer keys_tmp, vrs_buffer values,
                       vrs_buffer values_tmp, uint32_t num_elements) {
    if (!values || !values_tmp) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "values buffers are NULL");
    return sort_all_passes(ctx, keys, keys_tmp, values, values_tmp, num_elements);
}

int vrs_sort_pairs_u64(vrs_context ctx, vrs_buffer keys, vrs_buffer keys_tmp, vrs_buffer values,
                       vrs_buffer values_tmp, uint32_t num_elements) {
    if (!values || !values_tmp) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "values buffers are NULL");
    return sort_all_passes(ctx, keys, keys_tmp, values, values_tmp, num_elements, 8);
}
}  // extern "C"
