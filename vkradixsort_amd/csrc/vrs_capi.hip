// vrs_capi.hip -- implementation of the C ABI declared in include/vkradixsort_amd.h.
// Host side only: handles, argument validation, stream-ordered launches, event profiling.
// There is NO CPU fallback: without a HIP device every entry point fails with VRS_ERROR_NO_DEVICE.
#include "vkradixsort_amd.h"

#include <hip/hip_runtime.h>

#include <sched.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <set>
#include <string>
#include <vector>

#include "vrs_kernels.h"

struct vrs_context_t {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    std::string last_error;
    vrs::PrefixScratch scratch;
    uint32_t scratch_workgroups = 0;  // capacity of scratch.offsets in workgroups
    uint32_t scratch_chunks = 0;      // capacity of scratch.chunk_sums in chunks
    uint32_t last_offsets_workgroups = 0;  // contract workgroups of the most recent RADIX_SORT stage
    void *offsets_hook_out = nullptr;      // one-shot (vrs_multi_radixsort_offsets_hook): the next RADIX_SORT stage copies its digit
    void *offsets_hook_event = nullptr;    //   offsets here and records this event BEFORE its scatter kernel
    uint32_t last_offsets_stride = 1;      // rows of scratch.offsets per contract workgroup (sub-tiled launches)
    // NUM_BLOCKS_PER_WORKGROUP > 32: the histogram stage also keeps the 8192-key sub-tile table it folded the
    // caller's table from; the sort stage uses it iff it is called for exactly the same (keys, N, shift, B)
    uint32_t *sub_hist = nullptr;
    uint32_t sub_hist_rows = 0;
    struct {
        const void *keys = nullptr;
        const void *hist = nullptr;  // the caller's table the sub-tile table was folded into
        uint32_t n = 0, shift = 0, blocks = 0;
        int key_bytes = 4;
        bool valid = false;
    } sub_cache;
    bool xcd_remap = true;
    bool fused_prefix = true;
    vrs::ScatterLaunch scatter;
    bool atomic_rank_verified = false;  // device self-test result (context creation)
    vrs_buffer sort_hist = nullptr;     // histogram table owned by the one-call entry points
    // profiling
    uint32_t profile_mask = 0;  // bit k: attach timing events to launches of vrs_kernel_id k
    struct EventPair {
        hipEvent_t start, stop;
    };
    std::vector<EventPair> events[VRS_KERNEL_COUNT];
    size_t events_used[VRS_KERNEL_COUNT] = {};
    // one-call sort for large N (K5 in vrs_one_call.hip)
    uint32_t one_call_min_keys = 1u << 13;  // measured: the one-read form wins from the single-launch threshold on (profiles/r02_one_call_crossover.csv)
    uint32_t single_max_keys = 4096;     // one-call uint32 key sorts up to this size run as ONE single_radixsort launch
    uint32_t *os_tables = nullptr;       // [4][kStreams][256] digit tables, zero between sorts
    vrs::OnesweepPlan *os_plan = nullptr;
    uint32_t *os_status = nullptr;       // look-back status rows
    size_t os_status_rows = 0;
    bool os_status_clean = false;        // every status word is zero: the last kernel on the stream that touched them was a local sort that cleared them
    vrs::OnesweepPlanHead *os_host_head = nullptr;      // pinned host copy of the plan's head (the plan kernel writes it)
    vrs::OnesweepPlanHead *os_host_head_dev = nullptr;  // the same memory as the device sees it
    uint32_t os_stamp = 0;               // stamp of the most recent plan (never 0)
    uint32_t *os_ticket = nullptr;       // fused plan: ticket word of the counting read's workgroups (zero between launches)
    bool os_fused_plan = false;          // the counting read's last workgroup makes the plan (VRS_TUNE_FUSED_PLAN)
    uint32_t os_groups = 0;              // groups per pass of the counting read: 8, 16, 32 or 0 = by size, VRS_TUNE_DIGIT_TABLE_GROUPS
    uint32_t os_spin_budget = 4096;      // polls of an unpublished look-back row before a tile recounts, VRS_TUNE_LOOKBACK_SPIN_BUDGET
    int os_hold_tile = -1;               // test hook, VRS_TUNE_DEBUG_HOLD_TILE
    bool os_misplace = false;            // test hook, VRS_TUNE_DEBUG_MISPLACE_STREAMS
    uint32_t drift_seen = 0;             // OnesweepPlanHead::drift (host copy) as of the last probe
    uint64_t reprobes = 0;               // probes run because sorts reported workgroups off the probed placement
    bool xcc_map_valid = false;          // the probe found block b on an XCC that depends on b % 8 only
    unsigned long long xcc_map = 0;      // byte x = that XCC for b % 8 == x
    uint64_t os_lookback_passes = 0;
    uint64_t os_relaunched_passes = 0;
    // hybrid form (K5b)
    bool os_hybrid = true;               // VRS_TUNE_HYBRID
    int os_fast_count = 1;               // VRS_TUNE_HYBRID_FAST_COUNT: 0 never, 1 adaptive, 2 always
    bool os_fast_count_armed[2] = {false, false};  // adaptive: the context's last hybrid-capable sort of keys [0] / pairs [1] took the hybrid form
    bool os_wide_refused = false;        // 64-bit keys: the last attempt at the hybrid form was refused
    uint32_t os_wide_skipped = 0;        //   ... sorts since (every 16th tries again)
    uint64_t os_hybrid_recounts = 0;     // sorts that started over as LSD sorts after a fast count and a refusal
    // VRS_TUNE_HYBRID_MIN_KEYS; 0 (default) = the measured crossovers per kind of sort: 1.3e7 bare uint32 keys (small buckets are
    // sorted one wave per bucket: profiles/labs/r03_hybrid_by_size.txt), 2.5e7 pairs, 2e7 64-bit keys (profiles/labs/r02_*).  A set
    // value v means v keys, 5/8 v pairs, v/2 64-bit keys.
    uint32_t os_hybrid_min_keys = 0u;
    uint32_t *os_msd_counts = nullptr;   // [16384] top-14-bit histogram + [8][256] top-byte counts per pass-0 group, zero between sorts
    vrs::MsdPlan *os_msd_plan = nullptr;
    vrs::OnesweepPlan *os_plan_a = nullptr;  // seeds and streams of the first MSD pass
    uint64_t os_hybrid_sorts = 0;        // one-call sorts that took the hybrid form
    uint64_t os_fallback_passes = 0;
    uint64_t os_skipped_passes = 0;      // identity passes (one digit value holds every key) the one-call sort left out     // passes the one-call sort ran through the contract path (unbalanced streams)
    bool os_async = true;                // VRS_TUNE_ASYNC_SORT (default 1): the one-call sorts return without waiting for the plan; vrs_sort_settle finishes them
    uint32_t os_plan_wait_ms = 60000;    // VRS_TUNE_PLAN_WAIT_MS: longest wait for a plan's head (0 = no limit)
    int os_reserve = 1;                  // VRS_TUNE_MSD_RESERVE: the MSD passes over bare keys reserve their output instead of looking back
                                         // (1 or 2: whenever the hybrid form runs on bare keys; 0: never)
    // pool form (vrs_msd_pool.hip): the hybrid form of bare uint32 keys without the counting read
    int os_pool = 1;                     // VRS_TUNE_MSD_POOL: 0 never, 1 (default) adaptive -- after a refusal the next 15 such sorts take the counted form --, 2 always tried
    uint32_t os_pool_skip = 0;           // adaptive: hybrid-capable sorts of bare keys left before the pool form is tried again
    uint32_t os_pool_skip_n = 0;         //   keys of the sort whose refusal started the count: a sort of another size class (beyond a factor of two) is another workload and starts afresh
    vrs::PoolPlan *os_pool_plan = nullptr;
    uint32_t *os_pool_overflow = nullptr;  // overflow regions of the first pass
    uint32_t os_pool_overflow_cap = 0;     //   keys they hold
    uint32_t *os_pool_slack = nullptr;     // the buckets' regions the second pass scatters into (about 1.5 n slots)
    uint32_t os_pool_slack_cap = 0;        //   slots
    uint32_t *os_pool_overflow_vals = nullptr, *os_pool_slack_vals = nullptr;  // key + payload pairs: the payloads' twins of the two (made with the first pool sort of pairs)
    uint32_t os_pool_vals_overflow_cap = 0, os_pool_vals_slack_cap = 0;
    int os_pool_top_bits = 7;              // VRS_TUNE_MSD_POOL_TOP_BITS: how a sort's 16384 buckets are cut between the two passes -- 7 + 7 bits (default: the first pass, which
                                           // reads cold input, writes 64-key segments instead of 32-key ones: pairs -2.4 %, 10^7 keys -2.5 %, 10^8 keys -1 %), 8 + 6, or 6 + 8
    int os_pool_pairs = 1;                 // VRS_TUNE_MSD_POOL_PAIRS: key + payload pairs may take the (stable) pool form
    uint64_t os_pool_pair_sorts = 0;
    uint64_t os_pool_sorts = 0, os_pool_refusals = 0, os_pool_retries = 0;  // (retries: sorts whose local sort was enqueued again in a larger shape)
    uint32_t os_pool_min_keys = 1u << 22;   // VRS_TUNE_MSD_POOL_MIN_KEYS: the form's own floor -- with one wave per small bucket it beats the LSD passes from there on (labs/r05_pool_form.txt section 6)
    int os_pool_sub_bits = 0;               // VRS_TUNE_MSD_POOL_SUB_BITS: 0 = by size (pool_shape), 6 or 7
    uint32_t os_pool_epoch = 0;             // pool sorts / finishes enqueued: its parity picks the PoolPlan::fail word of each
    // The regions of the first pass, kept from one sort to the next (VRS_TUNE_MSD_POOL_REUSE_LAYOUT, default on): a sort of the same
    // size and key floor as the context's last TAKEN pool sort runs its first pass in the regions that sort's sample laid out -- no
    // sample and layout kernel (13 us and two launch gaps at 10^8 keys).  Nothing is trusted: a region that does not fit, a key
    // outside the kept range flag the sort as ever; a refusal forgets the layout and the re-run samples.
    bool os_pool_reuse = true;
    bool os_pool_reuse_rooms = true;      // ... and the buckets' slack regions with them (the plan kernel then samples nothing): VRS_TUNE_MSD_POOL_REUSE_LAYOUT = 2 keeps the first pass's regions only
    bool os_pool_layout_valid = false;
    uint32_t os_pool_layout_n = 0, os_pool_layout_base = 0, os_pool_layout_sub_bits = 0;
    uint32_t os_pool_stale_run = 0, os_pool_reuse_pause = 0;  // kept layouts found stale in a row / sorts left that sample for themselves although a layout is kept
    uint32_t os_pool_fail_alloc = 0;       // VRS_TUNE_DEBUG_POOL_NO_MEMORY: allocations of the pool form's scratch left to fail (test hook)
    uint64_t os_pool_no_memory = 0;        // pool sorts / finishes that found no room for the form's scratch and took another form
    uint64_t os_pool_layout_reuses = 0, os_pool_stale_layouts = 0;  // sorts that started in a kept layout / of those, sorts it did not fit (run again, sampled)
    bool os_cursors_open = false;        // a reserving pass may have run without the local sort that re-arms its counters behind it
                                         // (a refused plan, a partition with no finish, an error in between): cleared before the next use
    // a one-call sort between its two halves (see one_read_enqueue / one_read_complete)
    struct OneRead {
        bool active = false;    // enqueued, its plan not yet looked at
        bool deferred = false;  // async mode: the caller did not wait
        void *kptr[2] = {nullptr, nullptr}, *vptr[2] = {nullptr, nullptr};  // [0] the caller's buffers, [1] the ping-pong partners
        uint32_t n = 0;
        int key_bytes = 4;
        uint32_t group = 0;     // group of four passes that is on the stream
        uint32_t stamp = 0;     // of that group's plan
        uint32_t cur = 0, cur_at_start = 0;  // which of the two buffers holds the data (now / when the group started)
        uint32_t blind_passes = 0;
        bool msd_capable = false, fast_count = false, blind_tail = false, no_hybrid = false;
        bool pool = false, no_pool = false;  // the pool form is on the stream / was refused for this sort
        uint32_t pool_sub_bits = 0, pool_local = 0;  // its shape; pool_retried: a larger local sort has been enqueued behind a first one that left
        bool pool_retried = false;
        uint32_t pool_top_bits = 8;  // bits of its first pass
        uint32_t pool_par = 0;     // parity of its pool epoch
        bool pool_reused = false;  // its first pass ran in a kept layout
        size_t ev_lb_before = 0, ev_ls_before = 0;
        uint32_t key_base = 0;      // vrs_sort_keys_u32_ranged: every key is promised to be >= this (a multiple of 2^24)
        uint32_t bucket_hint = 0;   // blind tail: expected largest bucket (0 = from n); picks the local sort's workgroup shape
        uint32_t sub_bits = 6;      // bucket bits the second MSD pass sorts by (what its plan is made with)
        uint32_t pass_b_groups = 0; // second MSD pass over fewer than 256 groups (vrs_msd_finish_grouped_u32): XCD x walks groups x, x + 8, ...
    } one_read;
    bool one_read_settling = false;
    uint32_t os_msd_half_stamp = 0;      // stamp of the most recent vrs_msd_finish_u32's plan
};

struct vrs_buffer_t {
    vrs_context ctx = nullptr;  // owner, compared for identity only after creation (may be destroyed before the buffer)
    int device = 0;
    void *ptr = nullptr;
    size_t size = 0;
    bool owned = false;
};

namespace {

// the kernels' own launch tile: 8192 uint32 keys or 4096 uint64 keys (32 KiB either way)
constexpr uint32_t launch_tile_blocks(int key_bytes) { return key_bytes == 8 ? 16u : 32u; }

thread_local std::string g_global_error;
// Contexts that are alive (vrs_context_create* ... vrs_context_destroy): a buffer may outlive its context (host-language finalisers
// run in any order), so vrs_buffer_release asks here before it touches buf->ctx.
std::mutex g_live_mutex;
std::set<vrs_context> g_live_contexts;

// second half of a pending async one-call sort (vrs_capi.hip, "one_read_settle"); every entry point that puts work on the
// stream or waits for it calls this first
int settle_pending(vrs_context ctx);

int fail(vrs_context ctx, int code, const std::string &msg) {
    if (ctx)
        ctx->last_error = msg;
    else
        g_global_error = msg;
    return code;
}

int fail_hip(vrs_context ctx, const char *what, hipError_t e) {
    std::string msg = std::string(what) + ": " + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ")";
    return fail(ctx, e == hipErrorOutOfMemory ? VRS_ERROR_OUT_OF_MEMORY : VRS_ERROR_HIP, msg);
}

#define VRS_HIP(ctx, call)                                       \
    do {                                                         \
        hipError_t e__ = (call);                                 \
        if (e__ != hipSuccess) return fail_hip((ctx), #call, e__); \
    } while (0)

int ensure_scratch(vrs_context ctx, uint32_t W) {
    const uint32_t C = vrs::prefix_chunk_tiles(W);
    const uint32_t G = (W + C - 1) / C;
    if (W > ctx->scratch_workgroups) {
        if (ctx->scratch.offsets) VRS_HIP(ctx, hipFree(ctx->scratch.offsets));
        ctx->scratch.offsets = nullptr;
        ctx->scratch_workgroups = 0;
        VRS_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->scratch.offsets),
                               static_cast<size_t>(W) * VRS_RADIX_SORT_BINS * sizeof(uint32_t)));
        ctx->scratch_workgroups = W;
    }
    if (G > ctx->scratch_chunks) {
        if (ctx->scratch.chunk_sums) VRS_HIP(ctx, hipFree(ctx->scratch.chunk_sums));
        if (ctx->scratch.granules) VRS_HIP(ctx, hipFree(ctx->scratch.granules));
        ctx->scratch.chunk_sums = nullptr;
        ctx->scratch.granules = nullptr;
        ctx->scratch_chunks = 0;
        VRS_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->scratch.chunk_sums),
                               static_cast<size_t>(G) * VRS_RADIX_SORT_BINS * sizeof(uint32_t)));
        const size_t gbytes = static_cast<size_t>(G) * VRS_RADIX_SORT_BINS * sizeof(unsigned long long);
        VRS_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->scratch.granules), gbytes));
        VRS_HIP(ctx, hipMemsetAsync(ctx->scratch.granules, 0, gbytes, ctx->stream));  // tag 0 == never published
        ctx->scratch_chunks = G;
    }
    // the fused single-launch prefix needs all G chunk workgroups resident at once
    ctx->scratch.fused_max_chunks =
        ctx->fused_prefix ? std::min<uint32_t>(ctx->scratch_chunks, static_cast<uint32_t>(ctx->scatter.compute_units)) : 0u;
    if (++ctx->scratch.epoch == 0) ctx->scratch.epoch = 1;
    return VRS_OK;
}

// Hands out a (start, stop) event pair for one launch when profiling is on; the events ride on the
// kernel's own dispatch packet (hipExtLaunchKernel), so profiling does not insert barrier packets.
int profile_events(vrs_context ctx, int id, vrs::LaunchEvents *ev) {
    *ev = vrs::LaunchEvents{};
    if (!(ctx->profile_mask & (1u << id))) return VRS_OK;
    auto &pool = ctx->events[id];
    if (ctx->events_used[id] == pool.size()) {
        vrs_context_t::EventPair p{};
        VRS_HIP(ctx, hipEventCreate(&p.start));
        VRS_HIP(ctx, hipEventCreate(&p.stop));
        pool.push_back(p);
    }
    const auto &pair = pool[ctx->events_used[id]++];
    ev->start = pair.start;
    ev->stop = pair.stop;
    return VRS_OK;
}

int check_push_constants(vrs_context ctx, const vrs_push_constants *pc, int key_bytes = 4) {
    if (!pc) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "push constants are NULL");
    if (pc->g_shift > 8u * key_bytes - 8u || (pc->g_shift & 7u) != 0)
        return fail(ctx, VRS_ERROR_INVALID_ARGUMENT,
                    key_bytes == 8 ? "g_shift must be a multiple of 8 in [0, 56]" : "g_shift must be 0, 8, 16 or 24");
    if (pc->g_num_elements == 0) return VRS_OK;
    if (pc->g_num_blocks_per_workgroup == 0)
        return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "g_num_blocks_per_workgroup must be >= 1");
    // the tiles described by (W, B) must cover all N keys, otherwise keys would silently be dropped
    const uint64_t covered = static_cast<uint64_t>(pc->g_num_workgroups) * pc->g_num_blocks_per_workgroup *
                             VRS_WORKGROUP_SIZE;
    if (covered < pc->g_num_elements)
        return fail(ctx, VRS_ERROR_INVALID_ARGUMENT,
                    "g_num_workgroups * g_num_blocks_per_workgroup * 256 does not cover g_num_elements");
    // and no workgroup may start past the end (the reference never launches one: ComputePass.h:24-29)
    const uint64_t last_begin = static_cast<uint64_t>(pc->g_num_workgroups - 1) * pc->g_num_blocks_per_workgroup *
                                VRS_WORKGROUP_SIZE;
    if (last_begin >= pc->g_num_elements)
        return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "g_num_workgroups is larger than ceil(ceil(N/B)/256)");
    return VRS_OK;
}

int check_buffer(vrs_context ctx, vrs_buffer b, size_t need, const char *name) {
    if (!b || !b->ptr) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, std::string(name) + " buffer is NULL or released");
    if (b->ctx != ctx) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, std::string(name) + " buffer belongs to another context");
    if (b->size < need) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, std::string(name) + " buffer is too small");
    return VRS_OK;
}

// Runs the lane-order self-test the RANK_ATOMIC scatter variants depend on (see vrs_contract.hip).
int atomic_rank_selftest(vrs_context ctx, uint32_t rounds, uint32_t seed, uint64_t *mismatches) {
    unsigned long long *d = nullptr;
    VRS_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&d), sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d, 0, sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess) e = vrs::launch_atomic_rank_selftest(ctx->stream, rounds, seed, d);
    unsigned long long h = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail_hip(ctx, "atomic rank selftest", e);
    *mismatches = h;
    return VRS_OK;
}

// Where do the blocks of a big grid run?  The look-back streams of the one-call sort are laid out for "block b runs
// on XCC b % 8" (any fixed function of b % 8 will do, a single XCC included).  That placement is observed, not
// promised, so it is probed here; the kernels re-check it per workgroup and stay correct without it.
int probe_xcc_map(vrs_context ctx, int stray_block = -1) {
    constexpr uint32_t kBlocks = 4096;
    uint32_t *d = nullptr;
    VRS_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&d), kBlocks * sizeof(uint32_t)));
    std::vector<uint32_t> h(kBlocks, 0);
    hipError_t e = vrs::launch_xcc_probe(ctx->stream, d, kBlocks);
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d, kBlocks * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail_hip(ctx, "XCC placement probe", e);
    // test hook (VRS_TUNE_DEBUG_XCC_STRAY_BLOCK): pretend ONE block of the probe ran elsewhere -- a placement that holds for most
    // blocks only must switch every form that leans on it off, like one that holds for none
    if (stray_block >= 0 && static_cast<uint32_t>(stray_block) < kBlocks) h[static_cast<size_t>(stray_block)] ^= 1u;
    unsigned long long map = 0;
    bool valid = true;
    for (uint32_t b = 0; b < kBlocks && valid; ++b) {
        if (b < 8)
            map |= static_cast<unsigned long long>(h[b] & 0xFFu) << (8 * b);
        else
            valid = h[b] == h[b & 7u];
    }
    ctx->xcc_map = map;
    ctx->xcc_map_valid = valid;
    return VRS_OK;
}

// the word of the pinned host head the look-back / reserving kernels report placement drift in (device view), once it exists
static uint32_t *drift_word(vrs_context ctx) { return ctx->os_host_head_dev ? &ctx->os_host_head_dev->drift : nullptr; }
// Before a one-call sort is enqueued: did workgroups of earlier sorts find themselves on other XCCs than the probe said (the stream
// moved to another hardware queue, whose round-robin starts elsewhere)?  Then the probe is run again -- once; the sorts in between
// were exact, on their placement-independent routes.
static int reprobe_if_drifted(vrs_context ctx) {
    if (!ctx->os_host_head) return VRS_OK;
    const uint32_t now = __atomic_load_n(&ctx->os_host_head->drift, __ATOMIC_RELAXED);
    if (now == ctx->drift_seen) return VRS_OK;
    VRS_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (late reports of what is still running belong to the old placement too)
    const bool was_valid = ctx->xcc_map_valid;
    const int rc = probe_xcc_map(ctx);
    if (rc) return rc;
    if (!was_valid) ctx->xcc_map_valid = false;  // (a placement that was found broken stays distrusted: only the order is refreshed)
    ctx->drift_seen = __atomic_load_n(&ctx->os_host_head->drift, __ATOMIC_RELAXED);
    ctx->reprobes++;
    return VRS_OK;
}

int create_context(int device_ordinal, hipStream_t borrowed, bool borrow, vrs_context *out_ctx) {
    if (!out_ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "out_ctx is NULL");
    *out_ctx = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(nullptr, VRS_ERROR_NO_DEVICE,
                    std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    if (device_ordinal < 0 || device_ordinal >= count)
        return fail(nullptr, VRS_ERROR_NO_DEVICE, "device ordinal out of range");
    e = hipSetDevice(device_ordinal);
    if (e != hipSuccess) return fail_hip(nullptr, "hipSetDevice", e);
    vrs_context ctx = new (std::nothrow) vrs_context_t();
    if (!ctx) return fail(nullptr, VRS_ERROR_OUT_OF_MEMORY, "host allocation failed");
    ctx->device = device_ordinal;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0)
            ctx->scatter.compute_units = prop.multiProcessorCount;
    }
    if (borrow) {
        ctx->stream = borrowed;
        ctx->owns_stream = false;
        // whoever lends a stream waits on it with calls of his own (hipStreamSynchronize, torch.cuda.synchronize): what a sort puts
        // on such a stream must be the whole sort when the call returns -- the blocking form is the default there
        ctx->os_async = false;
    } else {
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete ctx;
            return fail_hip(nullptr, "hipStreamCreateWithFlags", e);
        }
        ctx->owns_stream = true;
    }
    // Pick the ranking method for this device: returning LDS atomics are ~25 % faster in the scatter but
    // need same-address lanes served in ascending lane order, which is observed, not promised.  Probe it
    // (about 0.3 ms); fall back to the __ballot ranking if a single lane disagrees.
    {
        uint64_t mismatches = 1;
        if (atomic_rank_selftest(ctx, 512, 0x5EEDu, &mismatches) == VRS_OK && mismatches == 0)
            ctx->atomic_rank_verified = true;
        ctx->scatter.atomic_rank = ctx->atomic_rank_verified;
        (void)probe_xcc_map(ctx);
        ctx->last_error.clear();
    }
    {
        std::lock_guard<std::mutex> lock(g_live_mutex);
        g_live_contexts.insert(ctx);
    }
    *out_ctx = ctx;
    return VRS_OK;
}

int run_sort_stage(vrs_context ctx, vrs_buffer keys_in, vrs_buffer keys_out, vrs_buffer values_in,
                   vrs_buffer values_out, vrs_buffer histograms, const vrs_push_constants *pc, bool pairs,
                   int key_bytes = 4) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    int rc = check_push_constants(ctx, pc, key_bytes);
    if (rc) return rc;
    const uint32_t n = pc->g_num_elements;
    if (n == 0) return VRS_OK;
    const uint32_t W = pc->g_num_workgroups;
    const size_t keys_size = static_cast<size_t>(n) * key_bytes, values_size = static_cast<size_t>(n) * sizeof(uint32_t);
    if ((rc = check_buffer(ctx, keys_in, keys_size, "keys_in"))) return rc;
    if ((rc = check_buffer(ctx, keys_out, keys_size, "keys_out"))) return rc;
    if ((rc = check_buffer(ctx, histograms, static_cast<size_t>(W) * VRS_RADIX_SORT_BINS * sizeof(uint32_t),
                           "histograms")))
        return rc;
    if (keys_in->ptr == keys_out->ptr) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "keys_in and keys_out alias");
    if ((reinterpret_cast<uintptr_t>(keys_in->ptr) | reinterpret_cast<uintptr_t>(keys_out->ptr)) & (key_bytes - 1))
        return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "key buffers must be aligned to the key size");
    if (pairs) {
        if ((rc = check_buffer(ctx, values_in, values_size, "values_in"))) return rc;
        if ((rc = check_buffer(ctx, values_out, values_size, "values_out"))) return rc;
        if (values_in->ptr == values_out->ptr)
            return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "values_in and values_out alias");
    }
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    if (const int settled = settle_pending(ctx)) return settled;  // an async one-call sort may still owe its second half
    const uint32_t B = pc->g_num_blocks_per_workgroup;
    const uint32_t *table = static_cast<const uint32_t *>(histograms->ptr);
    const uint32_t kLaunchTileBlocks = launch_tile_blocks(key_bytes);
    uint32_t launch_W = W, launch_B = B, row_stride = 1, rows_per_contract_tile = 1;
    if (B > kLaunchTileBlocks && B % kLaunchTileBlocks == 0 && ctx->sub_cache.valid &&
        ctx->sub_cache.keys == keys_in->ptr && ctx->sub_cache.hist == histograms->ptr && ctx->sub_cache.n == n &&
        ctx->sub_cache.shift == pc->g_shift && ctx->sub_cache.blocks == B && ctx->sub_cache.key_bytes == key_bytes) {
        // large contract tiles: prefix + scatter at 8192-key sub-tile granularity from the table the histogram
        // stage kept (the caller's table is its fold, so both describe the same keys)
        launch_B = kLaunchTileBlocks;
        launch_W = vrs_workgroup_count(n, launch_B);
        rows_per_contract_tile = B / kLaunchTileBlocks;
        table = ctx->sub_hist;
    } else if (B < kLaunchTileBlocks && kLaunchTileBlocks % B == 0 && W > 1) {
        // small contract tiles: consecutive tiles are adjacent in every digit's output range, so the scatter
        // walks 8192-key launch tiles and takes the offset row of the first contract tile inside each
        row_stride = kLaunchTileBlocks / B;
    }
    ctx->sub_cache.valid = false;
    const uint32_t prefix_rows = rows_per_contract_tile > 1 ? launch_W : W;
    if ((rc = ensure_scratch(ctx, prefix_rows))) return rc;

    vrs::LaunchEvents ev;
    if ((rc = profile_events(ctx, VRS_KERNEL_PREFIX, &ev))) return rc;
    VRS_HIP(ctx, vrs::launch_prefix(ctx->stream, table, ctx->scratch, prefix_rows, ev));
    ctx->last_offsets_workgroups = W;
    ctx->last_offsets_stride = rows_per_contract_tile;
    if (ctx->offsets_hook_out || ctx->offsets_hook_event) {
        void *out = ctx->offsets_hook_out, *event = ctx->offsets_hook_event;
        ctx->offsets_hook_out = ctx->offsets_hook_event = nullptr;
        if (out)
            VRS_HIP(ctx, hipMemcpyAsync(out, ctx->scratch.offsets, VRS_RADIX_SORT_BINS * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
        if (event) VRS_HIP(ctx, hipEventRecord(static_cast<hipEvent_t>(event), ctx->stream));
    }

    if (row_stride > 1) {
        launch_B = kLaunchTileBlocks;
        launch_W = vrs_workgroup_count(n, launch_B);
    }
    if ((rc = profile_events(ctx, VRS_KERNEL_SCATTER, &ev))) return rc;
    VRS_HIP(ctx, vrs::launch_scatter(ctx->stream, keys_in->ptr, keys_out->ptr,
                                     pairs ? static_cast<const uint32_t *>(values_in->ptr) : nullptr,
                                     pairs ? static_cast<uint32_t *>(values_out->ptr) : nullptr, ctx->scratch.offsets,
                                     n, pc->g_shift, launch_W, launch_B, ctx->xcd_remap, ctx->scatter, ev, nullptr,
                                     row_stride, key_bytes));
    return VRS_OK;
}

}  // namespace

extern "C" {

const char *vrs_version(void) { return "vkradixsort_amd 0.1.0 (gfx950)"; }

int vrs_device_count(int *count) {
    if (!count) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "count is NULL");
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        *count = 0;
        return fail(nullptr, VRS_ERROR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    return VRS_OK;
}

int vrs_context_create(int device_ordinal, vrs_context *out_ctx) {
    return create_context(device_ordinal, nullptr, false, out_ctx);
}

int vrs_context_create_on_stream(int device_ordinal, void *hip_stream, vrs_context *out_ctx) {
    return create_context(device_ordinal, static_cast<hipStream_t>(hip_stream), true, out_ctx);
}

int vrs_context_destroy(vrs_context ctx) {
    if (!ctx) return VRS_OK;
    {
        std::lock_guard<std::mutex> lock(g_live_mutex);
        g_live_contexts.erase(ctx);
    }
    (void)hipSetDevice(ctx->device);
    (void)settle_pending(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &pool : ctx->events)
        for (auto &p : pool) {
            (void)hipEventDestroy(p.start);
            (void)hipEventDestroy(p.stop);
        }
    if (ctx->sort_hist) (void)vrs_buffer_release(ctx->sort_hist);
    if (ctx->scratch.offsets) (void)hipFree(ctx->scratch.offsets);
    if (ctx->scratch.chunk_sums) (void)hipFree(ctx->scratch.chunk_sums);
    if (ctx->scratch.granules) (void)hipFree(ctx->scratch.granules);
    if (ctx->sub_hist) (void)hipFree(ctx->sub_hist);
    if (ctx->os_tables) (void)hipFree(ctx->os_tables);
    if (ctx->os_plan) (void)hipFree(ctx->os_plan);
    if (ctx->os_status) (void)hipFree(ctx->os_status);
    if (ctx->os_host_head) (void)hipHostFree(ctx->os_host_head);
    if (ctx->os_pool_plan) (void)hipFree(ctx->os_pool_plan);
    if (ctx->os_pool_overflow) (void)hipFree(ctx->os_pool_overflow);
    if (ctx->os_pool_slack) (void)hipFree(ctx->os_pool_slack);
    if (ctx->os_pool_overflow_vals) (void)hipFree(ctx->os_pool_overflow_vals);
    if (ctx->os_pool_slack_vals) (void)hipFree(ctx->os_pool_slack_vals);
    if (ctx->os_msd_counts) (void)hipFree(ctx->os_msd_counts);
    if (ctx->os_msd_plan) (void)hipFree(ctx->os_msd_plan);
    if (ctx->os_plan_a) (void)hipFree(ctx->os_plan_a);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return VRS_OK;
}

const char *vrs_last_error(vrs_context ctx) { return ctx ? ctx->last_error.c_str() : g_global_error.c_str(); }

void *vrs_context_stream(vrs_context ctx) { return ctx ? static_cast<void *>(ctx->stream) : nullptr; }

int vrs_device_info(vrs_context ctx, char *name, size_t name_cap, int *compute_units, uint64_t *global_mem_bytes) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    hipDeviceProp_t prop;
    VRS_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_cap) std::snprintf(name, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (global_mem_bytes) *global_mem_bytes = prop.totalGlobalMem;
    return VRS_OK;
}

int vrs_buffer_create(vrs_context ctx, size_t size_bytes, vrs_buffer *out_buf) {
    if (!ctx || !out_buf) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "context or out_buf is NULL");
    *out_buf = nullptr;
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    vrs_buffer b = new (std::nothrow) vrs_buffer_t();
    if (!b) return fail(ctx, VRS_ERROR_OUT_OF_MEMORY, "host allocation failed");
    // a zero-sized Buffer is legal at the boundary (N == 0); keep a real allocation behind it
    hipError_t e = hipMalloc(&b->ptr, size_bytes ? size_bytes : 16);
    if (e != hipSuccess) {
        delete b;
        return fail_hip(ctx, "hipMalloc", e);
    }
    b->ctx = ctx;
    b->device = ctx->device;
    b->size = size_bytes;
    b->owned = true;
    *out_buf = b;
    return VRS_OK;
}

int vrs_buffer_wrap(vrs_context ctx, void *device_ptr, size_t size_bytes, vrs_buffer *out_buf) {
    if (!ctx || !out_buf) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "context or out_buf is NULL");
    *out_buf = nullptr;
    if (!device_ptr) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "device_ptr is NULL");
    if (reinterpret_cast<uintptr_t>(device_ptr) & 3u)
        return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "device_ptr must be 4-byte aligned");
    vrs_buffer b = new (std::nothrow) vrs_buffer_t();
    if (!b) return fail(ctx, VRS_ERROR_OUT_OF_MEMORY, "host allocation failed");
    b->ctx = ctx;
    b->device = ctx->device;
    b->ptr = device_ptr;
    b->size = size_bytes;
    b->owned = false;
    *out_buf = b;
    return VRS_OK;
}

int vrs_buffer_release(vrs_buffer buf) {
    if (!buf) return VRS_OK;
    int rc = VRS_OK;
    if (buf->ptr && buf->owned) {
        // the owning context may already be gone (host-language finalisers run in any order): touch it only if it is alive
        (void)hipSetDevice(buf->device);
        {
            // An enqueue-only sort (the default on a context with its own stream) may still owe its second half, which works on
            // the raw pointers of its four buffers: freeing one of them first would hand that half freed memory.  Settle it, and
            // let what is on the stream finish with the buffer, before the memory goes.  (hipFree waits for the device, but a
            // second half enqueued AFTER it would not be waited for.)
            // The registry's lock is held from the look-up to the end of the settle: vrs_context_destroy takes it before it frees anything,
            // so a context found alive here stays alive while its pending sort is settled (a destroy on another thread waits).  A context is
            // otherwise one thread's at a time (include/vkradixsort_amd.h, "Threading"): releasing a buffer WHILE another thread sorts with
            // its context is the caller's to serialise.
            std::lock_guard<std::mutex> lock(g_live_mutex);
            const bool alive = g_live_contexts.count(buf->ctx) != 0;
            if (alive && buf->ctx->one_read.active) {
                const vrs_context_t::OneRead &st = buf->ctx->one_read;
                const char *lo = static_cast<const char *>(buf->ptr), *hi = lo + buf->size;
                bool used = false;
                for (const void *q : {st.kptr[0], st.kptr[1], st.vptr[0], st.vptr[1]})
                    used = used || (q != nullptr && static_cast<const char *>(q) >= lo && static_cast<const char *>(q) < hi);
                if (used) {
                    const int settled = settle_pending(buf->ctx);
                    if (settled) rc = settled;
                }
            }
        }
        hipError_t e = hipFree(buf->ptr);
        if (e != hipSuccess) rc = fail_hip(nullptr, "hipFree", e);
    }
    buf->ptr = nullptr;
    delete buf;
    return rc;
}

int vrs_buffer_upload(vrs_context ctx, vrs_buffer buf, const void *host_data, size_t size_bytes) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    int rc = check_buffer(ctx, buf, size_bytes, "upload");
    if (rc) return rc;
    if (size_bytes == 0) return VRS_OK;
    if (!host_data) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "host_data is NULL");
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    if (const int settled = settle_pending(ctx)) return settled;  // an async one-call sort may still owe its second half
    ctx->sub_cache.valid = false;  // a buffer is rewritten: the kept sub-tile table may no longer describe its keys
    VRS_HIP(ctx, hipMemcpyAsync(buf->ptr, host_data, size_bytes, hipMemcpyHostToDevice, ctx->stream));
    VRS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VRS_OK;
}

int vrs_buffer_download(vrs_context ctx, vrs_buffer buf, void *host_data, size_t size_bytes) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    int rc = check_buffer(ctx, buf, size_bytes, "download");
    if (rc) return rc;
    if (size_bytes == 0) return VRS_OK;
    if (!host_data) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "host_data is NULL");
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    if (const int settled = settle_pending(ctx)) return settled;  // an async one-call sort may still owe its second half
    VRS_HIP(ctx, hipMemcpyAsync(host_data, buf->ptr, size_bytes, hipMemcpyDeviceToHost, ctx->stream));
    VRS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VRS_OK;
}

int vrs_buffer_copy(vrs_context ctx, vrs_buffer dst, vrs_buffer src, size_t size_bytes) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    int rc = check_buffer(ctx, dst, size_bytes, "copy dst");
    if (rc) return rc;
    if ((rc = check_buffer(ctx, src, size_bytes, "copy src"))) return rc;
    if (size_bytes == 0) return VRS_OK;
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    if (const int settled = settle_pending(ctx)) return settled;  // an async one-call sort may still owe its second half
    ctx->sub_cache.valid = false;  // a buffer is rewritten: the kept sub-tile table may no longer describe its keys
    VRS_HIP(ctx, hipMemcpyAsync(dst->ptr, src->ptr, size_bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return VRS_OK;
}

void *vrs_buffer_device_ptr(vrs_buffer buf) { return buf ? buf->ptr : nullptr; }

size_t vrs_buffer_size_bytes(vrs_buffer buf) { return buf ? buf->size : 0; }

uint32_t vrs_global_invocation_size(uint32_t num_elements, uint32_t blocks_per_workgroup) {
    if (blocks_per_workgroup == 0) return 0;
    return num_elements / blocks_per_workgroup + (num_elements % blocks_per_workgroup ? 1u : 0u);
}

uint32_t vrs_workgroup_count(uint32_t num_elements, uint32_t blocks_per_workgroup) {
    const uint32_t gis = vrs_global_invocation_size(num_elements, blocks_per_workgroup);
    return gis / VRS_WORKGROUP_SIZE + (gis % VRS_WORKGROUP_SIZE ? 1u : 0u);
}

static int run_histogram_stage(vrs_context ctx, vrs_buffer keys_in, vrs_buffer histograms,
                               const vrs_push_constants *pc, int key_bytes) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    int rc = check_push_constants(ctx, pc, key_bytes);
    if (rc) return rc;
    if (pc->g_num_elements == 0) return VRS_OK;
    if ((rc = check_buffer(ctx, keys_in, static_cast<size_t>(pc->g_num_elements) * key_bytes, "keys_in")))
        return rc;
    if (reinterpret_cast<uintptr_t>(keys_in->ptr) & (key_bytes - 1))
        return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "key buffers must be aligned to the key size");
    if ((rc = check_buffer(ctx, histograms,
                           static_cast<size_t>(pc->g_num_workgroups) * VRS_RADIX_SORT_BINS * sizeof(uint32_t),
                           "histograms")))
        return rc;
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    if (const int settled = settle_pending(ctx)) return settled;  // an async one-call sort may still owe its second half
    vrs::LaunchEvents ev;
    if ((rc = profile_events(ctx, VRS_KERNEL_HISTOGRAM, &ev))) return rc;
    const uint32_t B = pc->g_num_blocks_per_workgroup, n = pc->g_num_elements;
    const uint32_t kLaunchTileBlocks = launch_tile_blocks(key_bytes);
    ctx->sub_cache.valid = false;
    if (B > kLaunchTileBlocks && B % kLaunchTileBlocks == 0) {
        // contract tile = S sub-tiles of 8192 keys: histogram the sub-tiles (enough workgroups to fill the chip
        // whatever B is), then fold them into the caller's [W][256] table
        const uint32_t S = B / kLaunchTileBlocks;
        const uint32_t sub_rows = vrs_workgroup_count(n, kLaunchTileBlocks);
        if (sub_rows > ctx->sub_hist_rows) {
            if (ctx->sub_hist) VRS_HIP(ctx, hipFree(ctx->sub_hist));
            ctx->sub_hist = nullptr;
            ctx->sub_hist_rows = 0;
            VRS_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->sub_hist),
                                   static_cast<size_t>(sub_rows) * VRS_RADIX_SORT_BINS * sizeof(uint32_t)));
            ctx->sub_hist_rows = sub_rows;
        }
        VRS_HIP(ctx, vrs::launch_histograms(ctx->stream, keys_in->ptr, ctx->sub_hist, n, pc->g_shift, sub_rows,
                                            kLaunchTileBlocks, vrs::LaunchEvents{ev.start, nullptr}, nullptr, key_bytes));
        VRS_HIP(ctx, vrs::launch_fold_histograms(ctx->stream, ctx->sub_hist, static_cast<uint32_t *>(histograms->ptr),
                                                 sub_rows, pc->g_num_workgroups, S, vrs::LaunchEvents{nullptr, ev.stop}));
        ctx->sub_cache.keys = keys_in->ptr;
        ctx->sub_cache.hist = histograms->ptr;
        ctx->sub_cache.n = n;
        ctx->sub_cache.shift = pc->g_shift;
        ctx->sub_cache.blocks = B;
        ctx->sub_cache.key_bytes = key_bytes;
        ctx->sub_cache.valid = true;
        return VRS_OK;
    }
    VRS_HIP(ctx, vrs::launch_histograms(ctx->stream, keys_in->ptr, static_cast<uint32_t *>(histograms->ptr), n,
                                        pc->g_shift, pc->g_num_workgroups, B, ev, nullptr, key_bytes));
    return VRS_OK;
}

int vrs_multi_radixsort_histograms(vrs_context ctx, vrs_buffer keys_in, vrs_buffer histograms,
                                   const vrs_push_constants *pc) {
    return run_histogram_stage(ctx, keys_in, histograms, pc, 4);
}

int vrs_multi_radixsort_histograms_u64(vrs_context ctx, vrs_buffer keys_in, vrs_buffer histograms,
                                       const vrs_push_constants *pc) {
    return run_histogram_stage(ctx, keys_in, histograms, pc, 8);
}

int vrs_multi_radixsort_u64(vrs_context ctx, vrs_buffer keys_in, vrs_buffer keys_out, vrs_buffer histograms,
                            const vrs_push_constants *pc) {
    return run_sort_stage(ctx, keys_in, keys_out, nullptr, nullptr, histograms, pc, false, 8);
}

int vrs_multi_radixsort_pairs_u64(vrs_context ctx, vrs_buffer keys_in, vrs_buffer keys_out, vrs_buffer values_in,
                                  vrs_buffer values_out, vrs_buffer histograms, const vrs_push_constants *pc) {
    return run_sort_stage(ctx, keys_in, keys_out, values_in, values_out, histograms, pc, true, 8);
}

int vrs_multi_radixsort(vrs_context ctx, vrs_buffer keys_in, vrs_buffer keys_out, vrs_buffer histograms,
                        const vrs_push_constants *pc) {
    return run_sort_stage(ctx, keys_in, keys_out, nullptr, nullptr, histograms, pc, false);
}

int vrs_multi_radixsort_pairs(vrs_context ctx, vrs_buffer keys_in, vrs_buffer keys_out, vrs_buffer values_in,
                              vrs_buffer values_out, vrs_buffer histograms, const vrs_push_constants *pc) {
    return run_sort_stage(ctx, keys_in, keys_out, values_in, values_out, histograms, pc, true);
}

int vrs_queue_wait_idle(vrs_context ctx) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    int rc = settle_pending(ctx);
    if (rc) return rc;
    VRS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VRS_OK;
}

int vrs_single_radixsort(vrs_context ctx, vrs_buffer buffer0, vrs_buffer buffer1, uint32_t g_num_elements) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    if (g_num_elements == 0) return VRS_OK;
    const size_t bytes = static_cast<size_t>(g_num_elements) * sizeof(uint32_t);
    int rc = check_buffer(ctx, buffer0, bytes, "buffer0");
    if (rc) return rc;
    if ((rc = check_buffer(ctx, buffer1, bytes, "buffer1"))) return rc;
    if (buffer0->ptr == buffer1->ptr) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "buffer0 and buffer1 alias");
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    if (const int settled = settle_pending(ctx)) return settled;  // an async one-call sort may still owe its second half
    vrs::LaunchEvents ev;
    if ((rc = profile_events(ctx, VRS_KERNEL_SINGLE, &ev))) return rc;
    VRS_HIP(ctx, vrs::launch_single(ctx->stream, static_cast<uint32_t *>(buffer0->ptr),
                                    static_cast<uint32_t *>(buffer1->ptr), g_num_elements, ev));
    return VRS_OK;
}

static int ensure_sort_hist(vrs_context ctx, uint32_t workgroups) {
    const size_t need = static_cast<size_t>(workgroups) * VRS_RADIX_SORT_BINS * sizeof(uint32_t);
    if (!ctx->sort_hist || ctx->sort_hist->size < need) {
        if (ctx->sort_hist) {
            VRS_HIP(ctx, hipStreamSynchronize(ctx->stream));
            vrs_buffer_release(ctx->sort_hist);
            ctx->sort_hist = nullptr;
        }
        int rc = vrs_buffer_create(ctx, need, &ctx->sort_hist);
        if (rc) return rc;
    }
    return VRS_OK;
}

// one contract pass (stage 0 + stage 1) of the one-call forms
static int contract_pass(vrs_context ctx, vrs_buffer kin, vrs_buffer kout, vrs_buffer vin, vrs_buffer vout,
                         vrs_push_constants *pc, uint32_t shift, int key_bytes) {
    pc->g_shift = shift;
    int rc = run_histogram_stage(ctx, kin, ctx->sort_hist, pc, key_bytes);
    if (rc) return rc;
    return run_sort_stage(ctx, kin, kout, vin, vout, ctx->sort_hist, pc, vin != nullptr, key_bytes);
}

// The plan kernel writes the head of the plan straight into pinned host memory and stamps it last; wait for the stamp.
// The plan is at most a counting read behind whatever the stream still has to run: a short spin (the usual case: it is
// there already, or microseconds away), then the thread yields between looks, sleeping a little longer each time, and asks
// the stream now and then so that a faulted queue surfaces as an error instead of an endless wait.  Bounded in time
// (VRS_TUNE_PLAN_WAIT_MS, default 60 s): a stream stuck behind work that never finishes returns VRS_ERROR_TIMEOUT.
static int wait_for_host_word(vrs_context ctx, const std::function<bool()> &arrived, bool *never = nullptr) {
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

static int wait_for_plan(vrs_context ctx, uint32_t stamp) {
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
static vrs_buffer_t stack_view(vrs_context ctx, void *ptr, size_t bytes) {
    vrs_buffer_t b;
    b.ctx = ctx;
    b.device = ctx->device;
    b.ptr = ptr;
    b.size = bytes;
    b.owned = false;
    return b;
}

struct OneReadGeometry {
    uint32_t G, T, tiles_total, group_len, tiles0, tile_cap, blind_cap, tiles_b_cap, local_cap;
    size_t rows;
    vrs::StreamCuts cuts0;
};

// everything here is a function of (n, key type, payload or not, the form) alone: both halves compute the same
static OneReadGeometry one_read_geometry(vrs_context ctx, const vrs_context_t::OneRead &st) {
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

static int one_read_scratch(vrs_context ctx, const vrs_context_t::OneRead &st, const OneReadGeometry &g) {
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

static int one_read_lookback_pass(vrs_context ctx, vrs_context_t::OneRead &st, uint32_t i, uint32_t shift, uint32_t grid_tiles, bool forced) {
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
static bool reserves(vrs_context ctx, uint32_t n, bool pairs) {
    (void)n;  // (measured from 1.5e7 to 1e8 keys: 2 to 5 % of the sort at every size the hybrid form takes)
    return !pairs && ctx->os_reserve != 0;
}
static int reservation_begin(vrs_context ctx) {
    if (!ctx->os_reserve || !ctx->os_msd_plan) return VRS_OK;
    if (ctx->os_cursors_open)
        VRS_HIP(ctx, hipMemsetAsync(reinterpret_cast<char *>(ctx->os_msd_plan) + offsetof(vrs::MsdPlan, cursor_a), 0, vrs::kMsdCursorBytes, ctx->stream));
    ctx->os_cursors_open = true;
    return VRS_OK;
}

// second MSD pass + local sort of the hybrid form: partner -> home, then the buckets in place
static int one_read_hybrid_tail(vrs_context ctx, vrs_context_t::OneRead &st, const OneReadGeometry &g, uint32_t tiles_b, uint32_t max_bucket,
                                bool status_was_clean = false) {
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

// ---- pool form (vrs_msd_pool.hip): the hybrid form of bare uint32 keys without the counting read: 24 bytes per key.
static int one_read_enqueue_pool(vrs_context ctx, const struct OneReadGeometry &g);
constexpr int kPoolNoMemory = -4242;  // (internal: pool_scratch found no room on the device; never leaves the library)


static int one_read_enqueue(vrs_context ctx) {
    vrs_context_t::OneRead &st = ctx->one_read;
    const uint32_t n = st.n;
    const int key_bytes = st.key_bytes;
    const bool wide = key_bytes == 8, pairs = st.vptr[0] != nullptr;
    if (st.group == 0) {
        // Hybrid form (K5b): uint32 keys, with or without uint32 payloads, and bare 64-bit keys, from os_hybrid_min_keys on
        // (default 1.3e7 keys, 2.5e7 pairs, 2e7 64-bit keys: below, the fixed costs of the 16384-bin counting read and of a
        // launch per bucket outweigh the saved pass -- measured crossovers, profiles/labs/r03_hybrid_by_size.txt,
        // r02_hybrid_pairs.txt, r02_hybrid_u64.txt); above
        // about 2.3 * 10^8 uniform keys (2.1 * 10^8 pairs or 64-bit keys) the largest bucket no longer fits a workgroup's LDS and the plan
        // says no.  Its local sort ranks with returning LDS atomics, so the lane-order self-test must have passed.
        // 64-bit keys: the counting read never makes LSD tables (their LSD form counts twice anyway), so a refusal always
        // starts over; after one, only every 16th such sort of the context tries again.
        const uint32_t set = ctx->os_hybrid_min_keys;
        const uint32_t hybrid_min = set == 0u ? (wide ? 20000000u : pairs ? 25000000u : 13000000u)
                                              : (wide ? set / 2u : pairs ? set / 8u * 5u : set);
        bool wide_try = wide;  // (with payloads too: round 5)
        // bare uint32 keys the pool form may take: from ITS threshold on (below the counted form's: its first half costs a sample,
        // not a counting read)
        // (pairs: the stable pool form, from the counted form's threshold on -- it has not been measured below)
        const bool pool_size = !pairs && !wide && !st.no_pool && ctx->os_pool != 0 && reserves(ctx, n, false) && n >= ctx->os_pool_min_keys &&
                               n <= vrs::kPoolMaxKeys && (ctx->os_pool == 2 || ctx->os_pool_skip == 0 || n / 2u > ctx->os_pool_skip_n || n < ctx->os_pool_skip_n / 2u);
        if (wide_try && !st.no_hybrid && ctx->os_wide_refused && (++ctx->os_wide_skipped % 16u) != 0u) wide_try = false;
        st.msd_capable = !st.no_hybrid && (key_bytes == 4 || wide_try) && ctx->os_hybrid && ctx->atomic_rank_verified &&
                         ctx->scatter.atomic_rank && (n >= hybrid_min || pool_size) && n >= (1u << 22) &&
                         static_cast<uint64_t>(n) <= 2ull * vrs::kMsdBucketCount * (wide && pairs ? vrs::msd_local_capacity_pairs_u64(false) : vrs::msd_local_capacity(pairs || wide)) &&
                         (ctx->os_groups == 0 || ctx->os_groups == 8);
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
    if (st.group == 0) {
        // Pool form: bare uint32 keys the hybrid form may take skip the counting read altogether.  A refusal (a sample that
        // misjudged a region, a key range the probe missed, a bucket above the local sort's capacity) costs the first pass, so
        // the default is adaptive: after one, the next 15 such sorts of the context take the counted form.
        // (sizes: a bucket must fit the local sort's larger shape -- uniform keys up to about 2.2e8 --, in every mode: beyond it a
        // refusal is certain; pairs up to pool_max_pairs(), the last size whose fullest uniform bucket fits a pairs shape; from
        // os_pool_min_keys on, 2^22 by default: round 5's one-wave local sort made the form the faster one from its own floor)
        const bool candidate = st.msd_capable && !wide && !st.no_pool && ctx->os_pool != 0 && n >= ctx->os_pool_min_keys && n <= vrs::kPoolMaxKeys &&
                               (pairs ? ctx->os_pool_pairs != 0 && n <= vrs::pool_max_pairs() : reserves(ctx, n, false));
        if (candidate && ctx->os_pool_skip && (n / 2u > ctx->os_pool_skip_n || n < ctx->os_pool_skip_n / 2u)) ctx->os_pool_skip = 0;
        st.pool = candidate && (ctx->os_pool == 2 || ctx->os_pool_skip == 0);
        if (candidate && !st.pool) --ctx->os_pool_skip;
    }
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

// the pool form's scratch: its plan (once), the first pass's overflow regions and the slack buffer (grown when a sort needs more).
// kPoolNoMemory: the device has no room for it (about 1.5 n slots, twice that for pairs) -- not an error of the SORT: whatever was
// allocated is released and the caller takes a form that needs no such scratch (the counted form, the LSD passes).
static void pool_scratch_release(vrs_context ctx, bool payloads_only = false) {
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
static int pool_scratch(vrs_context ctx, uint32_t room, uint32_t slack, bool pairs = false) {
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

static int one_read_enqueue_pool(vrs_context ctx, const OneReadGeometry &g) {
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

// the plan's head has arrived: finish the group; *done = the whole sort is on the stream
static int one_read_complete(vrs_context ctx, bool *done) {
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
static int one_read_settle(vrs_context ctx) {
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

}  // extern "C"
namespace {
int settle_pending(vrs_context ctx) { return ctx->one_read.active && !ctx->one_read_settling ? one_read_settle(ctx) : VRS_OK; }
}  // namespace
extern "C" {

static int sort_one_read(vrs_context ctx, vrs_buffer keys, vrs_buffer keys_tmp, vrs_buffer values, vrs_buffer values_tmp,
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
static int sort_all_passes(vrs_context ctx, vrs_buffer keys, vrs_buffer keys_tmp, vrs_buffer values,
                           vrs_buffer values_tmp, uint32_t n, int key_bytes = 4, uint32_t key_base = 0) {
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
    if (key_bytes == 4 && !values && n <= ctx->single_max_keys) {
        vrs::LaunchEvents ev;
        if ((rc = profile_events(ctx, VRS_KERNEL_SINGLE, &ev))) return rc;
        VRS_HIP(ctx, vrs::launch_single(ctx->stream, static_cast<uint32_t *>(keys->ptr), static_cast<uint32_t *>(keys_tmp->ptr),
                                        n, ev));
        return VRS_OK;
    }
    // the look-back status words carry 28-bit stream counts
    if (ctx->xcc_map_valid && ctx->one_call_min_keys != 0 && n >= ctx->one_call_min_keys && n < (1u << 30))
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

static_assert(VRS_MSD_COUNT_WORDS == vrs::kMsdCountWords && VRS_MSD_SHIFT_WORD == vrs::kMsdBucketCount + 8u * 256u, "the public layout of the count words is the kernels' own");

// ---- the hybrid form in two halves, for callers that move the keys between its two MSD passes (vrs_dist_*: the exchange
// between the GPUs sits there).  Both halves only enqueue.
static int msd_half_setup(vrs_context ctx, uint32_t n, vrs_context_t::OneRead *st, OneReadGeometry *g, uint32_t bucket_hint = 0,
                          uint32_t pass_b_groups = 0) {
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

// the own parts into the holes of the grouped buffer (neighbouring ones merged: device copies are launch-bound below a megabyte)
static int fill_own_holes(vrs_context ctx, vrs_buffer grouped, vrs_buffer own, uint64_t own_offset, uint32_t top_bytes, const uint32_t *counts,
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

int vrs_context_device(vrs_context ctx) { return ctx ? ctx->device : -1; }

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

int vrs_transform_keys(vrs_context ctx, vrs_buffer keys, uint32_t num_elements, int mode) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    if (mode < VRS_KEYS_INT32 || mode > VRS_KEYS_SORTABLE_TO_FLOAT32)
        return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "unknown key transform");
    if (num_elements == 0) return VRS_OK;
    int rc = check_buffer(ctx, keys, static_cast<size_t>(num_elements) * sizeof(uint32_t), "keys");
    if (rc) return rc;
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    if (const int settled = settle_pending(ctx)) return settled;  // an async one-call sort may still owe its second half
    ctx->sub_cache.valid = false;  // a buffer is rewritten: the kept sub-tile table may no longer describe its keys
    VRS_HIP(ctx, vrs::launch_transform_keys(ctx->stream, static_cast<uint32_t *>(keys->ptr), num_elements, mode));
    return VRS_OK;
}

int vrs_verify_keys_u32(vrs_context ctx, vrs_buffer keys, uint32_t num_elements, uint64_t *descents, uint64_t *key_sum,
                        uint64_t *key_mix) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    if (descents) *descents = 0;
    if (key_sum) *key_sum = 0;
    if (key_mix) *key_mix = 0;
    if (num_elements == 0) return VRS_OK;
    int rc = check_buffer(ctx, keys, static_cast<size_t>(num_elements) * sizeof(uint32_t), "keys");
    if (rc) return rc;
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    if (const int settled = settle_pending(ctx)) return settled;  // an async one-call sort may still owe its second half
    unsigned long long *d = nullptr;
    VRS_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&d), 3 * sizeof(unsigned long long)));
    unsigned long long h[3] = {0, 0, 0};
    hipError_t e = hipMemsetAsync(d, 0, sizeof h, ctx->stream);
    if (e == hipSuccess) e = vrs::launch_verify_keys(ctx->stream, static_cast<const uint32_t *>(keys->ptr), num_elements, d);
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail_hip(ctx, "vrs_verify_keys_u32", e);
    if (descents) *descents = h[0];
    if (key_sum) *key_sum = h[1];
    if (key_mix) *key_mix = h[2];
    return VRS_OK;
}

int vrs_profile_enable(vrs_context ctx, int enabled) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    if (const int settled = settle_pending(ctx)) return settled;  // a pending sort's second half counts its events under the mask its first half saw
    ctx->profile_mask = enabled ? (1u << VRS_KERNEL_COUNT) - 1u : 0u;
    return VRS_OK;
}

int vrs_profile_enable_mask(vrs_context ctx, uint32_t kernel_mask) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    if (const int settled = settle_pending(ctx)) return settled;
    ctx->profile_mask = kernel_mask & ((1u << VRS_KERNEL_COUNT) - 1u);
    return VRS_OK;
}

int vrs_profile_reset(vrs_context ctx) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    if (const int settled = settle_pending(ctx)) return settled;  // (its second half rewinds event slots it remembered)
    VRS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto &u : ctx->events_used) u = 0;
    return VRS_OK;
}

int vrs_profile_query(vrs_context ctx, int kernel_id, uint64_t *launches, double *total_ms) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    if (kernel_id < 0 || kernel_id >= VRS_KERNEL_COUNT)
        return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "kernel_id out of range");
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    if (const int settled = settle_pending(ctx)) return settled;
    VRS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double sum = 0.0;
    const size_t used = ctx->events_used[kernel_id];
    for (size_t i = 0; i < used; ++i) {
        float ms = 0.f;
        VRS_HIP(ctx, hipEventElapsedTime(&ms, ctx->events[kernel_id][i].start, ctx->events[kernel_id][i].stop));
        sum += ms;
    }
    if (launches) *launches = used;
    if (total_ms) *total_ms = sum;
    return VRS_OK;
}

int vrs_profile_query_launch(vrs_context ctx, int kernel_id, uint64_t index, double *ms) {
    if (!ctx || !ms) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "context or ms is NULL");
    if (kernel_id < 0 || kernel_id >= VRS_KERNEL_COUNT)
        return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "kernel_id out of range");
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    if (const int settled = settle_pending(ctx)) return settled;
    if (index >= ctx->events_used[kernel_id]) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "no such instrumented launch");
    VRS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float t = 0.f;
    VRS_HIP(ctx, hipEventElapsedTime(&t, ctx->events[kernel_id][index].start, ctx->events[kernel_id][index].stop));
    *ms = t;
    return VRS_OK;
}

int vrs_range_partition(vrs_context ctx, vrs_buffer keys_in, vrs_buffer keys_out, vrs_buffer splitters,
                        uint32_t num_splitters, uint32_t num_elements) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    if (num_splitters > 255) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "at most 255 splitters (256 ranges)");
    const uint32_t n = num_elements;
    if (n == 0) return VRS_OK;
    int rc;
    const size_t keys_size = static_cast<size_t>(n) * sizeof(uint32_t);
    if ((rc = check_buffer(ctx, keys_in, keys_size, "keys_in"))) return rc;
    if ((rc = check_buffer(ctx, keys_out, keys_size, "keys_out"))) return rc;
    if ((rc = check_buffer(ctx, splitters, static_cast<size_t>(num_splitters) * sizeof(uint32_t), "splitters"))) return rc;
    if (keys_in->ptr == keys_out->ptr) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "keys_in and keys_out alias");
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    if (const int settled = settle_pending(ctx)) return settled;  // an async one-call sort may still owe its second half
    const uint32_t B = launch_tile_blocks(4);
    const uint32_t W = vrs_workgroup_count(n, B);
    // the [W][256] bucket-count table lives in the context (same scratch as the sub-tile histograms)
    if (W > ctx->sub_hist_rows) {
        if (ctx->sub_hist) VRS_HIP(ctx, hipFree(ctx->sub_hist));
        ctx->sub_hist = nullptr;
        ctx->sub_hist_rows = 0;
        VRS_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->sub_hist),
                               static_cast<size_t>(W) * VRS_RADIX_SORT_BINS * sizeof(uint32_t)));
        ctx->sub_hist_rows = W;
    }
    ctx->sub_cache.valid = false;
    if ((rc = ensure_scratch(ctx, W))) return rc;
    vrs::LaunchEvents ev;
    if ((rc = profile_events(ctx, VRS_KERNEL_HISTOGRAM, &ev))) return rc;
    VRS_HIP(ctx, vrs::launch_histograms(ctx->stream, keys_in->ptr, ctx->sub_hist, n, 0, W, B, ev, nullptr, 4, splitters->ptr,
                                        num_splitters));
    if ((rc = profile_events(ctx, VRS_KERNEL_PREFIX, &ev))) return rc;
    VRS_HIP(ctx, vrs::launch_prefix(ctx->stream, ctx->sub_hist, ctx->scratch, W, ev));
    ctx->last_offsets_workgroups = W;
    ctx->last_offsets_stride = 1;
    if ((rc = profile_events(ctx, VRS_KERNEL_SCATTER, &ev))) return rc;
    VRS_HIP(ctx, vrs::launch_range_partition(ctx->stream, static_cast<const uint32_t *>(keys_in->ptr),
                                             static_cast<uint32_t *>(keys_out->ptr), ctx->scratch.offsets, n, W,
                                             ctx->xcd_remap, ctx->scatter.atomic_rank,
                                             static_cast<const uint32_t *>(splitters->ptr), num_splitters, ev));
    return VRS_OK;
}

int vrs_multi_radixsort_digit_offsets(vrs_context ctx, void *host_u32x256) {
    if (!ctx || !host_u32x256) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "context or output is NULL");
    if (ctx->last_offsets_workgroups == 0)
        return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "no RADIX_SORT stage has run on this context yet");
    // workgroup 0 has no predecessors, so its offset row IS the global exclusive digit prefix
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    VRS_HIP(ctx, hipMemcpyAsync(host_u32x256, ctx->scratch.offsets, VRS_RADIX_SORT_BINS * sizeof(uint32_t),
                                hipMemcpyDeviceToHost, ctx->stream));
    VRS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VRS_OK;
}

int vrs_multi_radixsort_digit_offsets_device(vrs_context ctx, vrs_buffer out_u32x256) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    if (ctx->last_offsets_workgroups == 0)
        return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "no RADIX_SORT stage has run on this context yet");
    int rc = check_buffer(ctx, out_u32x256, VRS_RADIX_SORT_BINS * sizeof(uint32_t), "digit offsets");
    if (rc) return rc;
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    VRS_HIP(ctx, hipMemcpyAsync(out_u32x256->ptr, ctx->scratch.offsets, VRS_RADIX_SORT_BINS * sizeof(uint32_t),
                                hipMemcpyDeviceToDevice, ctx->stream));
    return VRS_OK;
}

int vrs_multi_radixsort_offsets_hook(vrs_context ctx, vrs_buffer out_u32x256, void *event) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    if (out_u32x256) {
        const int rc = check_buffer(ctx, out_u32x256, VRS_RADIX_SORT_BINS * sizeof(uint32_t), "digit offsets");
        if (rc) return rc;
    }
    ctx->offsets_hook_out = out_u32x256 ? out_u32x256->ptr : nullptr;
    ctx->offsets_hook_event = event;
    return VRS_OK;
}

int vrs_debug_download_offsets(vrs_context ctx, void *host_data, size_t size_bytes) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    const size_t row = VRS_RADIX_SORT_BINS * sizeof(uint32_t);
    const size_t have = static_cast<size_t>(ctx->last_offsets_workgroups) * row;
    if (!host_data || size_bytes > have || size_bytes % row != 0)
        return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "offset table is smaller than the requested size");
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    // sub-tiled launches keep one row per 8192-key sub-tile: the contract tile's row is its first sub-tile's
    VRS_HIP(ctx, hipMemcpy2DAsync(host_data, row, ctx->scratch.offsets, row * ctx->last_offsets_stride, row,
                                  size_bytes / row, hipMemcpyDeviceToHost, ctx->stream));
    VRS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VRS_OK;
}

int vrs_debug_atomic_rank_selftest(vrs_context ctx, uint32_t rounds, uint32_t seed, uint64_t *mismatches) {
    if (!ctx || !mismatches) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "context or mismatches is NULL");
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    return atomic_rank_selftest(ctx, rounds, seed, mismatches);
}

int vrs_one_call_stats(vrs_context ctx, uint64_t *lookback_passes, uint64_t *fallback_passes, uint64_t *skipped_passes) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    if (lookback_passes) *lookback_passes = ctx->os_lookback_passes;
    if (fallback_passes) *fallback_passes = ctx->os_fallback_passes;
    if (skipped_passes) *skipped_passes = ctx->os_skipped_passes;
    return VRS_OK;
}

int vrs_one_call_relaunched_passes(vrs_context ctx, uint64_t *relaunched_passes) {
    if (!ctx || !relaunched_passes) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "context or output is NULL");
    *relaunched_passes = ctx->os_relaunched_passes;
    return VRS_OK;
}

int vrs_one_call_hybrid_sorts(vrs_context ctx, uint64_t *hybrid_sorts) {
    if (!ctx || !hybrid_sorts) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "context or output is NULL");
    *hybrid_sorts = ctx->os_hybrid_sorts;
    return VRS_OK;
}

int vrs_one_call_hybrid_recounts(vrs_context ctx, uint64_t *recounts) {
    if (!ctx || !recounts) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "context or output is NULL");
    *recounts = ctx->os_hybrid_recounts;
    return VRS_OK;
}

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

int vrs_debug_xcc_placement(vrs_context ctx, uint64_t *reprobes, uint64_t *xcc_map, int *valid) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    if (reprobes) *reprobes = ctx->reprobes;
    if (xcc_map) *xcc_map = ctx->xcc_map;
    if (valid) *valid = ctx->xcc_map_valid ? 1 : 0;
    return VRS_OK;
}

int vrs_rank_mode(vrs_context ctx) { return ctx && ctx->scatter.atomic_rank ? 2 : 1; }

int vrs_set_tuning(vrs_context ctx, int key, int value) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    if (key != VRS_TUNE_PLAN_WAIT_MS) {  // the second half of a pending sort must see the settings its first half saw
        const int rc = settle_pending(ctx);
        if (rc) return rc;
    }
    switch (key) {
        case VRS_TUNE_XCD_REMAP:
            ctx->xcd_remap = value != 0;
            return VRS_OK;
        case VRS_TUNE_SCATTER_VARIANT:
            ctx->scatter.variant = value;
            return VRS_OK;
        case VRS_TUNE_FUSED_PREFIX:
            ctx->fused_prefix = value != 0;
            return VRS_OK;
        case VRS_TUNE_RANK_MODE: {
            if (value == 1) {
                ctx->scatter.atomic_rank = false;
            } else if (value == 2) {
                ctx->scatter.atomic_rank = true;
            } else if (value == 0) {
                ctx->scatter.atomic_rank = ctx->atomic_rank_verified;
            } else {
                return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "rank mode must be 0 (auto), 1 (ballot) or 2 (atomic)");
            }
            return VRS_OK;
        }
        case VRS_TUNE_ONE_CALL_MIN_KEYS:
            if (value < 0) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "one-call threshold must be >= 0");
            ctx->one_call_min_keys = static_cast<uint32_t>(value);
            return VRS_OK;
        case VRS_TUNE_DEBUG_MISPLACE_STREAMS:
            ctx->os_misplace = value != 0;
            return VRS_OK;
        case VRS_TUNE_HYBRID:
            ctx->os_hybrid = value != 0;
            ctx->os_wide_refused = false;  // 64-bit keys: forget an earlier refusal
            ctx->os_wide_skipped = 0;
            return VRS_OK;
        case VRS_TUNE_HYBRID_MIN_KEYS:
            if (value < 0) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "hybrid threshold must be >= 0");
            ctx->os_hybrid_min_keys = static_cast<uint32_t>(value);
            return VRS_OK;
        case VRS_TUNE_HYBRID_FAST_COUNT:
            if (value < 0 || value > 2) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "fast count mode must be 0, 1 or 2");
            ctx->os_fast_count = value;
            ctx->os_fast_count_armed[0] = ctx->os_fast_count_armed[1] = false;
            return VRS_OK;
        case VRS_TUNE_FUSED_PLAN:
            ctx->os_fused_plan = value != 0;
            return VRS_OK;
        case VRS_TUNE_SINGLE_MAX_KEYS:
            if (value < 0) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "single-launch threshold must be >= 0");
            ctx->single_max_keys = static_cast<uint32_t>(value);
            return VRS_OK;
        case VRS_TUNE_LOOKBACK_SPIN_BUDGET:
            if (value < 0) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "spin budget must be >= 0");
            ctx->os_spin_budget = static_cast<uint32_t>(value);
            return VRS_OK;
        case VRS_TUNE_DEBUG_HOLD_TILE:
            ctx->os_hold_tile = value;
            return VRS_OK;
        case VRS_TUNE_ASYNC_SORT:
            ctx->os_async = value != 0;
            return VRS_OK;
        case VRS_TUNE_PLAN_WAIT_MS:
            if (value < 0) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "the plan wait limit must be >= 0 ms");
            ctx->os_plan_wait_ms = static_cast<uint32_t>(value);
            return VRS_OK;
        case VRS_TUNE_MSD_RESERVE:
            if (value < 0 || value > 2) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "MSD reservation must be 0 (never) or 1 / 2 (bare keys of the hybrid form)");
            ctx->os_reserve = value;
            return VRS_OK;
        case VRS_TUNE_MSD_POOL:
            if (value < 0 || value > 2) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "the pool form must be 0 (never), 1 (adaptive) or 2 (always tried)");
            ctx->os_pool = value;
            ctx->os_pool_skip = 0;
            return VRS_OK;
        case VRS_TUNE_DEBUG_XCC_STRAY_BLOCK:
            if (value >= 4096) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "the probe has 4096 blocks (a negative value probes again without a stray one)");
            return probe_xcc_map(ctx, value);  // (a negative value probes again as at creation; a pending sort was settled above)
        case VRS_TUNE_DEBUG_XCC_ROTATE: {
            if (value < 0 || value > 7) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "rotate the probed placement by 0 .. 7 places");
            if (const int rc = probe_xcc_map(ctx)) return rc;
            const unsigned sh = 8u * static_cast<unsigned>(value);
            if (sh) ctx->xcc_map = (ctx->xcc_map >> sh) | (ctx->xcc_map << (64u - sh));
            return VRS_OK;
        }
        case VRS_TUNE_MSD_POOL_TOP_BITS:
            if (value < 6 || value > 8) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "the pool form's first pass sorts by 6, 7 (the default) or 8 bits");
            ctx->os_pool_top_bits = value;
            ctx->os_pool_layout_valid = false;
            return VRS_OK;
        case VRS_TUNE_MSD_POOL_PAIRS:
            ctx->os_pool_pairs = value != 0 ? 1 : 0;
            return VRS_OK;
        case VRS_TUNE_DEBUG_POOL_NO_MEMORY:
            if (value < 0) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "allocations left to fail: 0 or more");
            ctx->os_pool_fail_alloc = static_cast<uint32_t>(value);
            return VRS_OK;
        case VRS_TUNE_MSD_POOL_REUSE_LAYOUT:
            ctx->os_pool_reuse = value != 0;
            ctx->os_pool_reuse_rooms = value == 1;
            ctx->os_pool_layout_valid = false;
            ctx->os_pool_stale_run = ctx->os_pool_reuse_pause = 0;
            return VRS_OK;
        case VRS_TUNE_MSD_POOL_SUB_BITS:
            if (value != 0 && (value < 6 || value > 8)) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "the pool form's second pass sorts by 6, 7 or 8 bits (0 = by size)");
            ctx->os_pool_sub_bits = value;
            ctx->os_pool_layout_valid = false;
            return VRS_OK;
        case VRS_TUNE_MSD_POOL_MIN_KEYS:
            if (value < (1 << 22)) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "the pool form takes 2^22 keys or more");
            ctx->os_pool_min_keys = static_cast<uint32_t>(value);
            return VRS_OK;
        case VRS_TUNE_DIGIT_TABLE_GROUPS:
            if (value != 0 && value != 8 && value != 16 && value != 32)
                return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "digit table groups must be 0 (by size), 8, 16 or 32");
            if (static_cast<uint32_t>(value) % vrs::kStreams != 0)
                return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "digit table groups must be a multiple of the stream count");
            ctx->os_groups = static_cast<uint32_t>(value);
            return VRS_OK;
        default:
            return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "unknown tuning key");
    }
}

}  // extern "C"
