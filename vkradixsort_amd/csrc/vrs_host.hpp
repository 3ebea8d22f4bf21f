// vrs_host.hpp -- what the parts of the C ABI's implementation (vrs_capi*.hip) share: the context and buffer objects, error helpers,
// and the prototypes of the host functions that cross a file boundary.  Internal: never installed, never included by a caller.
#pragma once
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
    int os_pool_pairs_packed = -1;         // VRS_TUNE_MSD_POOL_PAIRS_PACKED: the pairs' local sort in its packed form: -1 by size, 0 never, 1 always
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


namespace vrsh {


// the kernels' own launch tile: 8192 uint32 keys or 4096 uint64 keys (32 KiB either way)
constexpr uint32_t launch_tile_blocks(int key_bytes) { return key_bytes == 8 ? 16u : 32u; }
extern thread_local std::string g_global_error;
// Contexts that are alive (vrs_context_create* ... vrs_context_destroy): a buffer may outlive its context (host-language finalisers
// run in any order), so vrs_buffer_release asks here before it touches buf->ctx.
extern std::mutex g_live_mutex;
extern std::set<vrs_context> g_live_contexts;


#define VRS_HIP(ctx, call)                                       \
    do {                                                         \
        hipError_t e__ = (call);                                 \
        if (e__ != hipSuccess) return fail_hip((ctx), #call, e__); \
    } while (0)


struct OneReadGeometry {
    uint32_t G, T, tiles_total, group_len, tiles0, tile_cap, blind_cap, tiles_b_cap, local_cap;
    size_t rows;
    vrs::StreamCuts cuts0;
};

constexpr int kPoolNoMemory = -4242;  // (internal: pool_scratch found no room on the device; never leaves the library)

int fail(vrs_context ctx, int code, const std::string &msg);
int fail_hip(vrs_context ctx, const char *what, hipError_t e);
int ensure_scratch(vrs_context ctx, uint32_t W);
int profile_events(vrs_context ctx, int id, vrs::LaunchEvents *ev);
int check_push_constants(vrs_context ctx, const vrs_push_constants *pc, int key_bytes = 4);
int check_buffer(vrs_context ctx, vrs_buffer b, size_t need, const char *name);
int atomic_rank_selftest(vrs_context ctx, uint32_t rounds, uint32_t seed, uint64_t *mismatches);
int probe_xcc_map(vrs_context ctx, int stray_block = -1);
uint32_t *drift_word(vrs_context ctx);
int reprobe_if_drifted(vrs_context ctx);
int create_context(int device_ordinal, hipStream_t borrowed, bool borrow, vrs_context *out_ctx);
int run_sort_stage(vrs_context ctx, vrs_buffer keys_in, vrs_buffer keys_out, vrs_buffer values_in,
                   vrs_buffer values_out, vrs_buffer histograms, const vrs_push_constants *pc, bool pairs,
                   int key_bytes = 4);
int run_histogram_stage(vrs_context ctx, vrs_buffer keys_in, vrs_buffer histograms,
                               const vrs_push_constants *pc, int key_bytes);
int ensure_sort_hist(vrs_context ctx, uint32_t workgroups);
int contract_pass(vrs_context ctx, vrs_buffer kin, vrs_buffer kout, vrs_buffer vin, vrs_buffer vout,
                         vrs_push_constants *pc, uint32_t shift, int key_bytes);
int wait_for_host_word(vrs_context ctx, const std::function<bool()> &arrived, bool *never = nullptr);
int wait_for_plan(vrs_context ctx, uint32_t stamp);
vrs_buffer_t stack_view(vrs_context ctx, void *ptr, size_t bytes);
OneReadGeometry one_read_geometry(vrs_context ctx, const vrs_context_t::OneRead &st);
int one_read_scratch(vrs_context ctx, const vrs_context_t::OneRead &st, const OneReadGeometry &g);
int one_read_lookback_pass(vrs_context ctx, vrs_context_t::OneRead &st, uint32_t i, uint32_t shift, uint32_t grid_tiles, bool forced);
bool reserves(vrs_context ctx, uint32_t n, bool pairs);
int reservation_begin(vrs_context ctx);
int one_read_hybrid_tail(vrs_context ctx, vrs_context_t::OneRead &st, const OneReadGeometry &g, uint32_t tiles_b, uint32_t max_bucket,
                                bool status_was_clean = false);
int one_read_enqueue(vrs_context ctx);
void pool_scratch_release(vrs_context ctx, bool payloads_only = false);
int pool_scratch(vrs_context ctx, uint32_t room, uint32_t slack, bool pairs = false);
int one_read_enqueue_pool(vrs_context ctx, const OneReadGeometry &g);
int one_read_complete(vrs_context ctx, bool *done);
int one_read_settle(vrs_context ctx);
int settle_pending(vrs_context ctx);
int sort_one_read(vrs_context ctx, vrs_buffer keys, vrs_buffer keys_tmp, vrs_buffer values, vrs_buffer values_tmp,
                         uint32_t n, int key_bytes, uint32_t key_base);
int sort_all_passes(vrs_context ctx, vrs_buffer keys, vrs_buffer keys_tmp, vrs_buffer values,
                           vrs_buffer values_tmp, uint32_t n, int key_bytes = 4, uint32_t key_base = 0);
int msd_half_setup(vrs_context ctx, uint32_t n, vrs_context_t::OneRead *st, OneReadGeometry *g, uint32_t bucket_hint = 0,
                          uint32_t pass_b_groups = 0);
int fill_own_holes(vrs_context ctx, vrs_buffer grouped, vrs_buffer own, uint64_t own_offset, uint32_t top_bytes, const uint32_t *counts,
                          const uint32_t *own_counts);

}  // namespace vrsh
