// vrs_kernels.h -- launch wrappers of the gfx950 kernels (internal; the public surface is
// include/vkradixsort_amd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace vrs {

// largest power of two <= x (x >= 1)
inline uint32_t floor_pow2(uint32_t x) {
    uint32_t p = 1;
    while (p * 2u <= x) p *= 2u;
    return p;
}

// Scratch owned by the context: offsets[W*256] and chunk_sums[G*256].
struct PrefixScratch {
    uint32_t *offsets = nullptr;
    uint32_t *chunk_sums = nullptr;
    // fused single-launch prefix: [G][256] {epoch, value} granules, zero-initialised; nullptr = two-launch form
    unsigned long long *granules = nullptr;
    uint32_t fused_max_chunks = 0;  // largest G the fused form may be used for (<= compute units, <= allocation)
    uint32_t epoch = 0;             // bumped by the caller before every launch_prefix; never 0 when used
};

// Optional timing events attached to a launch's own dispatch packet (no extra barrier packets).
struct LaunchEvents {
    hipEvent_t start = nullptr;
    hipEvent_t stop = nullptr;
};

// How the scatter is launched (performance only).
struct ScatterLaunch {
    int variant = 0;          // 0 = choose from B; else OCC*100000 + ITEMS*1000 + WAVES*10 + RANK
    bool atomic_rank = false; // rank with returning LDS atomics (only after the device self-test passed)
    int wgs_per_cu = 0;     // 0 = occupancy API
    int compute_units = 256;
};

// tiles per chunk for the two-level prefix: smallest power of two C with C*C >= W
uint32_t prefix_chunk_tiles(uint32_t num_workgroups);

// keys are uint32 (key_bytes == 4) or uint64 (key_bytes == 8)
hipError_t launch_histograms(hipStream_t stream, const void *keys_in, uint32_t *hist, uint32_t n, uint32_t shift,
                             uint32_t W, uint32_t B, LaunchEvents ev = {}, const uint32_t *tile_order = nullptr,
                             int key_bytes = 4, const void *splitters = nullptr, uint32_t num_splitters = 0);

// range partition of uint32 keys (8192-key tiles, B = 32): stable scatter by bucket = #splitters <= key
hipError_t launch_range_partition(hipStream_t stream, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *offsets,
                                  uint32_t n, uint32_t W, bool xcd_remap, bool atomic_rank, const uint32_t *splitters,
                                  uint32_t num_splitters, LaunchEvents ev = {});

hipError_t launch_prefix(hipStream_t stream, const uint32_t *hist, const PrefixScratch &scratch,
                         uint32_t W, LaunchEvents ev = {});

hipError_t launch_scatter(hipStream_t stream, const void *keys_in, void *keys_out, const uint32_t *values_in,
                          uint32_t *values_out, const uint32_t *offsets, uint32_t n, uint32_t shift, uint32_t W,
                          uint32_t B, bool xcd_remap, const ScatterLaunch &cfg, LaunchEvents ev = {},
                          const uint32_t *tile_order = nullptr, uint32_t offset_row_stride = 1, int key_bytes = 4);

// hist[w][d] = sum of sub rows [w*S, (w+1)*S): contract table from the 8192-key sub-tile table
hipError_t launch_fold_histograms(hipStream_t stream, const uint32_t *sub, uint32_t *hist, uint32_t sub_rows,
                                  uint32_t W, uint32_t S, LaunchEvents ev = {});

hipError_t launch_atomic_rank_selftest(hipStream_t stream, uint32_t rounds, uint32_t seed,
                                       unsigned long long *mismatches);

hipError_t launch_transform_keys(hipStream_t stream, uint32_t *keys, uint32_t n, int mode);

// out3 (zeroed by the caller): descents keys[i] > keys[i+1], sum of the keys, sum of a 64-bit mix of every key
hipError_t launch_verify_keys(hipStream_t stream, const uint32_t *keys, uint32_t n, unsigned long long *out3);

// ---- one-call sort for large N (K5): one counting read, four look-back scatter passes
// The counting read sorts every key into one of `groups` GROUPS per pass (pass 0: slice of the input, pass p > 0:
// digit p-1 / (256 / groups)); plan_kernel merges neighbouring groups into kStreams balanced STREAMS, one per XCD
// (fewer streams = fewer open write fronts per pass: 2048 instead of 8192, which is what the next pass's write drain
// pays for).  groups is 8, 16 or 32 (a tuning choice: more groups = streams that follow skewed data better, fewer =
// a cheaper counting read).
constexpr int kMaxGroups = 32;
constexpr int kStreams = 8;             // a multiple of 8 that divides the group count: stream s runs on XCD s % 8
// one stream of one pass: a contiguous range of the pass's input
struct StreamDesc {
    uint32_t start;        // first key of the stream in the pass's input
    uint32_t len;
    uint32_t first_group;  // the stream starts with this group: its seed row is group_seed[p][that]
    uint32_t tiles;        // ceil(len / tile)
};
// what the plan says about one pass
enum : uint32_t { kPassLookback = 0, kPassIdentity = 1, kPassUnbalanced = 2, kPassLookbackWide = 3 };
// The part of the plan the scatter workgroups read (scalar loads) and the host reads back.  The host enqueues all
// four look-back passes BEFORE it knows the plan; a pass p >= first_abnormal leaves at once (its workgroups read the
// word and exit) and the host, once the head has arrived, enqueues passes first_abnormal..3 in the form they need.
struct OnesweepPlanHead {
    StreamDesc stream[4][kStreams];
    StreamDesc blind[4][kStreams];  // the same with tiles = 0 for every pass >= first_abnormal
    uint32_t max_tiles[4];      // tiles of the longest stream of each pass
    uint32_t mode[4];           // kPassLookback / kPassIdentity (one digit value holds every key) / kPassUnbalanced (a
                                // stream longer than tile_cap tiles: contract pass) / kPassLookbackWide (a stream longer
                                // than the speculative grid but within tile_cap: look-back pass, launched again)
    uint32_t first_abnormal;    // smallest p with mode[p] != kPassLookback, 4 if there is none
    uint32_t msd_ok;            // hybrid form (K5b): 1 = the MSD passes and the local sort take over (set by msd_plan_kernel)
    uint32_t msd_tiles_b;       // rows of workgroups of the second MSD pass
    uint32_t msd_max_bucket;    // keys in the largest top-14-bit bucket
    uint32_t msd_shift_a;       // the first MSD pass's digit shift (top 8 bits of the key range)
    uint32_t lsd_missing;       // 1 = the counting read counted only the bucket histogram (fast count): no LSD plan exists
    uint32_t msd_counted;       // (the first MSD pass's plan) 1 = the bucket histogram holds every key: the probed range was 27-32 bits wide and no key lay outside it
    uint32_t drift;             // host copy only: workgroups that found themselves on another XCC than the context's probe said (report_drift)
    uint32_t ready;             // host copy only: the sort's stamp, written after everything else
};
struct OnesweepPlan {
    OnesweepPlanHead head;
    uint32_t group_seed[4][kMaxGroups + 1][256];  // global offset of digit d at the start of group g of pass p
};
constexpr size_t kDigitTableWords = 4u * kMaxGroups * 256u;  // [4][groups][256], zero between sorts
// first group of every stream (first_group[kStreams] == groups)
struct StreamCuts {
    uint32_t first_group[kStreams + 1];
};
// Cut k (0 < k < kStreams) between streams k-1 and k: the group boundary closest to k * n / kStreams, where group g
// of the pass's input is keys [start_of(g), start_of(g + 1)) and start_of(groups) == n.  Non-decreasing in k.  One
// definition for the host (pass 0) and the plan kernel (passes 1-3).
template <typename StartOf>
__host__ __device__ inline uint32_t balanced_cut(const StartOf &start_of, uint32_t n, uint32_t k, uint32_t groups) {
    const uint64_t target = static_cast<uint64_t>(n) * k / kStreams;
    uint32_t g = 0;
    while (g < groups && start_of(g) < target) ++g;
    if (g > 0 && target - start_of(g - 1) < start_of(g) - target) --g;  // the boundary before is closer
    return g;
}
// pass 0: the groups are slices of group_len keys
StreamCuts pass0_stream_cuts(uint32_t n, uint32_t group_len, uint32_t groups);
// keys per look-back tile: 8192 uint32 or 4096 uint64 (32 KiB either way)
uint32_t onesweep_tile_keys(int key_bytes);
// Look-back status words: ONE region of kStreams * tile_cap rows of 256 words serves all four passes of a group --
// a word carries the pass's tag, so only the counting read zeroes it (once per group of four passes).
// Counts the four digits at bits [base_shift, base_shift + 32) of every key into tables[4][groups][256]; also zeroes
// status[0, status_words) (a multiple of 4 words, 16-byte aligned).  group_len: keys per pass-0 group (whole tiles).
// fused: the counting read's last workgroup also makes the plan (what launch_plan would do in a launch of its own);
// done = a zero-initialised ticket word the launches share
struct FusedPlan {
    OnesweepPlan *plan;
    OnesweepPlanHead *host_head;
    uint32_t *done;
    uint32_t stamp, tile, tile_cap, blind_cap;
    StreamCuts cuts0;
};
hipError_t launch_digit_tables(hipStream_t stream, const void *keys, uint32_t n, int key_bytes, uint32_t base_shift,
                               uint32_t group_len, uint32_t groups, uint32_t *tables, uint32_t *status,
                               size_t status_words, int compute_units, LaunchEvents ev = {},
                               const FusedPlan *fused = nullptr);
// host_head: device-visible address of a pinned host copy of the head (written with system-scope stores, `stamp` last)
// blind_cap: rows of workgroups of the speculatively enqueued passes 1-3 (<= tile_cap)
hipError_t launch_plan(hipStream_t stream, uint32_t *tables, OnesweepPlan *plan, OnesweepPlanHead *host_head,
                       uint32_t stamp, uint32_t n, uint32_t group_len, uint32_t groups, uint32_t tile, uint32_t tile_cap,
                       uint32_t blind_cap, const StreamCuts &cuts0);
// pass = 0..3 inside the group of four the plan was made for, shift = the pass's absolute bit position; the streams
// come from plan->head (device memory).  grid_tiles: rows of workgroups to launch (>= the pass's max_tiles, which the
// host may not know yet: tile_cap).  forced: 1 = run even if the plan marks an earlier pass abnormal (the host's second
// enqueue); 2 = the same, unless the plan's counts are void (OnesweepPlanHead::msd_counted == 0: the workgroups leave at once).  status: kStreams * grid_tiles rows of 256 tagged words.  spin_budget: polls of an unpublished row before
// a tile stops waiting and counts its stream's earlier keys itself; hold_tile >= 0: test hook, that tile of every
// stream never publishes.
hipError_t launch_onesweep_scatter(hipStream_t stream, const void *keys_in, void *keys_out, const uint32_t *values_in,
                                   uint32_t *values_out, const OnesweepPlan *plan, uint32_t pass, uint32_t shift,
                                   uint32_t *status, uint32_t grid_tiles, int forced, bool atomic_rank,
                                   unsigned long long xcc_map, int key_bytes, uint32_t spin_budget, int hold_tile,
                                   LaunchEvents ev = {}, bool misplace = false, uint32_t key_base = 0, struct MsdPlan *reserve = nullptr,
                                   uint32_t *drift = nullptr);
// drift: a word of pinned host memory (device view) the first blocks add to when they find themselves on another XCC than xcc_map says
// (report_drift, vrs_device.hpp), or nullptr
// ---- hybrid form of the one-call sort (K5b, uint32 keys): MSD partition by the top 14 bits in two look-back passes,
// then one workgroup per bucket sorts the low 18 bits inside LDS.
constexpr uint32_t kMsdBucketCount = 1u << 14;
struct MsdPlan {
    uint32_t shift;                       // bucket = key >> shift (the top 14 bits of the key range)
    uint32_t ok;                          // 1 = the plan took the hybrid form; 0 = a second MSD pass / local sort enqueued before the plan was known leaves at once
    uint32_t sub_bits;                    // bucket bits the second MSD pass sorts by (6: a whole sort; up to 8: vrs_msd_finish_grouped_u32)
    uint32_t pad;
    uint32_t xcd_tiles[8][33];            // XCD x walks top-byte buckets x, x+8, ...: exclusive prefix of their tile counts
    // Reservation (MSD passes over BARE keys, where the order inside a digit's range is free): a tile takes its place in the range
    // of (stream or bucket group, digit) with ONE atomic add on that range's counter -- in the L2 of the XCD all tiles of the
    // stream run on -- instead of looking back over its predecessors.  A tile that finds itself on another XCD takes its keys' room from
    // the END of the range with a device-wide atomic on a second counter: the two never meet, the range has room for exactly all.
    // The counters count keys from zero; the local sort -- the last kernel of the form -- leaves them zero for the next sort.
    uint32_t cursor_a[kStreams][256];     // first MSD pass: keys of (stream, top byte) placed so far
    uint32_t back_a[kStreams][256];       //   keys taken from the range's end (zero unless a tile ran off its stream's XCD)
    uint32_t cursor_b[kMsdBucketCount];   // second MSD pass: keys of bucket b placed so far
    uint32_t back_b[kMsdBucketCount];
    uint32_t base[kMsdBucketCount + 1];   // exclusive prefix of the bucket sizes = where bucket b starts when sorted
};
static_assert(offsetof(MsdPlan, base) % 16 == 0, "the plan kernel stores the bucket starts as 16-byte vectors");
constexpr size_t kMsdCursorBytes = offsetof(MsdPlan, base) - offsetof(MsdPlan, cursor_a);  // the four counter arrays, contiguous
static_assert(offsetof(MsdPlan, back_a) == offsetof(MsdPlan, cursor_a) + sizeof(uint32_t) * kStreams * 256 &&
                  offsetof(MsdPlan, cursor_b) == offsetof(MsdPlan, back_a) + sizeof(uint32_t) * kStreams * 256 &&
                  offsetof(MsdPlan, back_b) == offsetof(MsdPlan, cursor_b) + sizeof(uint32_t) * kMsdBucketCount,
              "rearm_reservation walks cursor_a, back_a, cursor_b, back_b as one block");
// words the hybrid's counting needs beside the digit tables: the 16384-bin histogram + 8 x 256 top-byte counts per
// pass-0 group, zero between sorts
constexpr size_t kMsdCountWords = kMsdBucketCount + 8u * 256u + 64u;  // + the probed shift and the out-of-range flag
constexpr uint32_t kShiftFromPlan = 0xFFFFFFFFu;  // launch_onesweep_scatter: take the shift from plan->head.msd_shift_a

// same as launch_digit_tables with 8 groups, and fills msd_counts (uint32 keys only); msd_only (fast count): a key range
// the hybrid form can take gets ONLY the bucket histogram -- launch_msd_plan must be told the same
hipError_t launch_digit_tables_msd(hipStream_t stream, const void *keys, uint32_t n, uint32_t group_len, uint32_t *tables,
                                   uint32_t *status, size_t status_words, int compute_units, uint32_t *msd_counts,
                                   bool msd_only, LaunchEvents ev = {}, uint32_t key_base = 0, uint32_t force_shift = 0);
// force_shift != 0: no range probe, bucket = (key - key_base) >> force_shift (a caller that knows the range: keys grouped by
// top byte, vrs_msd_finish_grouped_u32)
// ONE workgroup: the plan of the four LSD passes (what launch_plan does, 8 groups), then the hybrid form's: bucket
// offsets, the first MSD pass's seeds (into plan_a->group_seed[0]) and streams, the second pass's tile tables; decides
// msd_ok (key range 27-32 bits and fully probed, largest bucket <= the local sort's capacity, XCD tile counts <=
// tiles_b_cap), arms exactly one of the two speculative first passes (plan_a's or plan_lsd's blind descriptors), writes the
// host head and stamps it.  max_shift: the most low bits the local sort takes (18; 64-bit keys: 50).  msd_only (1: as the
// counting read was told; 2: 64-bit keys, never any tables): without LSD tables there is no LSD plan -- if msd_ok is 0 then, head.lsd_missing is 1
// and neither first pass is armed (the caller counts again, for the LSD passes)
hipError_t launch_msd_plan(hipStream_t stream, uint32_t *msd_counts, MsdPlan *msd, OnesweepPlan *plan_a,
                           OnesweepPlan *plan_lsd, OnesweepPlanHead *host_head, uint32_t stamp, uint32_t n, uint32_t tile,
                           uint32_t tiles_b_cap, uint32_t local_cap, uint32_t *tables, uint32_t group_len, uint32_t tile_cap,
                           uint32_t blind_cap, const StreamCuts &cuts0, uint32_t msd_only, uint32_t max_shift,
                           uint32_t *host_log = nullptr, uint32_t sub_bits = 6);
// sub_bits: the low bits of the 14-bit bucket index the second MSD pass sorts by; the input of that pass is grouped by the
// remaining 14 - sub_bits high bits (a whole sort: 8 + 6, pass A's digit + pass B's)
// host_log: kMsdLogWords words of pinned host memory (device view) or nullptr; word stamp % kMsdLogWords receives
// (stamp << 1) | msd_ok before the head's stamp
constexpr uint32_t kMsdLogWords = 32;
// second MSD pass: bits [18, 24) inside every top-byte bucket; grid of 8 * tiles_b workgroups; status rows: 8 * tiles_b
// values_in / values_out: uint32 payloads that follow their keys (nullptr: keys only)
hipError_t launch_msd_pass_b(hipStream_t stream, const void *keys_in, void *keys_out, const uint32_t *values_in,
                             uint32_t *values_out, MsdPlan *msd, uint32_t *status, uint32_t tiles_b, bool atomic_rank,
                             unsigned long long xcc_map, int key_bytes, uint32_t spin_budget, LaunchEvents ev = {}, uint32_t key_base = 0, uint32_t sub_bits = 6,
                             bool reserve = false, uint32_t *drift = nullptr);
// key_base (uint32 keys of the hybrid form only): the caller promises keys >= key_base (a multiple of 2^24); buckets and MSD
// digits are taken from key - key_base, so a sub-range of the key space gets the same 16384 buckets a full range would;
// a key below it makes the counting read flag the sort and the plan refuse the hybrid form
// 64-bit keys: the counting read of the hybrid form (bucket histogram + top-byte counts of the 8 input slices only; zeroes
// the status words) and the local sort of every bucket by its low msd->shift bits (ceil(shift / 9) LDS passes)
hipError_t launch_msd_count_u64(hipStream_t stream, const void *keys, uint32_t n, uint32_t group_len, uint32_t *status,
                                size_t status_words, int compute_units, uint32_t *msd_counts, LaunchEvents ev = {});
// clear_status / clear_words (a multiple of 4): look-back status words the kernel clears on the side (for the next sort), or nullptr
hipError_t launch_msd_local_sort_u64(hipStream_t stream, void *keys, MsdPlan *msd, uint32_t max_bucket, LaunchEvents ev = {},
                                     uint32_t *clear_status = nullptr, size_t clear_words = 0, uint32_t *values = nullptr);
// values: uint32 payloads that follow their 64-bit keys (buckets up to msd_local_capacity_pairs_u64(false) elements)
// max_bucket: the plan's msd_max_bucket (picks the workgroup shape: 256 threads up to 7165 keys, else 512)
hipError_t launch_msd_local_sort(hipStream_t stream, uint32_t *keys, uint32_t *values, MsdPlan *msd, uint32_t max_bucket,
                                 LaunchEvents ev = {}, uint32_t *clear_status = nullptr, size_t clear_words = 0);
// keys the local sort of one bucket can hold (the plan refuses the hybrid form when a bucket has more)
uint32_t msd_local_capacity_small();  // bare uint32 keys, 256-thread workgroup: 7165
uint32_t msd_local_capacity_wave();   // bare uint32 keys, one wave per bucket: 1789
uint32_t msd_local_capacity(bool pairs_or_wide);  // pairs and 64-bit keys: 13312, uint32 keys: 14333
uint32_t msd_local_capacity_pairs_small();         // pairs and 64-bit keys, 512-thread workgroup (two per CU): 6656
uint32_t msd_local_capacity_pairs_u64(bool small); // 64-bit keys with payloads: 4096 (two workgroups per CU) / 6656 (one)

// ---- hybrid form WITHOUT a counting read ("pool" form, vrs_msd_pool.hip; bare uint32 keys): 24 instead of 28 bytes per key.
// The counting read exists to tell the MSD passes where every bucket's keys go.  Here nothing is counted ahead:
//   sample      1/32 of the input (the first 256 keys of every 8192-key tile) counted by bucket (the top 14 bits of the probed key
//               range) and, per input slice, by top byte; the layout kernel sizes, for every (input slice, top byte), a region of the
//               partner buffer plus a few standard deviations of overflow room in context scratch;
//   first pass  every tile reserves its output in those regions (one L2-local atomic per tile and top byte);
//   plan        one workgroup: EXACT top-byte totals from the first pass's cursors, the second pass's tile tables, and for every one
//               of the 16384 buckets a region of the SLACK buffer (context scratch of n + about 6 sqrt(16384 * 32 * n) slots):
//               the bucket's share of its top byte's exact total as the sample saw it, plus six standard deviations of that; verdict 1;
//   second pass every tile of a (top byte, slice) share scatters its keys by the next 6 bits into the buckets' slack regions,
//               reserving its runs with one L2-local atomic per tile and bucket; a bucket that outgrows its region or the local
//               sort's capacity, or a key outside the probed range, flags the sort;
//   local sort  one workgroup per bucket reads the bucket -- ONE contiguous, 16-byte aligned piece of the slack buffer --, sorts it
//               inside LDS (lean_sort_body, vrs_local_sort.hpp) and streams it to its final place in the caller's buffer, which it
//               derives from the top byte's exact start and the second pass's counters of the buckets before it.
// A region that overflows its room, a key outside the sampled range or a bucket above the local sort's capacity make a verdict say
// no (MsdPlan::ok == 0) before the caller's buffer has been written: the sort starts over in the counted form.
constexpr uint32_t kPoolTile = 8192;           // keys per tile of both passes
constexpr uint32_t kPoolRoomFloor = 320;        // slots every region gets on top of its six deviations
constexpr uint32_t kPoolPackedPairsMean = 5120;  // pairs: buckets of up to this many pairs on average take the local sort's packed form (launch_pool_local_sort)
constexpr int kPoolPairItems = 13;              // pairs per thread of the pairs' local sort (512 or 1024 threads: buckets up to 6656 / 13312)
constexpr uint32_t kPoolSampleKeys = 256;      // leading keys of every tile the sample kernel counts
constexpr uint32_t kPoolSampleTiles = 32;      // tiles per workgroup of the sample kernel
constexpr uint32_t kPoolMaxKeys = 224000000u;  // the fullest of 16384 uniform buckets (mean + 5.5 deviations) must fit the local sort's 512-thread shape (14333 keys)
constexpr uint32_t kPoolMaxBuckets = 256u << 8; // the second pass sorts by 6 or 7 bits of 256 top bytes (16384 or 32768 buckets); the second half alone
                                                // (grouped keys) and the 8 + 8 cut (VRS_TUNE_MSD_POOL_SUB_BITS 8: 65536 buckets, one wave each) also by 8
constexpr uint32_t kPoolMaxTilesA = 3456;      // tiles per slice of the first pass at kPoolMaxKeys (3418)
constexpr uint32_t kPoolTileGeneral = 0xFFFFFFFFu;
constexpr uint32_t kPoolMaxTilesB = 4352;      // rows of workgroups of the second pass at kPoolMaxKeys (pool_tiles_b_cap: 4313)
struct PoolStreams {                           // the eight slices of the input the first pass walks (whole tiles), by value
    uint32_t start[8], len[8], sampled[8];     // first key, keys, keys the sample kernel counts
    uint32_t tiles_per_stream, tiles_total;
};
struct PoolPlan {
    uint32_t shift;             // bucket = (key - key_base) >> shift: the top 14 bits of the probed key range
    uint32_t armed;             // 1 = the sample kernel laid the regions out: the first pass runs
    uint32_t fail[2];           // (one word per PARITY of the context's pool epoch: the plan kernel of a sort zeroes the OTHER one, so the next sort's
                                //   first pass -- which may run without a sample and layout kernel in front, see launch_pool_sample -- finds its word clear)
                                // bit 0: a pass found a region out of room / a key outside the probed range / a workgroup behind an unknown L2 / a tile
                                //   claimed twice: the sort is refused; bit 1: a bucket has more keys than the local sort that was enqueued takes
                                //   (it lies whole in its region: a local sort of a larger shape can still finish the sort).  Re-armed by the layout kernel
    uint32_t ticket;            // (unused)
    uint32_t ok_a;              // verdict 1 (plan kernel): the second pass runs
    uint32_t max_bucket;        // second pass: the fullest bucket AMONG those beyond the enqueued local sort's capacity (zero if none; re-armed by the plan kernel)
    uint32_t pad[1];
    uint32_t sample[8][256];    // sampled keys of (slice, top byte), zero between sorts
    uint32_t base[8][256];      // primary region of (slice, top byte): first slot in the partner buffer
    uint32_t cap[8][256];       //   its slots
    uint32_t obase[8][256];     // overflow region: first slot in the overflow scratch
    uint32_t ocap[8][256];
    uint32_t top_base[260];     // where top byte a starts in the sorted order (exact), [256] = n
    uint32_t tiles_b[8][260];   // second pass: XCD x walks entries e = 8 k + s (top byte x + 8 k, slice s): exclusive prefix of their tile counts, [256] = all
    // Which tile a workgroup of a pass takes follows the XCC it RUNS on (list = that XCC's place in the probed order, tile = block
    // index / 8): whatever the dispatcher's rotation, the eight blocks of a group then take eight different lists -- as long as they
    // run on eight different XCCs.  That is observed, not promised: every workgroup leaves a claim, and a tile claimed twice refuses the sort.
    uint32_t claim_a[8 * kPoolMaxTilesA];  // first pass: += 1 by the workgroup of (list, tile); checked and zeroed by the plan kernel
    uint32_t claim_b[8 * kPoolMaxTilesB];  // second pass: exchanged with the sort's stamp by the workgroup of (list, tile); a stamp found there = a second claim
    // second pass, tile j of XCD x's list: .x = the virtual slot of the tile's first key if the tile is 8192 keys inside ONE piece (five
    // tiles in six: its loads then depend on this one word), else kPoolTileGeneral (it gathers from the top byte's piece row);
    // .y = top byte | the tile's index inside its top byte << 8
    alignas(16) uint2 tile_map[8][kPoolMaxTilesB];
    alignas(16) uint2 pieces[256][16];                     // top byte a's keys lie in 16 pieces (slice s's primary region: 2 s, its overflow region: 2 s + 1):
                                                           //   .x = keys of the top byte up to and including the piece, .y = the piece's first virtual slot
    alignas(16) uint32_t sub_start[kPoolMaxBuckets + 4];   // bucket b's region of the slack buffer: first slot (a multiple of 4), [buckets] = slots in all
    alignas(16) uint32_t sub_cursor[kPoolMaxBuckets];      // second pass: keys of bucket b placed so far (zeroed by the plan kernel)
};
PoolStreams pool_streams(uint32_t n);
uint32_t pool_overflow_capacity(uint32_t n);   // keys of overflow scratch a sort of n keys may use
uint32_t pool_slack_capacity(uint32_t n, uint32_t sub_bits, uint32_t top_bytes = 256);  // slots of the slack buffer (regions of all buckets + one tile where refused runs are dumped)
// The shape of a pool sort, chosen from n alone (the form is enqueued blind): bits of the second pass and the local sort's workgroup.
struct PoolShape {
    uint32_t sub_bits;   // 6 or 7
    uint32_t local;      // 3: one WAVE per bucket (up to 1789 keys), 0: 256 threads x 16 slots (4093, five workgroups per CU), 1: 256 x 28 (7165, four), 2: 512 x 28 (14333, two);
                         // key + payload pairs: 4: 512 threads x 13 pairs (6656), 5: 1024 x 13 (13312)
};
PoolShape pool_shape(uint32_t n, int forced_sub_bits = 0);  // forced_sub_bits: 0 = by size, 6, 7 or 8
PoolShape pool_shape_pairs(uint32_t n);
struct PoolCut {
    uint32_t top_bits, sub_bits;  // bits of the first and of the second pass
    uint32_t local;               // PoolShape::local
};
PoolCut pool_cut(uint32_t n, bool pairs, int top_bits_setting, int forced_sub_bits);  // what one_read_enqueue_pool runs (host only)
uint32_t pool_max_pairs();  // the most pairs whose fullest uniform bucket fits the shape pool_shape_pairs gives them
// Key + payload pairs (the STABLE pool form: a tile's place in a region is its rank there, by decoupled look-back): the payloads'
// twins of the keys' buffers and the look-back's status words (the one-call sort's; `status_words` of them, all cleared by the local
// sort of a sort that is taken).  By value: a kernel argument.
struct PoolPayloads {
    uint32_t *values_home = nullptr;      // the caller's payloads: read by the first pass, written by the local sort
    uint32_t *values_partner = nullptr;   // the primary regions' twin
    uint32_t *overflow_values = nullptr;  // the overflow regions' twin
    uint32_t *slack_values = nullptr;     // the slack buffer's twin
    uint32_t *status = nullptr;
    size_t status_words = 0;
    uint32_t spin_budget = 0;
    int hold_tile = -1;                   // test hook (VRS_TUNE_DEBUG_HOLD_TILE): this tile of every slice never publishes in the first pass
    int packed = -1;                      // VRS_TUNE_MSD_POOL_PAIRS_PACKED: the local sort's packed form -- -1 by the buckets' mean size, 0 never, 1 always
    uint32_t top_bits = 8;                // bits the first pass took (the cut of the buckets' bits between the passes): the local sort derives the bits a bucket's keys differ in
};
// the second half alone, for n keys grouped by `top_bytes` top bytes: sub_bits 6 .. 8 (0: no shape takes them)
PoolShape pool_grouped_shape(uint32_t n, uint32_t top_bytes);
struct PoolGroups {          // keys of every top byte (grouped keys; by value: a kernel argument)
    uint32_t count[256];     // zero from top_bytes on
    uint32_t top_bytes;      // top bytes that exist: top_bytes << sub_bits buckets
    // A part of every top byte's keys may lie ELSEWHERE (a rank's own keys, left where its partition pass wrote them): the LAST own[a]
    // slots of top byte a's range of the grouped buffer are a hole, and those keys are own[a] consecutive slots of a second buffer (the
    // second pass's `overflow` pointer), the top bytes' own parts following each other from slot own_first on.  All zero: none.
    uint32_t own_first;
    uint32_t own[256];
};
uint32_t pool_local_capacity(uint32_t local);
uint32_t pool_tiles_b_cap(uint32_t n);         // rows of workgroups of the second pass (its grid is sized before the plan is known)
// par: the parity of the context's pool epoch (0 / 1; the same for all kernels of one sort: PoolPlan::fail)
hipError_t launch_pool_sample(hipStream_t stream, const uint32_t *keys, uint32_t n, uint32_t key_base, const PoolStreams &ps,
                              PoolPlan *pool, uint32_t overflow_capacity, uint32_t par, LaunchEvents ev = {}, uint32_t top_bits = 8);
// top_bits (every launcher of the form; lab switch VRS_TUNE_MSD_POOL_TOP_BITS): bits of the first pass's digit, 8 -- or 7, with a second pass of 7
// keys_out: the partner buffer (n slots); overflow: pool_overflow_capacity(n) slots; cursors: MsdPlan::cursor_a (zero when the pass
// starts); misplace: test hook, odd rows of workgroups walk the neighbouring slice
hipError_t launch_pool_pass_a(hipStream_t stream, const uint32_t *keys_in, uint32_t *keys_out, uint32_t *overflow, uint32_t n,
                              uint32_t key_base, const PoolStreams &ps, PoolPlan *pool, MsdPlan *msd, unsigned long long xcc_map,
                              bool misplace, uint32_t overflow_capacity, uint32_t par, LaunchEvents ev = {}, const PoolPayloads *pv = nullptr, uint32_t top_bits = 8);
// after the first pass, one workgroup per top byte: top-byte starts, tile tables, piece rows, the buckets' slack regions (from a
// sample of the first pass's OUTPUT: regions / overflow), verdict 1 (slack_capacity: slots the slack buffer has)
hipError_t launch_pool_plan(hipStream_t stream, MsdPlan *msd, PoolPlan *pool, uint32_t n, uint32_t tiles_b_cap, uint32_t slack_capacity,
                            const uint32_t *regions, const uint32_t *overflow, uint32_t key_base, const PoolStreams &ps, uint32_t sub_bits,
                            uint32_t par, const PoolGroups *groups = nullptr, bool keep_rooms = false, uint32_t top_bits = 8);
// groups != nullptr: the second half alone (vrs_msd_finish_grouped_counts_u32) -- `regions` holds keys grouped by top byte, top byte a
// (counted from key_base >> 24) holds groups->count[a] of them; no first pass ran
// second pass, regions -> slack buffer: grid of 8 * tiles_b workgroups (tiles_b = pool_tiles_b_cap(n)); local_cap: keys the local
// sort that follows takes per bucket; slack_capacity: as given to the plan (the last kPoolTile slots take refused runs)
hipError_t launch_pool_pass_b(hipStream_t stream, const uint32_t *regions, const uint32_t *overflow, uint32_t *slack, uint32_t n, MsdPlan *msd,
                              PoolPlan *pool, uint32_t tiles_b, uint32_t key_base, uint32_t local_cap, uint32_t slack_capacity,
                              unsigned long long xcc_map, uint32_t stamp, uint32_t sub_bits, uint32_t par, LaunchEvents ev = {}, bool grouped = false,
                              const PoolPayloads *pv = nullptr, uint32_t top_bits = 8);
// sorts every bucket from its slack region to keys_out[its exact start ...) with the workgroup shape.local.  Gives verdict 2 (verdict 1, no flag from the passes) = MsdPlan::ok and the host head
// (msd_ok, lsd_missing = 1, stamped last); re-arms the first pass's reservation counters
hipError_t launch_pool_local_sort(hipStream_t stream, const uint32_t *slack, uint32_t *keys_out, uint32_t n, MsdPlan *msd, const PoolPlan *pool,
                                  PoolShape shape, OnesweepPlanHead *dev_head, OnesweepPlanHead *host_head, uint32_t stamp, uint32_t par,
                                  LaunchEvents ev = {}, uint32_t top_bytes = 256, uint32_t *host_log = nullptr, bool retry = false,
                                  const PoolPayloads *pv = nullptr);
// retry: the second attempt after a first local sort found a bucket beyond its shape (PoolPlan::fail bit 1): that bit no longer refuses
// top_bytes: the top bytes that exist (a sort: 256; the second half alone: the caller's); host_log: see launch_msd_plan

// out[b] = HW_REG_XCC_ID of block b of a `blocks`-block grid of 512-thread workgroups
hipError_t launch_xcc_probe(hipStream_t stream, uint32_t *out, uint32_t blocks);

hipError_t launch_single(hipStream_t stream, uint32_t *buffer0, uint32_t *buffer1, uint32_t n,
                         LaunchEvents ev = {});

}  // namespace vrs
