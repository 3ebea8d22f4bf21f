/*
 * vkradixsort_amd.h -- C ABI of the MI355X-native multi-block LSD radix sort.
 *
 * This is the drop-in boundary for VkRadixSort's `multi_radixsort` path.  The reference has no
 * FFI layer of its own: its boundary is the C++ class API (GPUContext / Buffer / ComputePass /
 * MultiRadixSortPass / MultiRadixSort).  The C++ host mirror under vkradixsort_amd/host keeps those
 * classes and calls ONLY the functions below; any other host language binds the same symbols
 * (see INTEGRATION.md).  Plain C types, opaque handles, no exceptions, 0 == success.
 *
 * Each entry point cites the reference interface it replaces (file:line in VkRadixSort @ v2).
 *
 * Threading: one HIP stream per context, a context is not thread-safe (the reference is single
 * threaded: one compute queue, one host thread).  Stage calls are ASYNCHRONOUS and stream-ordered;
 * the only blocking calls are vrs_queue_wait_idle, upload/download and the profile query.
 */
#ifndef VKRADIXSORT_AMD_H
#define VKRADIXSORT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VRS_RADIX_SORT_BINS 256u /* multi_radixsort.comp:12 */
#define VRS_WORKGROUP_SIZE 256u  /* multi_radixsort.comp:11 -- the CONTRACT workgroup size that
                                    defines tiles and the [W][256] histogram layout */

typedef enum vrs_status {
    VRS_OK = 0,
    VRS_ERROR_INVALID_ARGUMENT = 1,
    VRS_ERROR_HIP = 2,          /* a HIP runtime call failed; see vrs_last_error */
    VRS_ERROR_NO_DEVICE = 3,    /* no gfx950-capable device / bad ordinal */
    VRS_ERROR_OUT_OF_MEMORY = 4,
    VRS_ERROR_UNBALANCED = 5,   /* vrs_dist_sort_keys_u32: too many equal keys -- no cut between key values balances the ranks */
    VRS_ERROR_PEER = 7,         /* vrs_dist_sort_keys_u32: another rank could not take part; every rank left the step together */
    VRS_ERROR_TIMEOUT = 6       /* a one-call sort's plan did not reach the host within VRS_TUNE_PLAN_WAIT_MS (the stream is held
                                   up by earlier work); the sort is still queued, vrs_sort_settle may be called again */
} vrs_status;

typedef struct vrs_context_t *vrs_context; /* replaces engine::GPUContext (GPUContext.h:15-111) */
typedef struct vrs_buffer_t *vrs_buffer;   /* replaces engine::Buffer     (Buffer.h:14-177)     */

/*
 * The 16-byte push-constant block shared by both stages, std430 order:
 * MultiRadixSortPass::PushConstantsHistograms (MultiRadixSortPass.h:17-22) and
 * MultiRadixSortPass::PushConstants           (MultiRadixSortPass.h:26-31), mirroring
 * multi_radixsort_histograms.comp:13-18 and multi_radixsort.comp:17-22.
 */
typedef struct vrs_push_constants {
    uint32_t g_num_elements;             /* N */
    uint32_t g_shift;                    /* 0, 8, 16, 24 -- set by the caller each pass
                                            (MultiRadixSort.cpp:57-58) */
    uint32_t g_num_workgroups;           /* W = ceil(ceil(N/B)/256)  (MultiRadixSort.cpp:13-20) */
    uint32_t g_num_blocks_per_workgroup; /* B = NUM_BLOCKS_PER_WORKGROUP (MultiRadixSort.cpp:12) */
} vrs_push_constants;

/* ---- device / context: GPUContext::init / shutdown (GPUContext.cpp:7-13) ------------------- */

int vrs_device_count(int *count);
/* Creates a context on `device_ordinal` with its own HIP stream.  Never prompts on stdin (the
 * reference does when >1 device: GPUContext.cpp:167-174). */
int vrs_context_create(int device_ordinal, vrs_context *out_ctx);
/* Same, but borrows a caller-owned hipStream_t (e.g. torch's current stream); never destroyed. */
int vrs_context_create_on_stream(int device_ordinal, void *hip_stream, vrs_context *out_ctx);
int vrs_context_destroy(vrs_context ctx);
/* Message of the last failure on this context (ctx may be NULL for creation failures). */
const char *vrs_last_error(vrs_context ctx);
/* hipStream_t of the context, for callers that interleave their own work. */
void *vrs_context_stream(vrs_context ctx);
int vrs_context_device(vrs_context ctx); /* the device ordinal the context was created on */
/* Device facts for reports: name, CU count, HBM bytes.  Any out pointer may be NULL. */
int vrs_device_info(vrs_context ctx, char *name, size_t name_cap, int *compute_units,
                    uint64_t *global_mem_bytes);

/* ---- buffers: Buffer ctor / release / staging copies (Buffer.h:25-72) ---------------------- */

/* Device-local allocation (VK_MEMORY_PROPERTY_DEVICE_LOCAL_BIT buffers of prepareBuffers,
 * MultiRadixSort.cpp:83-95).  size_t, not the reference's uint32 byte size (Buffer.h:17). */
int vrs_buffer_create(vrs_context ctx, size_t size_bytes, vrs_buffer *out_buf);
/* Wraps caller-owned device memory (hipMalloc / torch tensor); release does not free it.
 * `device_ptr` must be aligned to the element size (4 bytes; 8 for uint64 keys); 16-byte aligned bases
 * (every hipMalloc / torch allocation) take the fastest load path.  ("own usage", README.md:151-241.) */
int vrs_buffer_wrap(vrs_context ctx, void *device_ptr, size_t size_bytes, vrs_buffer *out_buf);
/* Idempotent (Buffer::release, Buffer.h:36-45).  Also frees the handle. */
int vrs_buffer_release(vrs_buffer buf);
/* Synchronous H2D copy of `size_bytes` (Buffer::fillDeviceWithStagingBuffer, Buffer.h:47-62). */
int vrs_buffer_upload(vrs_context ctx, vrs_buffer buf, const void *host_data, size_t size_bytes);
/* Synchronous D2H copy (Buffer::downloadWithStagingBuffer, Buffer.h:64-72). */
int vrs_buffer_download(vrs_context ctx, vrs_buffer buf, void *host_data, size_t size_bytes);
/* Stream-ordered device-to-device copy (used to re-arm the unsorted input between timed reps). */
int vrs_buffer_copy(vrs_context ctx, vrs_buffer dst, vrs_buffer src, size_t size_bytes);
void *vrs_buffer_device_ptr(vrs_buffer buf);
size_t vrs_buffer_size_bytes(vrs_buffer buf); /* Buffer::getSizeBytes, Buffer.h:103-105 */

/* ---- launch-shape arithmetic: ComputePass::setGlobalInvocationSize/getWorkGroupCount ------- */

/* gis = ceil(N/B) (MultiRadixSort.cpp:13-15); W = ceil(gis/256) (ComputePass.h:16-29). */
uint32_t vrs_global_invocation_size(uint32_t num_elements, uint32_t blocks_per_workgroup);
uint32_t vrs_workgroup_count(uint32_t num_elements, uint32_t blocks_per_workgroup);

/* ---- the two stages: MultiRadixSortPass::recordCommands (MultiRadixSortPass.cpp:10-20) ----- */

/*
 * Stage RADIX_SORT_HISTOGRAMS (multi_radixsort_histograms.comp:31-55).
 * bindings: set 0 b0 = keys_in (uint32[N]), set 0 b1 = histograms (uint32[W*256], layout
 * [workgroup][digit], every entry overwritten).  Asynchronous.
 */
int vrs_multi_radixsort_histograms(vrs_context ctx, vrs_buffer keys_in, vrs_buffer histograms,
                                   const vrs_push_constants *pc);
/*
 * Stage RADIX_SORT (multi_radixsort.comp:45-127): global digit prefix + per-workgroup offsets from
 * `histograms`, then the stable scatter keys_in -> keys_out.  bindings: set 1 b0 in, b1 out,
 * b2 histograms.  Stream-ordered after the histogram stage (replaces the W->R pipeline barrier,
 * MultiRadixSortPass.cpp:13-14).  keys_in and keys_out must not alias.  Asynchronous.
 */
int vrs_multi_radixsort(vrs_context ctx, vrs_buffer keys_in, vrs_buffer keys_out,
                        vrs_buffer histograms, const vrs_push_constants *pc);
/*
 * Key + payload variant of the RADIX_SORT stage (build extension, BASELINE.json config 4: the
 * reference has only g_elements_in/out, multi_radixsort.comp:24-30).  values follow their keys;
 * order among equal digits is the input order (stable), so four passes == std::stable_sort by key.
 */
int vrs_multi_radixsort_pairs(vrs_context ctx, vrs_buffer keys_in, vrs_buffer keys_out,
                              vrs_buffer values_in, vrs_buffer values_out, vrs_buffer histograms,
                              const vrs_push_constants *pc);
/*
 * Global exclusive digit prefix (uint32[256]) computed by the most recent RADIX_SORT stage on this
 * context -- the `global_histogram` of multi_radixsort.comp:74-75: position of the first key of each
 * digit in that stage's output.  Synchronous.  Used by the multi-GPU key-range exchange (build
 * extension) to cut the locally grouped shard into per-rank slices.
 */
int vrs_multi_radixsort_digit_offsets(vrs_context ctx, void *host_u32x256);
/* The same 256 words copied into a device buffer, asynchronously (no host round trip: the multi-GPU step feeds
 * them straight into its count all-gather). */
int vrs_multi_radixsort_digit_offsets_device(vrs_context ctx, vrs_buffer out_u32x256);
/* One-shot hook for the NEXT RADIX_SORT stage of the context: as soon as its offset table is complete -- before its scatter kernel --
 * the stage copies the 256 digit offsets to out_u32x256 (may be NULL) and records `event` (a hipEvent_t of the caller's, may be NULL)
 * on the context's stream.  Work on another stream that needs only the offsets (the multi-GPU step's all-gather) then runs beside
 * the scatter.  (The reference has no counterpart: its prefix lives inside multi_radixsort.comp:63-87.) */
int vrs_multi_radixsort_offsets_hook(vrs_context ctx, vrs_buffer out_u32x256, void *event);
/*
 * 64-bit keys: the reference's SORT_64_BIT switch (MultiRadixSort.h:10-18; NUM_ITERATIONS = 8,
 * MultiRadixSort.cpp:51-55), which it leaves as a stub ("requires changes in the two shaders").  Same two
 * stages, same [W][256] table and push constants; keys are uint64, g_shift is 0, 8, ..., 56, the caller runs
 * eight passes (even count: the result is in buffer 0 again).  Payloads of the pairs form stay uint32.
 */
int vrs_multi_radixsort_histograms_u64(vrs_context ctx, vrs_buffer keys_in, vrs_buffer histograms,
                                       const vrs_push_constants *pc);
int vrs_multi_radixsort_u64(vrs_context ctx, vrs_buffer keys_in, vrs_buffer keys_out,
                            vrs_buffer histograms, const vrs_push_constants *pc);
int vrs_multi_radixsort_pairs_u64(vrs_context ctx, vrs_buffer keys_in, vrs_buffer keys_out,
                                  vrs_buffer values_in, vrs_buffer values_out, vrs_buffer histograms,
                                  const vrs_push_constants *pc);
/*
 * Range partition (build extension for the multi-GPU key-range exchange; no reference counterpart): one
 * stable pass that groups uint32 keys by range instead of by digit.  `splitters` holds `num_splitters` <= 255
 * ascending uint32 values on the device; a key goes to range r = number of splitters <= key.  keys_out holds
 * range 0's keys, then range 1's, ... each in input order; vrs_multi_radixsort_digit_offsets then returns
 * the first position of every range (entries 0..num_splitters).  Asynchronous.
 */
int vrs_range_partition(vrs_context ctx, vrs_buffer keys_in, vrs_buffer keys_out, vrs_buffer splitters,
                        uint32_t num_splitters, uint32_t num_elements);
/* vkQueueWaitIdle on the compute queue (MultiRadixSort.cpp:62). */
int vrs_queue_wait_idle(vrs_context ctx);

/*
 * single_radixsort path (single_radixsort.comp:42-140, SingleRadixSort.cpp:5-47): one workgroup,
 * four passes in one launch, push constant {g_num_elements}; result in buffer0.  Asynchronous.
 */
int vrs_single_radixsort(vrs_context ctx, vrs_buffer buffer0, vrs_buffer buffer1,
                         uint32_t g_num_elements);

/*
 * One-call form of MultiRadixSort::execute's hot loop (MultiRadixSort.cpp:50-61): four passes with
 * g_shift = 0, 8, 16, 24 ping-ponging `keys` <-> `keys_tmp` (the reference's buffer0 / buffer1); the
 * library picks NUM_BLOCKS_PER_WORKGROUP (32) and owns the histogram table.  Result in `keys`.
 * The pairs form is stable (== std::stable_sort by key); `values` follow their keys.
 *
 * Up to VRS_TUNE_SINGLE_MAX_KEYS uint32 keys (default 4096) the whole sort is one single_radixsort launch (the
 * reference's guidance for small inputs, README.md:18-21).  Below VRS_TUNE_ONE_CALL_MIN_KEYS elements (default 2^13)
 * these are the four (eight) contract passes, fully asynchronous.  From there on -- the library owns all passes, so the per-pass [W][256] table of the
 * reference's interface is not needed -- the keys are read ONCE to count all four digits of a 32-bit word
 * and every pass is a stable scatter that finds its offsets by decoupled look-back (36 instead of 48 bytes
 * per 32-bit key, 136 instead of 192 per 64-bit key; DESIGN.md "K5").  That form enqueues everything, then waits on
 * the host ONCE per four passes until the plan kernel has written the plan's head (a few hundred bytes) into pinned
 * host memory -- i.e. until the counting read has run; it never waits for the sort itself, which still completes
 * asynchronously on the context's stream.  Passes whose digit is the same for every key (small keys, constant bytes)
 * are the identity and are left out.  Same result, bit for bit.
 * uint32 keys from 1.3 * 10^7 keys on, uint32 key + payload pairs from 2.5 * 10^7 pairs on (VRS_TUNE_HYBRID,
 * VRS_TUNE_HYBRID_MIN_KEYS):
 * the same counting read also histograms the top 14 bits of the key range, and when every such bucket fits one workgroup's
 * LDS (14333 keys, 13312 pairs or 64-bit keys; uniform input: up to about 2.2 * 10^8 keys, 2.1 * 10^8 pairs) the four LSD passes are
 * replaced by an MSD partition in two look-back scatter passes (8 + 6 bits) plus one pass in which every bucket is sorted
 * inside LDS -- 28 bytes per key instead of 36, 52 per pair instead of 68 (DESIGN.md "K5b").  The choice is made on the
 * device from that one read; either form gives the same bits, payloads of equal keys in input order included.
 * vrs_sort_keys_u64 takes the same form from 2 * 10^7 keys on (56 instead of 144 bytes per key; the local sort then needs
 * ceil(low bits / 9) LDS passes, up to six); its counting read never makes LSD tables, so a refused sort starts over.
 *
 * Blocking behaviour.  On a context with its OWN stream (vrs_context_create) these calls only ENQUEUE and return at once -- like
 * the reference's ComputePass::execute (ComputePass.h:31-56), whose only blocking call is the queue-idle wait
 * (MultiRadixSort.cpp:62): a sort the hybrid form is expected to take (the context's previous one of its kind did) is put on the
 * stream completely (second MSD pass and local sort with grids sized for the worst plan the form accepts; the pool form with
 * all six kernels), the LSD form as four speculative look-back passes.  What the plan may still ask for (a refused hybrid
 * form: the LSD sort; a pass with unbalanced or wide streams; the copy home after an odd number of passes) is enqueued by
 * vrs_sort_settle, which waits for the plan's head -- never for the sort; vrs_queue_wait_idle, the blocking buffer transfers,
 * vrs_verify_keys_u32, every stage / sort entry point and vrs_context_destroy settle a pending sort first.  Until then the
 * buffers must stay alive, and a caller who queues work of his own behind the sort (vrs_context_stream) calls vrs_sort_settle
 * before he does.  vrs_sort_pending tells whether a second half is outstanding (0 / 1).
 * On a BORROWED stream (vrs_context_create_on_stream) the caller waits with calls of his own (hipStreamSynchronize,
 * torch.cuda.synchronize), so the default there is the blocking form: the call returns once the plan's head has reached the
 * host and everything the plan asks for is on the stream (it spins briefly, then yields; never longer than
 * VRS_TUNE_PLAN_WAIT_MS -> VRS_ERROR_TIMEOUT) -- on a stream that still has earlier work queued that means waiting for that
 * work.  VRS_TUNE_ASYNC_SORT selects either form on either kind of context.
 */
int vrs_sort_keys_u32(vrs_context ctx, vrs_buffer keys, vrs_buffer keys_tmp, uint32_t num_elements);
/* The same sort for keys the caller knows to lie in [key_floor, 2^32) -- a sub-range of a larger sort (the received key range
 * of a multi-GPU step): the hybrid form then takes its 16384 buckets from key - (key_floor rounded down to a multiple of 2^24),
 * as it would for a full key range, instead of finding almost all of them empty and the rest too large.  A hint only: a key
 * below the floor makes the plan refuse the hybrid form (the LSD passes run) -- the result is the same either way. */
int vrs_sort_keys_u32_ranged(vrs_context ctx, vrs_buffer keys, vrs_buffer keys_tmp, uint32_t num_elements, uint32_t key_floor);
int vrs_sort_settle(vrs_context ctx);
int vrs_sort_pending(vrs_context ctx);
int vrs_sort_pairs_u32(vrs_context ctx, vrs_buffer keys, vrs_buffer keys_tmp, vrs_buffer values,
                       vrs_buffer values_tmp, uint32_t num_elements);
int vrs_sort_keys_u64(vrs_context ctx, vrs_buffer keys, vrs_buffer keys_tmp, uint32_t num_elements); /* 8 LSD passes, or the hybrid form */
/* uint64 keys with uint32 payloads: two groups of [counting read + four look-back scatter passes] from
 * VRS_TUNE_ONE_CALL_MIN_KEYS pairs on (2 x (8 + 4 x 24) = 208 instead of 8 x (8 + 24) = 256 bytes per pair), the eight contract passes
 * below; stable either way.  No hybrid form. */
int vrs_sort_pairs_u64(vrs_context ctx, vrs_buffer keys, vrs_buffer keys_tmp, vrs_buffer values,
                       vrs_buffer values_tmp, uint32_t num_elements);

/*
 * The hybrid form of vrs_sort_keys_u32 in two halves, for callers that move the keys between its two MSD passes -- the
 * multi-GPU step (vrs_dist_*) puts the exchange between the GPUs there.  Both only enqueue.
 *   vrs_msd_partition_u32: counting read (histogram of the top 14 bits of the probed key range + top-byte counts) and the first
 *     MSD pass: `out` = the keys grouped by the top 8 bits of the range (in no particular order inside a group when the pass takes
 *     its places by reservation, VRS_TUNE_MSD_RESERVE: the default).  counts_out (VRS_MSD_COUNT_WORDS uint32):
 *     [0, 16384) the bucket histogram, [16384, 16384 + 8 * 256) the top-byte counts of the eight input slices,
 *     [VRS_MSD_SHIFT_WORD] the bucket shift (bucket = key >> shift; below 13 the key range is too narrow for the form: the
 *     histogram is empty and `out` is not a partition), [VRS_MSD_SHIFT_WORD + 1] != 0: a key above the probed range.
 *   vrs_msd_finish_u32: second MSD pass + local sort of n keys that ARE grouped by that top byte (`grouped`; clobbered as
 *     scratch), result in `out`.  counts: the same layout, [0, 16384) = the bucket histogram of exactly these n keys, the
 *     slice counts zero, the shift word set, the flag word zero; read in place by the plan kernel, which leaves the histogram zeroed.  bucket_hint: the largest bucket to expect (it picks the local
 *     sort's workgroup shape before the plan is known; 0 = num_elements / 16384 + 10 %).  The plan may refuse (a bucket beyond the local sort's
 *     capacity, top-byte buckets too unequal for the grid): vrs_msd_finish_status waits for the plan's head and tells
 *     (*took == 0: `out` holds nothing useful, `grouped` still holds the keys -- sort them with vrs_sort_keys_u32).
 *   A caller that enqueues several finishes before it looks (the rounds of the multi-GPU step: a wait per round would leave the
 *     GPU idle while the host enqueues the next) takes a ticket after each (vrs_msd_finish_ticket) and asks later:
 *     vrs_msd_finish_status_at(ticket) waits until THAT finish's plan has run.  The context keeps the last 32 decisions.
 */
#define VRS_MSD_COUNT_WORDS (16384u + 8u * 256u + 64u)
#define VRS_MSD_SHIFT_WORD (16384u + 8u * 256u)
int vrs_msd_partition_u32(vrs_context ctx, vrs_buffer keys, vrs_buffer out, vrs_buffer counts_out, uint32_t num_elements);
/* The same; counts_ready_event (a hipEvent_t of the caller's, or NULL) is recorded on the context's stream as soon as counts_out is
 * complete -- before the first MSD pass -- so that work on ANOTHER stream that needs only the counts (the multi-GPU step's
 * collectives) runs beside that pass. */
int vrs_msd_partition_signal_u32(vrs_context ctx, vrs_buffer keys, vrs_buffer out, vrs_buffer counts_out, uint32_t num_elements,
                                 void *counts_ready_event);
int vrs_msd_finish_u32(vrs_context ctx, vrs_buffer grouped, vrs_buffer out, vrs_buffer counts, uint32_t num_elements,
                       uint32_t bucket_hint);
/* The same second half for keys that are grouped by their TOP BYTE and lie in top bytes [first_top_byte, first_top_byte + top_bytes)
 * -- what a rank of a large multi-GPU sort holds after the exchange -- with no histogram from the caller: one counting read of the
 * n keys (buckets = top byte and the next min(8, 14 - ceil(log2(top_bytes))) bits), then the second MSD pass by those bits and the
 * local sort: 20 bytes per key instead of the 28 of a whole sort.  Refused (ticket / status as above; `grouped` untouched) when a
 * bucket exceeds the local sort's capacity: about 3.6 * 10^6 keys per top byte. */
int vrs_msd_finish_grouped_u32(vrs_context ctx, vrs_buffer grouped, vrs_buffer out, uint32_t num_elements, uint32_t first_top_byte,
                               uint32_t top_bytes);
/* The same for a caller that KNOWS how many keys each of those top bytes holds (counts[0 .. top_bytes): host memory, read before the
 * call returns; their sum must be num_elements) -- the multi-GPU step does, from the exchange's own bookkeeping.  Nothing is read to
 * be counted: one workgroup per top byte samples 1/32 of its keys and sizes a region of the context's slack buffer per bucket (top byte
 * and the next 6 .. 8 bits), the second MSD pass scatters there, the local sort reads every bucket in one piece -- the second half of
 * the pool form (VRS_TUNE_MSD_POOL), 16 bytes per key instead of 20.  Refused (ticket / status as above; `grouped` untouched) when a
 * bucket outgrows its region or the local sort's capacity, or a key does not carry the top byte its place says.  Falls back to
 * vrs_msd_finish_grouped_u32 where the form cannot run (switched off, no workgroup shape for these buckets, fewer than 2^20 keys,
 * counts == NULL). */
int vrs_msd_finish_grouped_counts_u32(vrs_context ctx, vrs_buffer grouped, vrs_buffer out, uint32_t num_elements, uint32_t first_top_byte,
                                      uint32_t top_bytes, const uint32_t *counts);
/* The same with a part of every top byte's keys lying ELSEWHERE -- the keys a rank of the multi-GPU step keeps for itself stay where its
 * partition pass wrote them instead of being copied beside the received ones (1 / world of the exchange's 8 bytes per key): top byte a's
 * range of `grouped` is laid out for all counts[a] keys, its LAST own_counts[a] slots are a hole, and those keys are own_counts[a]
 * consecutive keys of `own`, the top bytes' own parts following each other from key own_offset on.  The second pass reads a top byte as
 * two pieces; nothing else changes (verdicts, ticket / status as above; `grouped` and `own` untouched by a refusal).  Where the pool
 * form's second half cannot run the own parts are copied into their holes and vrs_msd_finish_grouped_u32 runs.  own == NULL (and
 * own_counts == NULL): vrs_msd_finish_grouped_counts_u32. */
int vrs_msd_finish_grouped_split_u32(vrs_context ctx, vrs_buffer grouped, vrs_buffer own, uint64_t own_offset, vrs_buffer out,
                                     uint32_t num_elements, uint32_t first_top_byte, uint32_t top_bytes, const uint32_t *counts,
                                     const uint32_t *own_counts);
int vrs_msd_finish_status(vrs_context ctx, int *took);
int vrs_msd_finish_ticket(vrs_context ctx, uint32_t *ticket);
int vrs_msd_finish_status_at(vrs_context ctx, uint32_t ticket, int *took);

/*
 * Key preprocessing the reference leaves to the integrator ("you have to preprocess negative numbers",
 * README.md:154-155): in-place, order-preserving maps between int32 / float32 bit patterns and the uint32
 * keys the sort orders.  Asynchronous.  Apply *_TO_SORTABLE before the four passes and the inverse after.
 */
typedef enum vrs_key_transform {
    VRS_KEYS_INT32 = 0,               /* int32 <-> sortable: flips the sign bit (self-inverse) */
    VRS_KEYS_FLOAT32_TO_SORTABLE = 1, /* IEEE-754 total order: -NaN < -inf < ... < -0 < +0 < ... < +inf < +NaN */
    VRS_KEYS_SORTABLE_TO_FLOAT32 = 2
} vrs_key_transform;
int vrs_transform_keys(vrs_context ctx, vrs_buffer keys, uint32_t num_elements, int mode);

/*
 * On-device counterpart of MultiRadixSort::verify / testSort (MultiRadixSort.cpp:97-102,148-161), for results too
 * many or too large to download and compare on the host: *descents = number of positions with keys[i] > keys[i+1]
 * (0 == ascending); *key_sum and *key_mix = order-independent fingerprints (plain sum, sum of a 64-bit mix of every
 * key) -- equal before and after a sort iff, up to hash collisions, the output is a permutation of the input.
 * Synchronous (one read of the keys).  Any of the three outputs may be NULL.
 */
int vrs_verify_keys_u32(vrs_context ctx, vrs_buffer keys, uint32_t num_elements, uint64_t *descents, uint64_t *key_sum,
                        uint64_t *key_mix);

/* ---- multi-GPU: key-range sharded sort, one rank per GPU (BASELINE.json configs[4]) ---- */
/*
 * No reference counterpart (VkRadixSort is single-GPU); north_star defines the path: shard by key range across the
 * GPUs of one node, ONE all-to-all over xGMI between the local step and the local sorts.  A step on rank g, in its
 * default (hybrid) shape -- the single-GPU hybrid sort with the exchange between its two MSD passes, 28 B/key per GPU:
 *   1. vrs_msd_partition_u32 of the shard: one counting read + the first MSD pass -- the shard grouped by the top byte;
 *   2. one all-gather of every rank's row (top-byte counts, shard size, bucket shift, status) and one all-reduce of the
 *      16384-bin bucket histograms; every rank derives the same byte-aligned splitters (vrs_dist_plan_splitters), its
 *      send and receive counts and where every message lands;
 *   3. `rounds` rounds of grouped send / recv on a second stream (round r = the r-th sub-range of every rank), one
 *      message per (sender, top byte) landing in top-byte order;
 *   4. vrs_msd_finish_u32 (second MSD pass + LDS-local sort) of round r's keys while the later rounds are on the wire.
 *      The sub-ranges are disjoint and ascending: their concatenation is rank g's range in ascending order.
 * Key ranges below 27 bits, shards below 2^16 keys or ranks that probed different key ranges make ALL ranks take the byte
 * shape instead (contract partition pass by the top byte, one message per (sender, round), vrs_sort_keys_u32 per received
 * sub-range; also forced by the environment variable VRS_DIST_SHAPE=byte, which must then be set on every rank).
 * Every decision to leave a step is taken by all ranks from the same gathered rows: a rank that cannot take part (shard
 * above its capacity, a failed local stage) says so in its row, still joins the collectives, and all ranks return together
 * (that rank its own error, the others VRS_ERROR_PEER).  A rank whose step fails LOCALLY outside those stages (a HIP or
 * transport call) returns at once without the collectives it has not reached: over RCCL the caller aborts the
 * communicator, as after any failed rank; the loopback transport marks its hub broken whenever one of ITS calls fails
 * (HIP, mismatch or misuse), so the peers leave their next rendezvous with an error instead of waiting.
 * Keys whose top bytes are too concentrated for byte-aligned ranges (more than 15 % over the even share, or more than the
 * SMALLEST capacity of all ranks: small keys, clustered keys) are cut at SAMPLED key values instead: every rank adds 2048
 * keys of its shard to a pool (a second all-gather), the world * rounds - 1 cut keys are the pool's weighted quantiles,
 * the shard is grouped by range (vrs_range_partition, 12 B/key) and a third all-gather hands out the range prefixes; one
 * message per (sender, round), vrs_sort_keys_u32_ranged per received sub-range.  Only keys with massive ties -- one key
 * VALUE holding more than a rank's share -- still return VRS_ERROR_UNBALANCED (on every rank); so does every
 * concentrated input under VRS_DIST_SAMPLED_SPLITTERS=0 (tests).  The Python orchestration
 * (vkradixsort_amd/distributed.py) adds a gather path for small totals on top of the same entry points.  Blocking for
 * the count exchange and, per round, for the round's plan; the exchange and the sorts complete on the context's stream.
 *
 * The wire is a table of functions.  vrs_dist_create binds RCCL at run time (dlopen): `nccl_comm` is an ncclComm_t of the
 * RCCL copy already in the process; NULL at world size 1 (every transfer is then a device copy).
 * vrs_dist_create_with_transport takes any table: every function returns 0 on success, counts are in uint32 words, `peer`
 * is a rank, `hip_stream` the stream the operation is ordered on; all_gather / all_reduce (sum) are called by every rank
 * with equal counts (recv of all_gather: world * words, in rank order), send / recv only between group_start and
 * group_end, matched in posting order per pair of ranks (the RCCL contract).
 * vrs_dist_loopback_*: an in-process transport -- the ranks are host THREADS of one process sharing a hub, every transfer a
 * device-to-device copy ordered by events (same GPU or peer GPUs of the process).  One process driving several GPUs can
 * use it instead of RCCL; the tests use it to run two ranks on one GPU.
 */
typedef struct vrs_dist_t *vrs_dist;
typedef struct vrs_dist_transport {
    void *user; /* handed back as the first argument of every call */
    int (*all_gather)(void *user, const void *send, void *recv, size_t words, void *hip_stream);
    int (*all_reduce)(void *user, const void *send, void *recv, size_t words, void *hip_stream);
    int (*group_start)(void *user);
    int (*send)(void *user, const void *buf, size_t words, int peer, void *hip_stream);
    int (*recv)(void *user, void *buf, size_t words, int peer, void *hip_stream);
    int (*group_end)(void *user);
    const char *(*error_string)(void *user, int code); /* may be NULL */
} vrs_dist_transport;
int vrs_dist_create(vrs_context ctx, void *nccl_comm, int rank, int world, uint32_t capacity_keys, int rounds,
                    vrs_dist *out_dist);
/* capacity_keys: shard size and receive capacity (the smallest capacity of all ranks bounds every rank's range);
 * rounds: 1 .. 32 / world (clamped).  The table is copied; `user` must outlive the vrs_dist. */
int vrs_dist_create_with_transport(vrs_context ctx, const vrs_dist_transport *transport, int rank, int world,
                                   uint32_t capacity_keys, int rounds, vrs_dist *out_dist);
int vrs_dist_destroy(vrs_dist dist);
/* keys: this rank's shard (num_elements <= capacity_keys; untouched).  *out_keys: a library-owned buffer holding this
 * rank's key range ascending in its first *out_count keys (valid until the next step or vrs_dist_destroy). */
int vrs_dist_sort_keys_u32(vrs_dist dist, vrs_buffer keys, uint32_t num_elements, vrs_buffer *out_keys,
                           uint32_t *out_count);
/* bounds[0] = 0 <= ... <= bounds[parts] = 256: part q owns top bytes [bounds[q], bounds[q+1]); host only */
int vrs_dist_plan_splitters(const uint64_t *counts256, int parts, uint32_t *bounds);
/* The cut keys of a sampled-splitter step (host only): samples[q * per_rank + i] = the i-th of per_rank keys taken at evenly spaced
 * positions of rank q's shard; a sample of rank q stands for shard_sizes[q] / per_rank keys.  splitters[p - 1], 1 <= p < parts, = the
 * first sample in key order at which p / parts of all keys have gone by (ascending; range r = number of splitters <= key). */
int vrs_dist_plan_sampled_splitters(const uint32_t *samples, const uint64_t *shard_sizes, int world, uint32_t per_rank, int parts,
                                    uint32_t *splitters);
const char *vrs_dist_last_error(vrs_dist dist);
/* cumulative: received sub-ranges finished in the hybrid shape / sorted by vrs_sort_keys_u32 after a refused plan; steps
 * that took the byte shape.  Any pointer may be NULL. */
int vrs_dist_stats(vrs_dist dist, uint64_t *hybrid_rounds, uint64_t *fallback_rounds, uint64_t *byte_shape_steps);
/* rounds of byte-shape steps finished by vrs_msd_finish_grouped_u32 (one counting read + second MSD pass + local sort: 20 B/key)
 * instead of a whole ranged sort (28) */
int vrs_dist_grouped_rounds(vrs_dist dist, uint64_t *grouped_rounds);
/* steps that cut the key ranges at sampled key values (top bytes too concentrated for byte-aligned ranges) */
int vrs_dist_splitter_steps(vrs_dist dist, uint64_t *splitter_steps);
typedef struct vrs_dist_loopback_t *vrs_dist_loopback;
int vrs_dist_loopback_create(int world, vrs_dist_loopback *out_hub);
/* the same hub over HOST memory (buffers are host pointers, transfers memcpy, streams ignored, no HIP call): the hub's matching and
 * barriers on a machine without a GPU -- test infrastructure of the sanitizer builds, not a transport for vrs_dist_create_with_transport */
int vrs_dist_loopback_create_host(int world, vrs_dist_loopback *out_hub);
/* fills *out with rank `rank`'s end of the hub; call it on the thread that drives the rank, its device current */
int vrs_dist_loopback_transport(vrs_dist_loopback hub, int rank, vrs_dist_transport *out);
int vrs_dist_loopback_destroy(vrs_dist_loopback hub);
/* A wire that costs something (a model of xGMI's point-to-point links for the SCHEDULE's sake -- how much of the exchange a given number
 * of rounds exposes --, never a measurement): every group of sends / receives holds the receiving rank's stream for (the most bytes one peer
 * sends it in the group) / link_gbps [10^9 bytes per second and link direction] + latency_us, every all-gather / all-reduce for latency_us.
 * 0 / 0 (the default) = the free wire.  Also read from VRS_LOOPBACK_LINK_GBPS / VRS_LOOPBACK_LATENCY_US when the hub is made.  Call
 * between steps. */
int vrs_dist_loopback_set_wire(vrs_dist_loopback hub, double link_gbps, double latency_us);

/* ---- measurement (no reference counterpart) ------------------------- */

typedef enum vrs_kernel_id {
    VRS_KERNEL_HISTOGRAM = 0, /* stage RADIX_SORT_HISTOGRAMS */
    VRS_KERNEL_PREFIX = 1,    /* global digit prefix + per-workgroup offsets (both launches) */
    VRS_KERNEL_SCATTER = 2,   /* stable scatter (the dominant kernel) */
    VRS_KERNEL_SINGLE = 3,    /* single_radixsort */
    VRS_KERNEL_DIGIT_TABLES = 4,     /* one-call sort, large N: the single counting read of all four digits */
    VRS_KERNEL_LOOKBACK_SCATTER = 5, /* one-call sort, large N: scatter pass with decoupled look-back (stable), or -- MSD passes over bare keys --
                                        with reserved places */
    VRS_KERNEL_LOCAL_SORT = 6,       /* one-call sort, hybrid form: every top-14-bit bucket sorted inside LDS */
    VRS_KERNEL_POOL_SAMPLE = 7,      /* one-call sort, pool form: the sample (1/32 of the keys) that sizes the first MSD pass's regions */
    VRS_KERNEL_POOL_PASS_A = 8,      /* one-call sort, pool form: the first MSD pass (pool_pass_a_kernel: reserves in sampled regions) */
    VRS_KERNEL_POOL_PASS_B = 9,      /* one-call sort, pool form: the second MSD pass (pool_pass_b_kernel: scatters into the buckets' slack regions) */
    VRS_KERNEL_COUNT = 10
} vrs_kernel_id;

/* When enabled, every kernel launch carries a (start, stop) hipEvent pair on its own dispatch packet
 * (hipExtLaunchKernel) on the context's stream -- no separate event-record packets. */
int vrs_profile_enable(vrs_context ctx, int enabled);
/* Same, for a subset: bit k of `kernel_mask` selects vrs_kernel_id k (timing only the dominant kernel
 * keeps the instrumentation out of the other launches of a timed region). */
int vrs_profile_enable_mask(vrs_context ctx, uint32_t kernel_mask);
int vrs_profile_reset(vrs_context ctx);
/* Synchronises the stream, then returns launches and summed event time for one kernel id. */
int vrs_profile_query(vrs_context ctx, int kernel_id, uint64_t *launches, double *total_ms);
/* Same for ONE launch: `index` counts the instrumented launches of `kernel_id` since vrs_profile_reset. */
int vrs_profile_query_launch(vrs_context ctx, int kernel_id, uint64_t index, double *ms);

/* Test hook: copies the per-workgroup offset table (uint32[W*256], multi_radixsort.comp:76's
 * global_offsets for every workgroup) computed by the most recent RADIX_SORT stage. */
int vrs_debug_download_offsets(vrs_context ctx, void *host_data, size_t size_bytes);

/* Test hook: checks on the device that one returning LDS atomic hands same-address lanes their
 * pre-values in ascending lane order (what the RANK_ATOMIC scatter variants rely on); returns the
 * number of lanes whose rank differs from the __ballot-based rank over `rounds` rounds per wave. */
int vrs_debug_atomic_rank_selftest(vrs_context ctx, uint32_t rounds, uint32_t seed, uint64_t *mismatches);

/* What the large-N form of the one-call sorts did on this context so far (cumulative): passes run as look-back
 * scatters, passes that fell back to a contract pass (streams too unequal), identity passes left out. */
int vrs_one_call_stats(vrs_context ctx, uint64_t *lookback_passes, uint64_t *fallback_passes, uint64_t *skipped_passes);
/* Look-back passes the one-call sort enqueued a second time: the speculative enqueue (made before the plan was known)
 * left at once because an earlier pass needed another form or because the pass's longest stream did not fit the
 * speculative grid.  Cumulative; diagnostics only. */
int vrs_one_call_relaunched_passes(vrs_context ctx, uint64_t *relaunched_passes);
/* One-call sorts that took the hybrid form (VRS_TUNE_HYBRID).  Cumulative; diagnostics only. */
int vrs_one_call_hybrid_sorts(vrs_context ctx, uint64_t *hybrid_sorts);
/* One-call sorts that, after a fast count (VRS_TUNE_HYBRID_FAST_COUNT) and a plan that refused the hybrid form, started over
 * as LSD sorts with a second counting read.  Cumulative; diagnostics only. */
int vrs_one_call_hybrid_recounts(vrs_context ctx, uint64_t *recounts);

/* WHICH FORM a one-call sort takes (host only, no device needed): the dispatcher's own decision function, so that its table --
 * form x size x kind x settings x what an earlier refusal left behind -- can be walked by a test.  key_bytes 4 or 8, pairs != 0 = with
 * uint32 payloads; knobs[VRS_FORM_KNOB_*] for knob_count entries, a negative entry (or one beyond knob_count) = what a fresh context on
 * a device whose probes passed holds; *form = VRS_FORM_*; memory_out (may be NULL): [0] = the adaptive pool skip, [1] = the 64-bit
 * keys' skip counter as the decision leaves them. */
typedef enum vrs_sort_form {
    VRS_FORM_NONE = 0,     /* nothing to sort */
    VRS_FORM_SINGLE = 1,   /* one single_radixsort launch */
    VRS_FORM_CONTRACT = 2, /* the reference's two stages, pass by pass */
    VRS_FORM_LSD = 3,      /* one counting read + a look-back scatter pass per key byte */
    VRS_FORM_COUNTED = 4,  /* the counted hybrid form */
    VRS_FORM_POOL = 5      /* the pool form (pairs: its stable variant) */
} vrs_sort_form;
typedef enum vrs_form_knob {
    VRS_FORM_KNOB_SINGLE_MAX_KEYS = 0, VRS_FORM_KNOB_ONE_CALL_MIN_KEYS = 1, VRS_FORM_KNOB_HYBRID_MIN_KEYS = 2, VRS_FORM_KNOB_POOL_MIN_KEYS = 3,
    VRS_FORM_KNOB_HYBRID = 4, VRS_FORM_KNOB_POOL = 5, VRS_FORM_KNOB_POOL_PAIRS = 6, VRS_FORM_KNOB_RESERVE = 7, VRS_FORM_KNOB_GROUPS = 8,
    VRS_FORM_KNOB_XCC_MAP_VALID = 9, VRS_FORM_KNOB_ATOMIC_RANK = 10,           /* what the context found out about the device */
    VRS_FORM_KNOB_POOL_SKIP = 11, VRS_FORM_KNOB_POOL_SKIP_N = 12, VRS_FORM_KNOB_WIDE_REFUSED = 13, VRS_FORM_KNOB_WIDE_SKIPPED = 14, /* memory */
    VRS_FORM_KNOB_NO_POOL = 15, VRS_FORM_KNOB_NO_HYBRID = 16,                   /* this sort is a retry after a refusal of that form */
    VRS_FORM_KNOB_COUNT = 17
} vrs_form_knob;
int vrs_sort_form_for(uint32_t num_elements, int key_bytes, int pairs, const int64_t *knobs, int knob_count, int *form, int64_t *memory_out);

/* One-call sorts of bare uint32 keys that took the pool form (the hybrid form without a counting read, VRS_TUNE_MSD_POOL), and
 * sorts whose pool form the plan refused (they ran in the counted form afterwards).  Cumulative; diagnostics only. */
int vrs_one_call_pool_sorts(vrs_context ctx, uint64_t *pool_sorts, uint64_t *pool_refusals);
/* What the pool form would do with num_elements bare uint32 keys (host only, no device needed): *sub_bits = S where the sort has
   256 << S buckets (6 or 7; 0 = the form does not take this size) -- the COUNT of buckets, not the cut: with the default cut a sort of
   16384 buckets runs a first pass by 7 bits and a second by 7, vrs_pool_form_shape_ex reports that --, *bucket_capacity = keys per bucket the local sort it enqueues takes (1789: one wave
   per bucket; 4093 / 7165: 256 threads; 14333: 512), *scratch_bytes = context scratch a sort of this size needs beside the caller's two
   buffers (slack buffer + first-pass overflow room + plan).  Any pointer may be NULL. */
int vrs_pool_form_shape(uint32_t num_elements, uint32_t *sub_bits, uint32_t *bucket_capacity, uint64_t *scratch_bytes);
/* The same with the cut as it runs: pairs != 0 = uint32 key + uint32 payload pairs (the stable pool form: other shapes, and the
   payloads' twins of the slack buffer and the overflow room in *scratch_bytes); top_bits_setting = VRS_TUNE_MSD_POOL_TOP_BITS as set on
   the context (0 = the library's default, 7).  *first_pass_bits / *second_pass_bits = the digits of the two MSD passes (7 + 7 by
   default, 8 + 7 where 32768 buckets are needed; 0 / 0 = the form does not take this size).  Any pointer may be NULL. */
int vrs_pool_form_shape_ex(uint32_t num_elements, int pairs, int top_bits_setting, uint32_t *first_pass_bits, uint32_t *second_pass_bits,
                           uint32_t *bucket_capacity, uint64_t *scratch_bytes);
/* sorts of the pool form whose local sort was enqueued a second time in a larger workgroup shape: the shape is chosen from
   num_elements alone (the form is enqueued blind), and a bucket of skewed keys may hold more than it takes -- no refusal, the bucket
   lies whole in its slack region; the settle asks for the larger shape (one more kernel, one more host round trip) */
int vrs_one_call_pool_retries(vrs_context ctx, uint64_t *retries);
/* The pool form keeps its scratch -- the slack buffer and the first pass's overflow regions, about 1.55 n uint32 slots for the largest sort
   the context has taken in that form (616 MB at 10^8 keys), twice that once pairs took it (vrs_pool_form_shape reports it) -- until the
   context is destroyed.  vrs_context_trim_scratch gives it back to the device now (after settling and waiting for what is on the stream;
   *released_bytes may be NULL); the next pool sort allocates again.  A device with no room for it is no error of a sort: the sort takes a
   form that needs no such scratch (28 / 36 bytes per key instead of 24), vrs_one_call_pool_no_memory counts those sorts, and with the
   adaptive setting (VRS_TUNE_MSD_POOL = 1) the next 15 sorts of that size do not ask again. */
int vrs_context_trim_scratch(vrs_context ctx, uint64_t *released_bytes);
int vrs_one_call_pool_no_memory(vrs_context ctx, uint64_t *sorts);
/* VRS_TUNE_MSD_POOL_REUSE_LAYOUT: *reused = pool sorts that started in the regions of an earlier sort, *stale = those of them whose keys
   did not fit (run again with their own sample).  Either pointer may be NULL. */
int vrs_one_call_pool_layouts(vrs_context ctx, uint64_t *reused, uint64_t *stale);
/* The forms of the one-call sorts that share L2-resident words between workgroups rest on where the blocks of a launch run
   ("block b on the XCC of place b % 8 of the probed order"), which the context probes when it is created.  Observed on MI355X: the
   dispatcher's round-robin starts at an XCC of the hardware queue's own, and a HIP stream may move to another queue -- the probed
   order is then rotated.  Every kernel checks (HW_REG_XCC_ID) and stays exact; the first workgroups that find themselves elsewhere
   say so, and the next vrs_sort_* call probes again.  *reprobes = how often that happened; *xcc_map = the probed order (byte x =
   the XCC of the blocks with index % 8 == x); *valid = the probe found the b % 8 rule intact.  Any pointer may be NULL. */
int vrs_debug_xcc_placement(vrs_context ctx, uint64_t *reprobes, uint64_t *xcc_map, int *valid);

/* Ranking method in effect: 1 = __ballot match-any, 2 = returning LDS atomics. */
int vrs_rank_mode(vrs_context ctx);

/* Tuning knobs (performance only, never results). */
typedef enum vrs_tuning_key {
    VRS_TUNE_XCD_REMAP = 0,      /* 1 (default): consecutive tiles share an XCD's L2 in the scatter */
    VRS_TUNE_SCATTER_VARIANT = 1, /* 0 (default): chosen from B; else ITEMS*1000 + WAVES*10 + RANK */
    VRS_TUNE_FUSED_PREFIX = 2,    /* 1 (default): single-launch prefix (chunk sums exchanged through tagged granules) */
    VRS_TUNE_RANK_MODE = 3,       /* 0 (default) auto: LDS-atomic ranking if the device self-test passed at
                                     context creation, else __ballot ranking; 1 force ballot; 2 force atomic */
    VRS_TUNE_ONE_CALL_MIN_KEYS = 4, /* vrs_sort_keys_u32 / _u64 / vrs_sort_pairs_u32 count all four digits in ONE read
                                     and scatter with decoupled look-back (36 instead of 48 bytes per key) from this many
                                     keys on; 0 = never (always the contract passes).  Default 2^13: it is the faster form at every
                                     size above the single-launch threshold (profiles/r02_one_call_crossover.csv). */
    VRS_TUNE_DEBUG_MISPLACE_STREAMS = 5, /* test hook (default 0): run every other tile of a look-back stream behind a
                                     different XCD's L2, i.e. without the placement the fast hand-off relies on */
    VRS_TUNE_LOOKBACK_SPIN_BUDGET = 6, /* polls of a predecessor's unpublished look-back row before a tile stops waiting
                                     and counts its stream's earlier keys itself (default 4096, about 2-4 ms) */
    VRS_TUNE_DEBUG_HOLD_TILE = 7,  /* test hook (default -1 = off): this tile of every look-back stream never publishes
                                     its counts, so its successors must run out of spin budget and recount */
    VRS_TUNE_DIGIT_TABLE_GROUPS = 8, /* groups per pass of the one-call sort's counting read: 8, 16, 32, or 0 (default):
                                     8 below 2^26 keys, 32 from there on */
    VRS_TUNE_HYBRID = 11,          /* vrs_sort_keys_u32 / vrs_sort_pairs_u32 / vrs_sort_keys_u64 of large inputs: 1 (default) = the 28-byte-per-key hybrid form (MSD
                                     partition by the top 14 bits in two look-back passes + an LDS-local sort of every
                                     bucket) whenever every bucket fits a workgroup's LDS, else the four LSD passes (decided
                                     on the device from the same counting read); 0 = always the LSD passes */
    VRS_TUNE_HYBRID_MIN_KEYS = 12, /* the hybrid form is considered from this many keys on, from 5/8 as many pairs and from
                                     half as many 64-bit keys; 0 (default): the measured crossovers -- 1.3 * 10^7 keys, 2.5 * 10^7
                                     pairs, 2 * 10^7 64-bit keys.  Never below 2^22 elements */
    VRS_TUNE_HYBRID_FAST_COUNT = 13, /* the counting read of a sort the hybrid form may take: 0 = always counts the LSD
                                     tables beside the bucket histogram (a refusal costs nothing extra); 2 = counts only
                                     the bucket histogram when the probed key range allows the hybrid form (1 LDS add per
                                     key instead of 5: about 20 us at 10^8 keys) and starts over as an LSD sort, with a second
                                     counting read, if the plan refuses; 1 (default) = adaptive: like 2 while the context's
                                     previous hybrid-capable sort of the same kind (keys / pairs) took the hybrid form, like 0 after a
                                     refusal (64-bit keys: always 2; after a refusal every 16th such sort tries again) */
    VRS_TUNE_FUSED_PLAN = 10,      /* 1: the last workgroup of the one-call sort's counting read turns the digit tables
                                     into the plan; 0 (default): a separate single-workgroup plan kernel (measured a
                                     tie at 10^7 and 10^8 keys) */
    VRS_TUNE_SINGLE_MAX_KEYS = 9,  /* vrs_sort_keys_u32 runs up to this many keys as ONE single_radixsort launch (one
                                     workgroup, four passes) instead of twelve launch-bound multi-block launches;
                                     0 = never.  Default 4096 (measured crossover, profiles/r02_small_n_crossover.csv) */
    VRS_TUNE_ASYNC_SORT = 14,      /* 1 (default on a context with its own stream): vrs_sort_keys_u32 / _pairs_u32 / _keys_u64 only
                                     enqueue and return at once; vrs_sort_settle (or any entry point that settles) finishes what the
                                     plan asks for.  0 (default on a borrowed stream): they return once the plan's head has reached
                                     the host and the whole sort is on the stream */
    VRS_TUNE_PLAN_WAIT_MS = 15,    /* longest wait for a plan's head in milliseconds (default 60000; 0 = no limit) */
    VRS_TUNE_MSD_RESERVE = 16,     /* the two MSD passes of the hybrid form over BARE keys reserve their output ranges with one L2-local atomic
                                      add per tile and digit instead of a decoupled look-back (the order inside a bucket is free there;
                                      payloads always take the stable look-back).  1 (default; 2 means the same): on; 0: look-back
                                      everywhere */
    VRS_TUNE_MSD_POOL = 17,        /* the hybrid form of BARE uint32 keys without its counting read (24 instead of 28 bytes per key): a sample of
                                      1/32 of the keys sizes a region of the partner buffer per (input slice, top byte), the first MSD
                                      pass reserves its output there, the second pass scatters by the next 6 or 7 bits into per-bucket
                                      regions of a context-owned slack buffer (about 1.5 n keys, sized from a sample of the first pass's output),
                                      the local sort reads every bucket in one piece and writes it to its final place.  A sort a verdict
                                      refuses (a region the sample misjudged, a bucket too large) starts over in the counted form with its input untouched.  1 (default) = adaptive: after a refusal the next 15
                                      such sorts of the context take the counted form; 2 = always tried; 0 = never.  Needs
                                      VRS_TUNE_MSD_RESERVE != 0. */
    VRS_TUNE_MSD_POOL_MIN_KEYS = 18, /* the pool form is considered from this many keys on (default and floor 2^22: with one wave per small bucket it beats the
                                        LSD passes from there on, profiles/labs/r05_pool_form.txt section 6) */
    VRS_TUNE_MSD_POOL_SUB_BITS = 20, /* S where a pool sort of bare keys has 256 << S buckets: 0 (default) = by size (6 while the buckets fit a 256-thread local sort, about
                                        1.1 * 10^8 uniform keys, else 7), 6, 7 or 8 (8 with VRS_TUNE_MSD_POOL_TOP_BITS 8: 65536 buckets of one wave each, a lab
                                        setting -- profiles/labs/r06_cut_8_8.txt) */
    VRS_TUNE_MSD_POOL_REUSE_LAYOUT = 22, /* (2: what follows; 1, the default: also the buckets' slack regions are kept -- the plan kernel of such a sort
                                       samples nothing; verified by the second pass like the first pass's regions by the first.)
                                       1 (default): a pool sort of the same size and key floor as the context's last TAKEN one runs its first
                                       pass in the regions that sort's sample laid out -- no sample and no layout kernel (13 us and two launch gaps
                                       at 10^8 keys).  Verified like any layout: keys it does not fit (another distribution, another key range)
                                       flag the sort, which then runs again with a sample of its own (vrs_one_call_pool_layouts counts both).
                                       0: every sort samples */
    VRS_TUNE_MSD_POOL_TOP_BITS = 24, /* how the pool form cuts a sort's 16384 buckets between its two passes: 7 (default) = 128 x 128, a first pass by 7 bits
                                        and a second by 7; 8 = 256 x 64 (the cut until late in round 5); 6 = 64 x 256.  Sorts whose second pass takes
                                        7 bits of 256 top bytes anyway (beyond about 1.1e8 keys) are not affected */
    VRS_TUNE_MSD_POOL_PAIRS = 23, /* 1 (default): uint32 key + uint32 payload pairs may take the pool form too -- its STABLE variant: a tile's place in
                                     a sampled region is its rank there (decoupled look-back, one chain per input slice / per top byte) instead
                                     of a reservation, so equal keys keep their input order; 48 instead of 52 bytes per pair.  0: pairs always
                                     take the counted form */
    VRS_TUNE_MSD_POOL_PAIRS_PACKED = 26, /* the local sort of a pool sort of pairs, per bucket: -1 (default) = by size, 1 = always, 0 = never in its PACKED form --
                                     inside a bucket a key's high bits are the bucket's, so a pair is sorted as ONE word (the key's low 18 bits | the
                                     pair's place in the bucket as read), the payloads wait in registers and take the words' LDS array once the words
                                     are sorted: 43 instead of 70 KB of LDS, three workgroups per CU instead of two.  5-9 % faster from 2.6e7 to 8e7
                                     pairs; level at 10^8 and 2e8 (buckets of 12-13 rows), where the form that carries the payloads through both
                                     passes stays (profiles/labs/r06_pairs_packed.txt) */
    VRS_TUNE_DEBUG_POOL_NO_MEMORY = 25, /* test hook: the next `value` allocations of the pool form's scratch fail as if the device were full */
    VRS_TUNE_DEBUG_XCC_ROTATE = 21, /* test hook: run the placement probe again and rotate its result by `value` places (0 .. 7), as if the probe had
                                       run on another hardware queue than the sorts do (the dispatcher starts every queue's round-robin at its
                                       own XCC, and a stream may move between queues): the pool form's passes take their work lists by the XCC
                                       they run on and stay exact AND fast; the look-back streams and the counted form's reservations fall back
                                       to their placement-independent routes (exact, slower) */
    VRS_TUNE_DEBUG_XCC_STRAY_BLOCK = 19 /* test hook: run the placement probe again and pretend block `value` (0 .. 4095) of it ran on
                                       another XCC: block b -> XCC (b % 8) then holds for most blocks only, and every form that leans on
                                       it (the one-call sorts' look-back streams, reservation, the hybrid and pool forms, vrs_msd_*) is
                                       switched off -- the sorts run the contract stages, which share nothing between workgroups but
                                       agent-scope data.  A negative value probes again without pretending */
} vrs_tuning_key;
int vrs_set_tuning(vrs_context ctx, int key, int value);

const char *vrs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VKRADIXSORT_AMD_H */
