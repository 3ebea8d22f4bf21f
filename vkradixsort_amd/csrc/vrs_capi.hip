// vrs_capi.hip -- the C ABI declared in include/vkradixsort_amd.h, part 1 of 5: contexts, buffers, the helpers every part shares
// (errors, scratch, profiling events, the placement probe), measurement, tuning and diagnostics.  Host side only.
// There is NO CPU fallback: without a HIP device every entry point fails with VRS_ERROR_NO_DEVICE.
// The other parts: vrs_capi_contract.hip (the reference's two stages, single_radixsort, range partition), vrs_capi_sort.hip (the one-call
// sorts: dispatcher, enqueue, settle), vrs_capi_pool.hip (the pool form's host side), vrs_capi_msd.hip (the halves of the hybrid form the
// multi-GPU step calls).  vrs_host.hpp holds the context and what the parts share.
#include "vrs_host.hpp"

using namespace vrsh;

namespace vrsh {

thread_local std::string g_global_error;
// Contexts that are alive (vrs_context_create* ... vrs_context_destroy): a buffer may outlive its context (host-language finalisers
// run in any order), so vrs_buffer_release asks here before it touches buf->ctx.
std::mutex g_live_mutex;
std::set<vrs_context> g_live_contexts;

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

int check_push_constants(vrs_context ctx, const vrs_push_constants *pc, int key_bytes) {
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
int probe_xcc_map(vrs_context ctx, int stray_block) {
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
uint32_t *drift_word(vrs_context ctx) { return ctx->os_host_head_dev ? &ctx->os_host_head_dev->drift : nullptr; }
// Before a one-call sort is enqueued: did workgroups of earlier sorts find themselves on other XCCs than the probe said (the stream
// moved to another hardware queue, whose round-robin starts elsewhere)?  Then the probe is run again -- once; the sorts in between
// were exact, on their placement-independent routes.
int reprobe_if_drifted(vrs_context ctx) {
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
}  // namespace vrsh

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

int vrs_queue_wait_idle(vrs_context ctx) {
    if (!ctx) return fail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context is NULL");
    VRS_HIP(ctx, hipSetDevice(ctx->device));
    int rc = settle_pending(ctx);
    if (rc) return rc;
    VRS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VRS_OK;
}

int vrs_context_device(vrs_context ctx) { return ctx ? ctx->device : -1; }

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
        case VRS_TUNE_MSD_POOL_PAIRS_PACKED:
            if (value < -1 || value > 1) return fail(ctx, VRS_ERROR_INVALID_ARGUMENT, "the pairs' packed local sort: -1 (by size), 0 (never) or 1 (always)");
            ctx->os_pool_pairs_packed = value;
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
