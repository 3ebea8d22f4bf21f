// vrs_capi_contract.hip -- the C ABI, part 2 of 5: the reference's stages (RADIX_SORT_HISTOGRAMS, RADIX_SORT: MultiRadixSortPass.h:14-17)
// with their push constants, single_radixsort, the range partition and the digit-offset hooks of the multi-GPU step.
#include "vrs_host.hpp"

using namespace vrsh;

namespace vrsh {

int run_sort_stage(vrs_context ctx, vrs_buffer keys_in, vrs_buffer keys_out, vrs_buffer values_in,
                   vrs_buffer values_out, vrs_buffer histograms, const vrs_push_constants *pc, bool pairs,
                   int key_bytes) {
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

int run_histogram_stage(vrs_context ctx, vrs_buffer keys_in, vrs_buffer histograms,
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
}  // namespace vrsh

extern "C" {

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
}  // extern "C"

namespace vrsh {

int ensure_sort_hist(vrs_context ctx, uint32_t workgroups) {
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
int contract_pass(vrs_context ctx, vrs_buffer kin, vrs_buffer kout, vrs_buffer vin, vrs_buffer vout,
                         vrs_push_constants *pc, uint32_t shift, int key_bytes) {
    pc->g_shift = shift;
    int rc = run_histogram_stage(ctx, kin, ctx->sort_hist, pc, key_bytes);
    if (rc) return rc;
    return run_sort_stage(ctx, kin, kout, vin, vout, ctx->sort_hist, pc, vin != nullptr, key_bytes);
}
}  // namespace vrsh

extern "C" {

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
}  // extern "C"
