// vrs_kernels.h -- launch wrappers of the gfx950 kernels (internal; the public surface is
// include/vkradixsort_amd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace vrs {

// Scratch owned by the context: offsets[W*256] and chunk_sums[G*256].
struct PrefixScratch {
    uint32_t *offsets = nullptr;
    uint32_t *chunk_sums = nullptr;
};

// tiles per chunk for the two-level prefix: smallest power of two C with C*C >= W
uint32_t prefix_chunk_tiles(uint32_t num_workgroups);

hipError_t launch_histograms(hipStream_t stream, const uint32_t *keys_in, uint32_t *hist,
                             uint32_t n, uint32_t shift, uint32_t W, uint32_t B);

hipError_t launch_prefix(hipStream_t stream, const uint32_t *hist, const PrefixScratch &scratch,
                         uint32_t W);

hipError_t launch_scatter(hipStream_t stream, const uint32_t *keys_in, uint32_t *keys_out,
                          const uint32_t *values_in, uint32_t *values_out, const uint32_t *offsets,
                          uint32_t n, uint32_t shift, uint32_t W, uint32_t B, bool xcd_remap);

hipError_t launch_single(hipStream_t stream, uint32_t *buffer0, uint32_t *buffer1, uint32_t n);

}  // namespace vrs
