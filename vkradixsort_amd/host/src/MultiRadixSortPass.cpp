#include "MultiRadixSortPass.h"

namespace engine {

static_assert(sizeof(MultiRadixSortPass::PushConstants) == sizeof(vrs_push_constants) &&
                  sizeof(MultiRadixSortPass::PushConstantsHistograms) == sizeof(vrs_push_constants),
              "push-constant blocks must stay the reference's 16-byte std430 layout");

void MultiRadixSortPass::recordCommands() {
    vrs_context ctx = m_gpuContext->handle();
    vrs_push_constants pc{};

    // stage RADIX_SORT_HISTOGRAMS: set 0, b0 = keys in, b1 = histograms
    pc = {m_pushConstantsHistogram.g_num_elements, m_pushConstantsHistogram.g_shift,
          m_pushConstantsHistogram.g_num_workgroups, m_pushConstantsHistogram.g_num_blocks_per_workgroup};
    const auto histograms = m_sort64Bit ? vrs_multi_radixsort_histograms_u64 : vrs_multi_radixsort_histograms;
    m_gpuContext->check(histograms(ctx, boundBuffer(RADIX_SORT_HISTOGRAMS, 0)->getBuffer(),
                                   boundBuffer(RADIX_SORT_HISTOGRAMS, 1)->getBuffer(), &pc),
                        "Failed to submit compute command buffer!");

    // stage RADIX_SORT: set 1, b0 = in, b1 = out, b2 = histograms (b3/b4 = payload in/out, extension)
    pc = {m_pushConstants.g_num_elements, m_pushConstants.g_shift, m_pushConstants.g_num_workgroups,
          m_pushConstants.g_num_blocks_per_workgroup};
    int status;
    const auto sortPairs = m_sort64Bit ? vrs_multi_radixsort_pairs_u64 : vrs_multi_radixsort_pairs;
    const auto sortKeys = m_sort64Bit ? vrs_multi_radixsort_u64 : vrs_multi_radixsort;
    if (m_sortPairs)
        status = sortPairs(ctx, boundBuffer(RADIX_SORT, 0)->getBuffer(),
                                           boundBuffer(RADIX_SORT, 1)->getBuffer(),
                                           boundBuffer(RADIX_SORT, 3)->getBuffer(),
                                           boundBuffer(RADIX_SORT, 4)->getBuffer(),
                                           boundBuffer(RADIX_SORT, 2)->getBuffer(), &pc);
    else
        status = sortKeys(ctx, boundBuffer(RADIX_SORT, 0)->getBuffer(),
                                     boundBuffer(RADIX_SORT, 1)->getBuffer(), boundBuffer(RADIX_SORT, 2)->getBuffer(),
                                     &pc);
    m_gpuContext->check(status, "Failed to submit compute command buffer!");
}

}  // namespace engine
