// engine::MultiRadixSortPass -- the drop-in boundary object of the multi_radixsort path
// (reference: multiradixsort/include/MultiRadixSortPass.h:7-41, src/MultiRadixSortPass.cpp:10-20).
// Same stage enum, same two push-constant structs with the same field names/order, same public members.
#pragma once

#include "engine/passes/ComputePass.h"

namespace engine {

class MultiRadixSortPass : public ComputePass {
public:
    explicit MultiRadixSortPass(GPUContext *gpuContext) : ComputePass(gpuContext) {}

    enum ComputeStage {  // also the descriptor-set numbers of the two stages
        RADIX_SORT_HISTOGRAMS = 0,
        RADIX_SORT = 1,
    };

    struct PushConstantsHistograms {
        uint32_t g_num_elements;
        uint32_t g_shift;
        uint32_t g_num_workgroups;
        uint32_t g_num_blocks_per_workgroup;
    };
    PushConstantsHistograms m_pushConstantsHistogram{};

    struct PushConstants {
        uint32_t g_num_elements;
        uint32_t g_shift;
        uint32_t g_num_workgroups;
        uint32_t g_num_blocks_per_workgroup;
    };
    PushConstants m_pushConstants{};

    // Build extension (BASELINE.json config 4): when set, bindings (1,3)/(1,4) carry payload in/out.
    bool m_sortPairs = false;
    // The reference's SORT_64_BIT switch (MultiRadixSort.h:10-18): buffers hold uint64 keys, g_shift runs to 56.
    bool m_sort64Bit = false;

protected:
    [[nodiscard]] uint32_t stageCount() const override { return 2; }
    // histograms stage, then sort stage; stream order stands in for the two W->R pipeline barriers
    void recordCommands() override;
};

}  // namespace engine
