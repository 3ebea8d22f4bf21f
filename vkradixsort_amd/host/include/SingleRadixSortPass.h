// engine::SingleRadixSortPass (reference: singleradixsort/include/SingleRadixSortPass.h:7-29).
#pragma once

#include "engine/passes/ComputePass.h"

namespace engine {

class SingleRadixSortPass : public ComputePass {
public:
    explicit SingleRadixSortPass(GPUContext *gpuContext) : ComputePass(gpuContext) {}

    enum ComputeStage {
        RADIX_SORT = 0,
    };

    struct PushConstants {
        uint32_t g_num_elements;
    };
    PushConstants m_pushConstants{};

protected:
    [[nodiscard]] uint32_t stageCount() const override { return 1; }
    void recordCommands() override;
};

}  // namespace engine
