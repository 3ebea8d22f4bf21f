// engine::ComputePass -- launch-shape bookkeeping and submit (reference:
// engine/include/engine/passes/ComputePass.h:6-117).
#pragma once

#include <vector>

#include "Pass.h"

namespace engine {

struct Extent3D {  // VkExtent3D
    uint32_t width = 0, height = 0, depth = 0;
};

class ComputePass : public Pass {
public:
    explicit ComputePass(GPUContext *gpuContext) : Pass(gpuContext) {}

    void create() override;

    // workgroup count per axis = ceil(global size / workgroup size); all kernels of this path keep the
    // reference shaders' contract workgroup size of 256x1x1
    void setGlobalInvocationSize(uint32_t stageIndex, uint32_t width, uint32_t height, uint32_t depth);
    static Extent3D getDispatchSize(uint32_t width, uint32_t height, uint32_t depth, Extent3D workGroupSize);
    [[nodiscard]] Extent3D getWorkGroupCount(uint32_t stageIndex) const;

    // asynchronous: enqueues the stages on the context's stream and returns a token for chaining
    Semaphore execute(Semaphore awaitBeforeExecution) override;

protected:
    [[nodiscard]] virtual uint32_t stageCount() const = 0;
    virtual void recordCommands() = 0;

private:
    std::vector<Extent3D> m_workGroupCounts;
    uint64_t m_submitCounter = 0;
};

}  // namespace engine
