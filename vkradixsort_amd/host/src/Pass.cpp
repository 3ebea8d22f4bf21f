#include "engine/passes/ComputePass.h"

#include <string>

namespace engine {

void Pass::create() {
    m_bindings.assign(m_gpuContext->getMultiBufferedCount(), {});
    m_created = true;
}

void Pass::release() {
    m_bindings.clear();
    m_created = false;
}

void Pass::setStorageBuffer(uint32_t set, uint32_t binding, Buffer *buffer) {
    if (!m_created) throw std::runtime_error("Pass was not created!");
    for (auto &copy : m_bindings) copy[{set, binding}] = buffer;
}

void Pass::setStorageBuffer(uint32_t multiBufferedIndex, uint32_t set, uint32_t binding, Buffer *buffer) {
    if (!m_created) throw std::runtime_error("Pass was not created!");
    if (multiBufferedIndex >= m_bindings.size()) throw std::runtime_error("multiBufferedIndex out of range!");
    m_bindings[multiBufferedIndex][{set, binding}] = buffer;
}

Buffer *Pass::boundBuffer(uint32_t set, uint32_t binding) const {
    const auto &copy = m_bindings.at(m_gpuContext->getActiveIndex());
    const auto it = copy.find({set, binding});
    if (it == copy.end() || it->second == nullptr)
        throw std::runtime_error("No storage buffer bound at (set " + std::to_string(set) + ", binding " +
                                 std::to_string(binding) + ")!");
    return it->second;
}

void ComputePass::create() {
    Pass::create();
    m_workGroupCounts.assign(stageCount(), Extent3D{});
}

Extent3D ComputePass::getDispatchSize(uint32_t width, uint32_t height, uint32_t depth, Extent3D workGroupSize) {
    const auto ceilDiv = [](uint32_t a, uint32_t b) { return a / b + (a % b ? 1u : 0u); };
    return {ceilDiv(width, workGroupSize.width), ceilDiv(height, workGroupSize.height),
            ceilDiv(depth, workGroupSize.depth)};
}

void ComputePass::setGlobalInvocationSize(uint32_t stageIndex, uint32_t width, uint32_t height, uint32_t depth) {
    m_workGroupCounts.at(stageIndex) = getDispatchSize(width, height, depth, Extent3D{VRS_WORKGROUP_SIZE, 1, 1});
}

Extent3D ComputePass::getWorkGroupCount(uint32_t stageIndex) const { return m_workGroupCounts.at(stageIndex); }

Semaphore ComputePass::execute(Semaphore awaitBeforeExecution) {
    (void)awaitBeforeExecution;  // submits on one in-order stream are already chained
    if (!m_created) throw std::runtime_error("Failed to submit compute pass: pass was not created!");
    recordCommands();
    return ++m_submitCounter;
}

}  // namespace engine
