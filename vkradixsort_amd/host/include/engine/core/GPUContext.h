// engine::GPUContext over the HIP C ABI.
// Keeps the API surface of the reference's Vulkan context that the radix-sort path uses
// (engine/include/engine/core/GPUContext.h:15-111: ctor(requiredQueueFamilies), init, shutdown,
// getMultiBufferedCount, getActiveIndex, incrementActiveIndex, m_activeIndex) and replaces its body:
// instance / physical-device / queue creation become one vrs_context (HIP device + stream).
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>

#include "vkradixsort_amd.h"

namespace engine {

// Shape of engine/include/engine/core/Queues.h:13-17 so `GPUContext gpu(COMPUTE_FAMILY | TRANSFER_FAMILY)`
// keeps compiling; HIP has one in-order stream doing both jobs.
struct Queues {
    enum QueueFamilies : uint32_t { GRAPHICS_FAMILY = 1u, COMPUTE_FAMILY = 2u, TRANSFER_FAMILY = 4u, PRESENT_FAMILY = 8u };
    enum Queue { GRAPHICS, COMPUTE, TRANSFER, PRESENT };
};

// What the reference returns from Pass::execute as a VkSemaphore: an opaque token to chain submits.
// One in-order HIP stream already serialises them, so the token only preserves the call shape.
using Semaphore = uint64_t;
constexpr Semaphore NULL_SEMAPHORE = 0;  // VK_NULL_HANDLE

class GPUContext {
public:
    // deviceOrdinal < 0: take $VRS_DEVICE, else device 0.  Never prompts on stdin
    // (the reference does for >1 device, GPUContext.cpp:167-174).
    explicit GPUContext(uint32_t requiredQueueFamilies = Queues::COMPUTE_FAMILY | Queues::TRANSFER_FAMILY,
                        int deviceOrdinal = -1);
    virtual ~GPUContext();

    virtual void init();
    virtual void shutdown();

    uint32_t m_activeIndex = 0;

    [[nodiscard]] uint32_t getMultiBufferedCount() const { return MAX_FRAMES_IN_FLIGHT; }
    [[nodiscard]] uint32_t getActiveIndex() const { return m_activeIndex; }
    void incrementActiveIndex() { m_activeIndex = (m_activeIndex + 1) % MAX_FRAMES_IN_FLIGHT; }

    // vkQueueWaitIdle(m_queues->getQueue(Queues::COMPUTE)) -- the path's only blocking sync
    void waitIdle();

    [[nodiscard]] vrs_context handle() const;
    // every failure is a std::runtime_error, as in the reference
    void check(int status, const char *what) const;
    [[nodiscard]] std::string deviceName() const;

private:
    vrs_context m_context = nullptr;
    int m_deviceOrdinal;
    uint32_t m_requiredQueueFamilies;
    static constexpr uint32_t MAX_FRAMES_IN_FLIGHT = 2;  // GPUContext.h:110
};

}  // namespace engine
