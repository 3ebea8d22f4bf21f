// engine::Buffer over the HIP C ABI (reference: engine/include/engine/core/Buffer.h:14-177).
// Device-local allocation owned by the object; upload / download are the synchronous staging copies.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>

#include "GPUContext.h"

namespace engine {

class Buffer {
public:
    struct BufferSettings {
        size_t m_sizeBytes = 0;            // the reference's field is uint32_t (Buffer.h:17): 4 GiB cap lifted
        uint32_t m_bufferUsages = 0;       // VkBufferUsageFlags: accepted, meaningless for HIP
        uint32_t m_memoryProperties = 0;   // VkMemoryPropertyFlags: always device-local HBM
        std::string m_name = "undefined";
    };

    Buffer(GPUContext *gpuContext, BufferSettings settings);
    // wraps caller-owned device memory (hipMalloc'ed elsewhere); release() will not free it
    Buffer(GPUContext *gpuContext, BufferSettings settings, void *devicePointer);
    ~Buffer();
    Buffer(const Buffer &) = delete;
    Buffer &operator=(const Buffer &) = delete;

    void release();  // idempotent

    // upload: allocate device-local + synchronous H2D of settings.m_sizeBytes from `data`
    static std::shared_ptr<Buffer> fillDeviceWithStagingBuffer(GPUContext *gpuContext, const BufferSettings &settings,
                                                               const void *data);
    // synchronous D2H of the whole buffer into `data`
    void downloadWithStagingBuffer(void *data);
    // stream-ordered device copy (re-arming inputs between timed repetitions)
    void copyFrom(Buffer &source);

    [[nodiscard]] size_t getSizeBytes() const { return m_bufferSettings.m_sizeBytes; }
    [[nodiscard]] uint64_t getDeviceAddress() const;
    [[nodiscard]] vrs_buffer getBuffer() const;  // the handle stage calls take (reference: VkBuffer)

private:
    GPUContext *m_gpuContext;
    vrs_buffer m_buffer = nullptr;
    BufferSettings m_bufferSettings;
};

}  // namespace engine
