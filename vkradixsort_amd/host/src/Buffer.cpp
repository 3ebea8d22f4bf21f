#include "engine/core/Buffer.h"

#include <utility>

namespace engine {

Buffer::Buffer(GPUContext *gpuContext, BufferSettings settings)
    : m_gpuContext(gpuContext), m_bufferSettings(std::move(settings)) {
    m_gpuContext->check(vrs_buffer_create(m_gpuContext->handle(), m_bufferSettings.m_sizeBytes, &m_buffer),
                        "Failed to create buffer!");
}

Buffer::Buffer(GPUContext *gpuContext, BufferSettings settings, void *devicePointer)
    : m_gpuContext(gpuContext), m_bufferSettings(std::move(settings)) {
    m_gpuContext->check(
        vrs_buffer_wrap(m_gpuContext->handle(), devicePointer, m_bufferSettings.m_sizeBytes, &m_buffer),
        "Failed to wrap device memory!");
}

Buffer::~Buffer() { release(); }

void Buffer::release() {
    if (m_buffer) {
        vrs_buffer_release(m_buffer);
        m_buffer = nullptr;
    }
}

std::shared_ptr<Buffer> Buffer::fillDeviceWithStagingBuffer(GPUContext *gpuContext, const BufferSettings &settings,
                                                            const void *data) {
    auto buffer = std::make_shared<Buffer>(gpuContext, settings);
    gpuContext->check(vrs_buffer_upload(gpuContext->handle(), buffer->m_buffer, data, settings.m_sizeBytes),
                      "Failed to upload buffer!");
    return buffer;
}

void Buffer::downloadWithStagingBuffer(void *data) {
    m_gpuContext->check(vrs_buffer_download(m_gpuContext->handle(), getBuffer(), data, m_bufferSettings.m_sizeBytes),
                        "Failed to download buffer!");
}

void Buffer::copyFrom(Buffer &source) {
    m_gpuContext->check(
        vrs_buffer_copy(m_gpuContext->handle(), getBuffer(), source.getBuffer(), m_bufferSettings.m_sizeBytes),
        "Failed to copy buffer!");
}

uint64_t Buffer::getDeviceAddress() const { return reinterpret_cast<uint64_t>(vrs_buffer_device_ptr(getBuffer())); }

vrs_buffer Buffer::getBuffer() const {
    if (!m_buffer) throw std::runtime_error("Buffer was released!");
    return m_buffer;
}

}  // namespace engine
