#include "engine/core/GPUContext.h"

#include <cstdlib>
#include <string>

namespace engine {

GPUContext::GPUContext(uint32_t requiredQueueFamilies, int deviceOrdinal)
    : m_deviceOrdinal(deviceOrdinal), m_requiredQueueFamilies(requiredQueueFamilies) {
    if (m_deviceOrdinal < 0) {
        const char *env = std::getenv("VRS_DEVICE");
        m_deviceOrdinal = env ? std::atoi(env) : 0;
    }
}

GPUContext::~GPUContext() { shutdown(); }

void GPUContext::init() {
    if (m_context) return;
    const int status = vrs_context_create(m_deviceOrdinal, &m_context);
    if (status != VRS_OK) {
        m_context = nullptr;
        throw std::runtime_error(std::string("Failed to create GPU context: ") + vrs_last_error(nullptr));
    }
    m_activeIndex = 0;
}

void GPUContext::shutdown() {
    if (m_context) {
        vrs_context_destroy(m_context);
        m_context = nullptr;
    }
}

void GPUContext::waitIdle() { check(vrs_queue_wait_idle(handle()), "Failed to wait for the compute queue"); }

vrs_context GPUContext::handle() const {
    if (!m_context) throw std::runtime_error("GPUContext is not initialised (call init())");
    return m_context;
}

void GPUContext::check(int status, const char *what) const {
    if (status == VRS_OK) return;
    throw std::runtime_error(std::string(what) + ": " + vrs_last_error(m_context));
}

std::string GPUContext::deviceName() const {
    char name[256] = {0};
    int cus = 0;
    uint64_t mem = 0;
    check(vrs_device_info(handle(), name, sizeof name, &cus, &mem), "Failed to query the device");
    return std::string(name) + ", " + std::to_string(cus) + " CUs, " + std::to_string(mem >> 30) + " GiB";
}

}  // namespace engine
