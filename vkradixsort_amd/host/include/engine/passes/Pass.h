// engine::Pass -- the part of the reference's generic pass the sort path programs against
// (engine/include/engine/passes/Pass.h:11-352): create/release, the two setStorageBuffer overloads and
// execute.  Descriptor sets become a table (descriptor copy, set, binding) -> Buffer*.
#pragma once

#include <cstdint>
#include <map>
#include <utility>
#include <vector>

#include "engine/core/Buffer.h"
#include "engine/core/GPUContext.h"

namespace engine {

class Pass {
public:
    explicit Pass(GPUContext *gpuContext) : m_gpuContext(gpuContext) {}
    virtual ~Pass() = default;

    virtual Semaphore execute(Semaphore awaitBeforeExecution) = 0;

    virtual void create();
    virtual void release();

    // bind in EVERY descriptor copy (Pass.h:54-79)
    void setStorageBuffer(uint32_t set, uint32_t binding, Buffer *buffer);
    // bind in ONE descriptor copy (Pass.h:81-104); the copy in use is GPUContext::getActiveIndex()
    void setStorageBuffer(uint32_t multiBufferedIndex, uint32_t set, uint32_t binding, Buffer *buffer);

protected:
    GPUContext *m_gpuContext;  // non-owning, like the reference (Pass.h:111)
    bool m_created = false;

    // buffer bound at (set, binding) in the live descriptor copy; throws if nothing is bound
    Buffer *boundBuffer(uint32_t set, uint32_t binding) const;

private:
    std::vector<std::map<std::pair<uint32_t, uint32_t>, Buffer *>> m_bindings;
};

}  // namespace engine
