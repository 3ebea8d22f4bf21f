// engine::SingleRadixSort (reference: singleradixsort/include/SingleRadixSort.h, src/SingleRadixSort.cpp:5-126).
#pragma once

#include <memory>
#include <vector>

#include "SingleRadixSortPass.h"

namespace engine {

class SingleRadixSort {
public:
    using SORT_TYPE = uint32_t;

    explicit SingleRadixSort(uint32_t numElements = 1000000, uint32_t seed = 1);

    void execute(GPUContext *gpuContext);

    [[nodiscard]] double gpuSortTimeMs() const { return m_gpuSortTime; }
    [[nodiscard]] double cpuSortTimeMs() const { return m_cpuSortTime; }

private:
    GPUContext *m_gpuContext = nullptr;
    std::shared_ptr<SingleRadixSortPass> m_pass;

    const uint32_t NUM_ELEMENTS;
    const size_t NUM_ELEMENTS_BYTES;
    const uint32_t m_seed;
    static constexpr uint32_t INPUT_BUFFER_INDEX = 0;

    std::vector<std::shared_ptr<Buffer>> m_buffers = std::vector<std::shared_ptr<Buffer>>(2);
    std::vector<SORT_TYPE> m_elementsIn;
    double m_gpuSortTime = 0.0, m_cpuSortTime = 0.0;

    static inline const char *PRINT_PREFIX = "[SingleRadixSort] ";
};

}  // namespace engine
