#include "SingleRadixSort.h"

#include <chrono>
#include <iostream>

#include "MultiRadixSort.h"  // shares the host helpers (generator, std::sort timing, testSort)

namespace engine {

void SingleRadixSortPass::recordCommands() {
    // set 0: b0 = buffer0 (input and final result), b1 = buffer1
    m_gpuContext->check(vrs_single_radixsort(m_gpuContext->handle(), boundBuffer(RADIX_SORT, 0)->getBuffer(),
                                             boundBuffer(RADIX_SORT, 1)->getBuffer(), m_pushConstants.g_num_elements),
                        "Failed to submit compute command buffer!");
}

SingleRadixSort::SingleRadixSort(uint32_t numElements, uint32_t seed)
    : NUM_ELEMENTS(numElements), NUM_ELEMENTS_BYTES(static_cast<size_t>(numElements) * sizeof(SORT_TYPE)), m_seed(seed) {}

void SingleRadixSort::execute(GPUContext *gpuContext) {
    m_gpuContext = gpuContext;
    m_pass = std::make_shared<SingleRadixSortPass>(gpuContext);
    m_pass->create();
    m_pass->setGlobalInvocationSize(SingleRadixSortPass::RADIX_SORT, 256, 1, 1);  // exactly one workgroup
    m_pass->m_pushConstants.g_num_elements = NUM_ELEMENTS;

    MultiRadixSort::generateRandomNumbers(m_elementsIn, NUM_ELEMENTS, m_seed, false);
    m_buffers[INPUT_BUFFER_INDEX] = Buffer::fillDeviceWithStagingBuffer(
        m_gpuContext, {.m_sizeBytes = NUM_ELEMENTS_BYTES, .m_name = "radixSort.elementBuffer0"}, m_elementsIn.data());
    m_buffers[1 - INPUT_BUFFER_INDEX] = std::make_shared<Buffer>(
        m_gpuContext, Buffer::BufferSettings{.m_sizeBytes = NUM_ELEMENTS_BYTES, .m_name = "radixSort.elementBuffer1"});
    std::cout << PRINT_PREFIX << "Sorting " << NUM_ELEMENTS << " " << (sizeof(m_elementsIn[0]) * 8) << "bit numbers."
              << std::endl;

    m_pass->setStorageBuffer(SingleRadixSortPass::RADIX_SORT, 0, m_buffers[INPUT_BUFFER_INDEX].get());
    m_pass->setStorageBuffer(SingleRadixSortPass::RADIX_SORT, 1, m_buffers[1 - INPUT_BUFFER_INDEX].get());

    const auto begin = std::chrono::steady_clock::now();
    m_pass->execute(NULL_SEMAPHORE);
    m_gpuContext->waitIdle();
    m_gpuSortTime = static_cast<double>(std::chrono::duration_cast<std::chrono::microseconds>(
                                            std::chrono::steady_clock::now() - begin)
                                            .count()) *
                    1e-3;
    std::cout << PRINT_PREFIX << "GPU sort finished in " << m_gpuSortTime << "[ms]." << std::endl;

    m_cpuSortTime = MultiRadixSort::sort(m_elementsIn);
    std::cout << PRINT_PREFIX << "CPU sort finished in " << m_cpuSortTime << "[ms]." << std::endl;

    std::vector<SORT_TYPE> data(NUM_ELEMENTS);
    m_buffers[INPUT_BUFFER_INDEX]->downloadWithStagingBuffer(data.data());
    MultiRadixSort::testSort(m_elementsIn, data, PRINT_PREFIX);

    for (const auto &buffer : m_buffers) buffer->release();
    m_pass->release();
}

}  // namespace engine
