#include "MultiRadixSort.h"

#include <algorithm>
#include <cassert>
#include <chrono>
#include <iostream>
#include <numeric>
#include <random>
#include <stdexcept>

namespace engine {

namespace {
double elapsedMs(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return static_cast<double>(std::chrono::duration_cast<std::chrono::microseconds>(b - a).count()) * 1e-3;
}
}  // namespace

template <typename T>
BasicMultiRadixSort<T>::BasicMultiRadixSort(uint32_t numElements, uint32_t numBlocksPerWorkgroup, uint32_t seed,
                               bool reference28BitKeys, uint32_t timedRepetitions)
    : NUM_ELEMENTS(numElements),
      NUM_BLOCKS_PER_WORKGROUP(numBlocksPerWorkgroup),
      NUM_ELEMENTS_BYTES(static_cast<size_t>(numElements) * sizeof(T)),
      m_seed(seed),
      m_reference28BitKeys(reference28BitKeys),
      m_timedRepetitions(timedRepetitions ? timedRepetitions : 1) {
    if (numBlocksPerWorkgroup == 0) throw std::runtime_error("NUM_BLOCKS_PER_WORKGROUP must be >= 1");
}

template <typename T>
void BasicMultiRadixSort<T>::execute(GPUContext *gpuContext) {
    m_gpuContext = gpuContext;

    // launch shape: one contract workgroup covers NUM_BLOCKS_PER_WORKGROUP blocks of 256 keys
    m_pass = std::make_shared<MultiRadixSortPass>(gpuContext);
    m_pass->create();
    m_pass->m_sort64Bit = sizeof(T) == 8;
    const uint32_t globalInvocationSize =
        NUM_ELEMENTS / NUM_BLOCKS_PER_WORKGROUP + (NUM_ELEMENTS % NUM_BLOCKS_PER_WORKGROUP ? 1u : 0u);
    m_pass->setGlobalInvocationSize(MultiRadixSortPass::RADIX_SORT_HISTOGRAMS, globalInvocationSize, 1, 1);
    m_pass->setGlobalInvocationSize(MultiRadixSortPass::RADIX_SORT, globalInvocationSize, 1, 1);

    const uint32_t NUM_WORKGROUPS = m_pass->getWorkGroupCount(MultiRadixSortPass::RADIX_SORT_HISTOGRAMS).width;
    assert(NUM_WORKGROUPS == m_pass->getWorkGroupCount(MultiRadixSortPass::RADIX_SORT).width);
    m_pass->m_pushConstantsHistogram = {NUM_ELEMENTS, 0, NUM_WORKGROUPS, NUM_BLOCKS_PER_WORKGROUP};
    m_pass->m_pushConstants = {NUM_ELEMENTS, 0, NUM_WORKGROUPS, NUM_BLOCKS_PER_WORKGROUP};

    prepareBuffers();
    std::cout << PRINT_PREFIX << "Sorting " << NUM_ELEMENTS << " " << (sizeof(m_elementsIn[0]) * 8) << "bit numbers."
              << std::endl;

    // ping-pong: the live descriptor copy alternates with GPUContext::incrementActiveIndex()
    const uint32_t even = m_gpuContext->getActiveIndex();
    const uint32_t odd = (even + 1) % 2;
    constexpr auto H = MultiRadixSortPass::RADIX_SORT_HISTOGRAMS;
    constexpr auto R = MultiRadixSortPass::RADIX_SORT;
    m_pass->setStorageBuffer(even, H, 0, m_buffers[0].get());  // passes 0, 2 read buffer0 ...
    m_pass->setStorageBuffer(even, R, 0, m_buffers[0].get());
    m_pass->setStorageBuffer(even, R, 1, m_buffers[1].get());  // ... and write buffer1
    m_pass->setStorageBuffer(odd, H, 0, m_buffers[1].get());   // passes 1, 3 read buffer1 ...
    m_pass->setStorageBuffer(odd, R, 0, m_buffers[1].get());
    m_pass->setStorageBuffer(odd, R, 1, m_buffers[0].get());   // ... and write buffer0
    m_pass->setStorageBuffer(H, 1, m_buffers[2].get());
    m_pass->setStorageBuffer(R, 2, m_buffers[2].get());
    if (m_sortPairs) {  // payloads ping-pong with their keys
        m_pass->m_sortPairs = true;
        m_pass->setStorageBuffer(even, R, 3, m_valueBuffers[0].get());
        m_pass->setStorageBuffer(even, R, 4, m_valueBuffers[1].get());
        m_pass->setStorageBuffer(odd, R, 3, m_valueBuffers[1].get());
        m_pass->setStorageBuffer(odd, R, 4, m_valueBuffers[0].get());
    }

    // timed region, as in the reference: first pass enqueue -> queue idle; data already resident
    std::shared_ptr<Buffer> pristine, pristineValues;
    if (m_timedRepetitions > 1) {
        pristine = Buffer::fillDeviceWithStagingBuffer(m_gpuContext, {.m_sizeBytes = NUM_ELEMENTS_BYTES}, m_elementsIn.data());
        if (m_sortPairs)
            pristineValues = Buffer::fillDeviceWithStagingBuffer(
                m_gpuContext, {.m_sizeBytes = static_cast<size_t>(NUM_ELEMENTS) * sizeof(uint32_t)}, m_valuesIn.data());
    }
    double best = 0.0;
    for (uint32_t rep = 0; rep < m_timedRepetitions; rep++) {
        if (rep > 0) {
            m_buffers[0]->copyFrom(*pristine);
            if (m_sortPairs) m_valueBuffers[0]->copyFrom(*pristineValues);
            m_gpuContext->waitIdle();
        }
        const auto begin = std::chrono::steady_clock::now();
        if (m_oneCallSort && m_sortPairs) {
            const auto sortPairsFn = sizeof(T) == 8 ? vrs_sort_pairs_u64 : vrs_sort_pairs_u32;
            m_gpuContext->check(sortPairsFn(m_gpuContext->handle(), m_buffers[0]->getBuffer(), m_buffers[1]->getBuffer(),
                                            m_valueBuffers[0]->getBuffer(), m_valueBuffers[1]->getBuffer(), NUM_ELEMENTS),
                                "Failed to enqueue the one-call sort");
        } else if (m_oneCallSort) {
            const auto sortKeys = sizeof(T) == 8 ? vrs_sort_keys_u64 : vrs_sort_keys_u32;
            m_gpuContext->check(sortKeys(m_gpuContext->handle(), m_buffers[0]->getBuffer(), m_buffers[1]->getBuffer(), NUM_ELEMENTS),
                                "Failed to enqueue the one-call sort");
        } else {
            Semaphore awaitBeforeExecution = NULL_SEMAPHORE;
            for (uint32_t i = 0; i < NUM_ITERATIONS; i++) {
                m_pass->m_pushConstantsHistogram.g_shift = 8 * i;
                m_pass->m_pushConstants.g_shift = 8 * i;
                awaitBeforeExecution = m_pass->execute(awaitBeforeExecution);
                m_gpuContext->incrementActiveIndex();
            }
        }
        m_gpuContext->waitIdle();
        const double ms = elapsedMs(begin, std::chrono::steady_clock::now());
        best = (rep == 0 || ms < best) ? ms : best;
    }
    m_gpuSortTime = best;
    std::cout << PRINT_PREFIX << "GPU sort finished in " << m_gpuSortTime << "[ms]." << std::endl;

    m_cpuSortTime = m_sortPairs ? sortPairs(m_elementsIn, m_valuesIn) : sort(m_elementsIn);
    std::cout << PRINT_PREFIX << "CPU sort finished in " << m_cpuSortTime << "[ms]." << std::endl;

    verify(m_elementsIn);

    releaseBuffers();
    m_pass->release();
}

template <typename T>
void BasicMultiRadixSort<T>::prepareBuffers() {
    generateRandomNumbers(m_elementsIn, NUM_ELEMENTS, m_seed, m_reference28BitKeys);
    m_buffers[0] = Buffer::fillDeviceWithStagingBuffer(
        m_gpuContext, {.m_sizeBytes = NUM_ELEMENTS_BYTES, .m_name = "radixSort.elementBuffer0"}, m_elementsIn.data());
    // buffer1 and the histogram table are fully overwritten before they are read: no zero upload
    m_buffers[1] = std::make_shared<Buffer>(
        m_gpuContext, Buffer::BufferSettings{.m_sizeBytes = NUM_ELEMENTS_BYTES, .m_name = "radixSort.elementBuffer1"});
    const size_t histogramBytes = static_cast<size_t>(m_pass->getWorkGroupCount(MultiRadixSortPass::RADIX_SORT_HISTOGRAMS).width) *
                                  RADIX_SORT_BINS * sizeof(uint32_t);
    m_buffers[2] = std::make_shared<Buffer>(
        m_gpuContext, Buffer::BufferSettings{.m_sizeBytes = histogramBytes, .m_name = "radixSort.histogramsBuffer"});
    if (m_sortPairs) {
        m_valuesIn.resize(NUM_ELEMENTS);
        std::iota(m_valuesIn.begin(), m_valuesIn.end(), 0u);
        const size_t valueBytes = static_cast<size_t>(NUM_ELEMENTS) * sizeof(uint32_t);
        m_valueBuffers[0] = Buffer::fillDeviceWithStagingBuffer(
            m_gpuContext, {.m_sizeBytes = valueBytes, .m_name = "radixSort.valueBuffer0"}, m_valuesIn.data());
        m_valueBuffers[1] = std::make_shared<Buffer>(
            m_gpuContext, Buffer::BufferSettings{.m_sizeBytes = valueBytes, .m_name = "radixSort.valueBuffer1"});
    }
}

template <typename T>
void BasicMultiRadixSort<T>::verify(std::vector<T> &reference) {
    std::vector<T> data(NUM_ELEMENTS);
    m_buffers[0]->downloadWithStagingBuffer(data.data());  // an even number of passes: the result is back in buffer0
    testSort(reference, data);
    if (m_sortPairs) {  // payloads: element for element equal to std::stable_sort's
        std::vector<uint32_t> values(NUM_ELEMENTS);
        m_valueBuffers[0]->downloadWithStagingBuffer(values.data());
        const auto mismatch = std::mismatch(m_valuesIn.begin(), m_valuesIn.end(), values.begin());
        if (mismatch.first != m_valuesIn.end()) {
            const auto i = mismatch.first - m_valuesIn.begin();
            std::cerr << PRINT_PREFIX << *mismatch.first << " = referenceValues[" << i << "] != outValues[" << i
                      << "] = " << *mismatch.second << std::endl;
            throw std::runtime_error("TEST FAILED.");
        }
        std::cout << PRINT_PREFIX << "Payloads follow their keys (stable)." << std::endl;
    }
}

template <typename T>
void BasicMultiRadixSort<T>::releaseBuffers() {
    for (const auto &buffer : m_buffers)
        if (buffer) buffer->release();
    for (auto &buffer : m_valueBuffers)
        if (buffer) {
            buffer->release();
            buffer.reset();
        }
}

template <typename T>
void BasicMultiRadixSort<T>::generateRandomNumbers(std::vector<T> &buffer, uint32_t numElements, uint32_t seed,
                                           bool reference28BitKeys) {
    std::mt19937 gen(seed);
    buffer.resize(numElements);
    // reference-faithful ranges: [0, 0x0FFFFFFF] (32 bit) / [0, 0x0FFFFFFFFFFF] (64 bit), i.e. the top 4 / 20 bits clear
    const uint32_t drop = reference28BitKeys ? (sizeof(T) == 8 ? 20u : 4u) : 0u;
    const auto draw = [](std::mt19937 &g) -> T {
        if constexpr (sizeof(T) == 8) {
            const uint64_t hi = g();
            return static_cast<T>((hi << 32) | g());
        } else {
            return static_cast<T>(g());
        }
    };
    for (auto &key : buffer) key = draw(gen) >> drop;
}

template <typename T>
double BasicMultiRadixSort<T>::sort(std::vector<T> &buffer) {
    const auto begin = std::chrono::steady_clock::now();
    std::sort(buffer.begin(), buffer.end());
    return elapsedMs(begin, std::chrono::steady_clock::now());
}

template <typename T>
double BasicMultiRadixSort<T>::sortPairs(std::vector<T> &keys, std::vector<uint32_t> &values) {
    const auto begin = std::chrono::steady_clock::now();
    std::vector<uint32_t> order(keys.size());
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
    std::vector<T> sortedKeys(keys.size());
    std::vector<uint32_t> sortedValues(values.size());
    for (size_t i = 0; i < order.size(); i++) {
        sortedKeys[i] = keys[order[i]];
        sortedValues[i] = values[order[i]];
    }
    keys.swap(sortedKeys);
    values.swap(sortedValues);
    return elapsedMs(begin, std::chrono::steady_clock::now());
}

template <typename T>
bool BasicMultiRadixSort<T>::testSort(std::vector<T> &reference, std::vector<T> &outBuffer,
                              const char *printPrefix) {
    if (reference.size() != outBuffer.size()) {
        std::cerr << printPrefix << "reference.size() != outBuffer.size()" << std::endl;
        throw std::runtime_error("TEST FAILED.");
    }
    const auto mismatch = std::mismatch(reference.begin(), reference.end(), outBuffer.begin());
    if (mismatch.first != reference.end()) {
        const auto i = mismatch.first - reference.begin();
        std::cerr << printPrefix << *mismatch.first << " = reference[" << i << "] != outBuffer[" << i
                  << "] = " << *mismatch.second << std::endl;
        throw std::runtime_error("TEST FAILED.");
    }
    std::cout << printPrefix << "Test passed." << std::endl;
    return true;
}

template class BasicMultiRadixSort<uint32_t>;
template class BasicMultiRadixSort<uint64_t>;

}  // namespace engine
