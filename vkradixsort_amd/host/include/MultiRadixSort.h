// engine::MultiRadixSort -- program logic of the multi_radixsort example
// (reference: multiradixsort/include/MultiRadixSort.h:9-54, src/MultiRadixSort.cpp:5-161).
// NUM_ELEMENTS stays THE entry-point parameter but is a constructor argument (default 1 000 000, the
// reference's compile-time value, MultiRadixSort.h:29); NUM_BLOCKS_PER_WORKGROUP likewise (default 32,
// MultiRadixSort.cpp:12).  Keys are generated from std::mt19937(seed)() -- reproducible, full 32 bit.
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "MultiRadixSortPass.h"

namespace engine {

// SORT_TYPE_T = uint32_t is the reference's SORT_32BIT (four passes), uint64_t its SORT_64_BIT (eight passes,
// MultiRadixSort.h:10-18 / MultiRadixSort.cpp:51-55, 44-bit keys from its generator, :128).
template <typename SORT_TYPE_T>
class BasicMultiRadixSort {
public:
    using SORT_TYPE = SORT_TYPE_T;

    explicit BasicMultiRadixSort(uint32_t numElements = 1000000, uint32_t numBlocksPerWorkgroup = 32, uint32_t seed = 1,
                            bool reference28BitKeys = false, uint32_t timedRepetitions = 1);

    static inline const char *PRINT_PREFIX = "[MultiRadixSort] ";
    static constexpr uint32_t NUM_ITERATIONS = sizeof(SORT_TYPE);  // one pass per key byte: 4 or 8

    void execute(GPUContext *gpuContext);

    // false (default): execute() drives the two stages pass by pass, exactly like the reference's loop
    // (MultiRadixSort.cpp:50-61).  true: the library runs the passes itself (vrs_sort_keys_u32 / _u64) -- for 32-bit
    // keys from 2^13 elements on that is ONE counting read plus four look-back scatter passes, 36 instead of
    // 48 bytes per key.  Same buffers, same result in buffer 0.
    bool m_oneCallSort = false;

    // Build extension (BASELINE.json configs[3]): every key carries a uint32 payload (its input index).  execute()
    // binds the payload ping-pong pair at (RADIX_SORT, 3) / (RADIX_SORT, 4) beside the keys' (RADIX_SORT, 0) / (1),
    // runs the same passes (MultiRadixSortPass::m_sortPairs) and verifies keys AND payloads against
    // std::stable_sort by key -- the sort is stable, so the payloads are the stable permutation.
    bool m_sortPairs = false;

    // results of the last execute() for programmatic callers / the sweep harness
    [[nodiscard]] double gpuSortTimeMs() const { return m_gpuSortTime; }
    [[nodiscard]] double cpuSortTimeMs() const { return m_cpuSortTime; }

    // host helpers, public so tests can exercise them without a device
    static void generateRandomNumbers(std::vector<SORT_TYPE> &buffer, uint32_t numElements, uint32_t seed,
                                      bool reference28BitKeys);
    static double sort(std::vector<SORT_TYPE> &buffer);
    // CPU reference of the pairs extension: std::stable_sort by key; values follow their keys
    static double sortPairs(std::vector<SORT_TYPE> &keys, std::vector<uint32_t> &values);
    static bool testSort(std::vector<SORT_TYPE> &reference, std::vector<SORT_TYPE> &outBuffer,
                         const char *printPrefix = PRINT_PREFIX);

private:
    GPUContext *m_gpuContext = nullptr;
    std::shared_ptr<MultiRadixSortPass> m_pass;

    const uint32_t RADIX_SORT_BINS = 256;
    const uint32_t NUM_ELEMENTS;
    const uint32_t NUM_BLOCKS_PER_WORKGROUP;
    const size_t NUM_ELEMENTS_BYTES;
    const uint32_t m_seed;
    const bool m_reference28BitKeys;
    const uint32_t m_timedRepetitions;

    std::vector<std::shared_ptr<Buffer>> m_buffers = std::vector<std::shared_ptr<Buffer>>(3);
    std::vector<std::shared_ptr<Buffer>> m_valueBuffers = std::vector<std::shared_ptr<Buffer>>(2);  // pairs only
    std::vector<SORT_TYPE> m_elementsIn;
    std::vector<uint32_t> m_valuesIn;  // pairs only: 0, 1, ..., N-1
    double m_gpuSortTime = 0.0, m_cpuSortTime = 0.0;

    void prepareBuffers();
    void verify(std::vector<SORT_TYPE> &reference);
    void releaseBuffers();
};

using MultiRadixSort = BasicMultiRadixSort<uint32_t>;
using MultiRadixSort64 = BasicMultiRadixSort<uint64_t>;

}  // namespace engine
