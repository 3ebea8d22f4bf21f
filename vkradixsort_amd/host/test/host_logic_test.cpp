// CPU-only checks of the C++ host mirror's logic (no device calls): the pieces of MultiRadixSort.cpp /
// ComputePass.h that are pure host arithmetic.  Exit code 0 == all passed; prints the first failure.
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "MultiRadixSort.h"

#define CHECK(cond)                                                             \
    do {                                                                        \
        if (!(cond)) {                                                          \
            std::cerr << "FAILED: " #cond " (line " << __LINE__ << ")" << std::endl; \
            return EXIT_FAILURE;                                                \
        }                                                                       \
    } while (0)

int main() {
    using engine::ComputePass;
    using engine::Extent3D;
    using engine::MultiRadixSort;
    using engine::MultiRadixSort64;

    // launch shape: W = ceil(ceil(N/B)/256), the (N, B) -> W pairs quoted by the reference (README.md:257-261)
    const struct { uint32_t n, b, w; } shapes[] = {{1000000, 32, 123}, {1000000, 1, 3907}, {1000000, 4096, 1}, {10000000, 32, 1221},
                                                   {10000000, 512, 77}, {100000000, 32, 12208}, {100000000, 4096, 96}, {1000, 32, 1}};
    for (const auto &s : shapes) {
        const uint32_t gis = s.n / s.b + (s.n % s.b ? 1u : 0u);
        const Extent3D d = ComputePass::getDispatchSize(gis, 1, 1, Extent3D{256, 1, 1});
        CHECK(d.width == s.w && d.height == 1 && d.depth == 1);
        CHECK(vrs_workgroup_count(s.n, s.b) == s.w);
    }

    // keys: std::mt19937(seed)() raw outputs; seed 12345 starts 3992670690, 3823185381;
    // the reference's own range [0, 0x0FFFFFFF] is raw >> 4
    std::vector<uint32_t> k;
    MultiRadixSort::generateRandomNumbers(k, 4, 12345, false);
    CHECK(k.size() == 4 && k[0] == 3992670690u && k[1] == 3823185381u && k[2] == 1358822685u && k[3] == 561383553u);
    MultiRadixSort::generateRandomNumbers(k, 2, 12345, true);
    CHECK(k[0] == 249541918u && k[1] == 238949086u);
    std::vector<uint64_t> k64;
    MultiRadixSort64::generateRandomNumbers(k64, 1000, 7, true);
    for (auto v : k64) CHECK(v <= 0x0FFFFFFFFFFFull);  // the reference's 64-bit key range (MultiRadixSort.cpp:128)

    // sort(): in-place std::sort, returns milliseconds; testSort(): passes on equality, throws "TEST FAILED." otherwise
    MultiRadixSort::generateRandomNumbers(k, 100000, 1, false);
    std::vector<uint32_t> sorted = k;
    const double ms = MultiRadixSort::sort(sorted);
    CHECK(ms >= 0.0 && std::is_sorted(sorted.begin(), sorted.end()));
    std::vector<uint32_t> same = sorted;
    CHECK(MultiRadixSort::testSort(sorted, same));
    same[777] ^= 1u;
    bool threw = false;
    try {
        MultiRadixSort::testSort(sorted, same);
    } catch (const std::runtime_error &e) {
        threw = std::string(e.what()) == "TEST FAILED.";
    }
    CHECK(threw);
    same.pop_back();
    threw = false;
    try {
        MultiRadixSort::testSort(sorted, same);
    } catch (const std::runtime_error &e) {
        threw = std::string(e.what()) == "TEST FAILED.";
    }
    CHECK(threw);

    // sortPairs(): the CPU reference of the pairs extension -- std::stable_sort by key, payloads follow their keys
    {
        std::vector<uint32_t> pk = {5u, 1u, 5u, 0u, 1u, 5u};
        std::vector<uint32_t> pv = {0u, 1u, 2u, 3u, 4u, 5u};
        const double pms = MultiRadixSort::sortPairs(pk, pv);
        CHECK(pms >= 0.0);
        CHECK((pk == std::vector<uint32_t>{0u, 1u, 1u, 5u, 5u, 5u}));
        CHECK((pv == std::vector<uint32_t>{3u, 1u, 4u, 0u, 2u, 5u}));  // ties keep their input order
        std::vector<uint64_t> pk64 = {9ull << 40, 3ull, 9ull << 40};
        std::vector<uint32_t> pv64 = {7u, 8u, 9u};
        MultiRadixSort64::sortPairs(pk64, pv64);
        CHECK(pk64[0] == 3ull && pv64[0] == 8u && pv64[1] == 7u && pv64[2] == 9u);
    }

    // push-constant blocks stay the reference's 16-byte layout
    static_assert(sizeof(engine::MultiRadixSortPass::PushConstants) == 16, "PushConstants");
    static_assert(sizeof(engine::MultiRadixSortPass::PushConstantsHistograms) == 16, "PushConstantsHistograms");
    static_assert(engine::MultiRadixSortPass::RADIX_SORT_HISTOGRAMS == 0 && engine::MultiRadixSortPass::RADIX_SORT == 1, "stage ids");

    // a context that was never initialised refuses work with std::runtime_error (no device needed for that)
    engine::GPUContext gpu;
    threw = false;
    try {
        gpu.waitIdle();
    } catch (const std::runtime_error &) {
        threw = true;
    }
    CHECK(threw);
    CHECK(gpu.getMultiBufferedCount() == 2 && gpu.getActiveIndex() == 0);
    gpu.incrementActiveIndex();
    CHECK(gpu.getActiveIndex() == 1);
    gpu.incrementActiveIndex();
    CHECK(gpu.getActiveIndex() == 0);

    std::cout << "host logic ok" << std::endl;
    return EXIT_SUCCESS;
}
