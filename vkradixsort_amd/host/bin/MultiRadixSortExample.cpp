// multiradixsortexample [NUM_ELEMENTS] [NUM_BLOCKS_PER_WORKGROUP] [seed] [timed repetitions] [28bit|full] [64bit|32bit] [onecall|stages] [pairs]
// Counterpart of the reference's multiradixsort/src/bin/MultiRadixSortExample.cpp: context -> execute ->
// shutdown, std::exception -> EXIT_FAILURE.  Without arguments it sorts the reference's 1 000 000 keys.
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>

#include "MultiRadixSort.h"
#include "engine/core/GPUContext.h"

int main(int argc, char **argv) {
    const uint32_t numElements = argc > 1 ? static_cast<uint32_t>(std::strtod(argv[1], nullptr)) : 1000000u;
    const uint32_t blocks = argc > 2 ? static_cast<uint32_t>(std::atoi(argv[2])) : 32u;
    const uint32_t seed = argc > 3 ? static_cast<uint32_t>(std::atoi(argv[3])) : 1u;
    const uint32_t reps = argc > 4 ? static_cast<uint32_t>(std::atoi(argv[4])) : 1u;
    const bool keys28 = argc > 5 && std::strcmp(argv[5], "28bit") == 0;
    const bool sort64 = argc > 6 && std::strcmp(argv[6], "64bit") == 0;  // the reference's SORT_64_BIT
    const bool oneCall = argc > 7 && std::strcmp(argv[7], "onecall") == 0;  // let the library run the passes itself
    const bool pairs = argc > 8 && std::strcmp(argv[8], "pairs") == 0;  // BASELINE.json configs[3]: key + uint32 payload

    engine::GPUContext gpu(engine::Queues::QueueFamilies::COMPUTE_FAMILY | engine::Queues::TRANSFER_FAMILY);
    try {
        gpu.init();
        if (sort64) {
            auto app = std::make_shared<engine::MultiRadixSort64>(numElements, blocks, seed, keys28, reps);
            app->m_oneCallSort = oneCall;
            app->m_sortPairs = pairs;
            app->execute(&gpu);
        } else {
            auto app = std::make_shared<engine::MultiRadixSort>(numElements, blocks, seed, keys28, reps);
            app->m_oneCallSort = oneCall;
            app->m_sortPairs = pairs;
            app->execute(&gpu);
        }
        gpu.shutdown();
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
