// singleradixsortexample [NUM_ELEMENTS] [seed]
// Counterpart of the reference's singleradixsort/src/bin/SingleRadixSortExample.cpp.
#include <cstdlib>
#include <iostream>
#include <memory>

#include "SingleRadixSort.h"
#include "engine/core/GPUContext.h"

int main(int argc, char **argv) {
    const uint32_t numElements = argc > 1 ? static_cast<uint32_t>(std::strtod(argv[1], nullptr)) : 1000000u;
    const uint32_t seed = argc > 2 ? static_cast<uint32_t>(std::atoi(argv[2])) : 1u;

    engine::GPUContext gpu(engine::Queues::QueueFamilies::COMPUTE_FAMILY | engine::Queues::TRANSFER_FAMILY);
    try {
        gpu.init();
        auto app = std::make_shared<engine::SingleRadixSort>(numElements, seed);
        app->execute(&gpu);
        gpu.shutdown();
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
