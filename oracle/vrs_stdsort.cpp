// vrs_stdsort.cpp -- TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// The reference's own ground truth and CPU baseline, restated: single-threaded in-place
// std::sort over a std::vector<uint32_t>, timed with steady_clock and reported in ms
// (multiradixsort/src/MultiRadixSort.cpp:141-146), and the element-wise comparison that
// decides "Test passed." / "TEST FAILED." (:148-161).  The reference's file cannot be
// compiled here (its header chain pulls <vulkan/vulkan_core.h>), so
// "kind" in bench.py's cpu_baseline is "port", not "reference".
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <thread>
#include <vector>
#ifdef _OPENMP
#include <parallel/algorithm>
#endif

extern "C" {

// MultiRadixSort::sort, MultiRadixSort.cpp:141-146 -- in place, returns milliseconds.
double vrs_stdsort_u32(uint32_t *data, uint64_t n) {
    std::vector<uint32_t> buffer(data, data + n);  // the reference sorts a std::vector<SORT_TYPE>
    auto begin = std::chrono::steady_clock::now();
    std::sort(buffer.begin(), buffer.end());
    auto end = std::chrono::steady_clock::now();
    std::memcpy(data, buffer.data(), n * sizeof(uint32_t));
    return static_cast<double>(std::chrono::duration_cast<std::chrono::microseconds>(end - begin).count()) *
           std::pow(10, -3);
}

// MultiRadixSort::testSort, MultiRadixSort.cpp:148-161.
// returns -1 when every element matches ("Test passed."), -2 on size mismatch (:149-151),
// otherwise the first differing index (:153-157).
int64_t vrs_test_sort(const uint32_t *reference, uint64_t n_reference, const uint32_t *out_buffer,
                      uint64_t n_out) {
    if (n_reference != n_out) return -2;
    for (uint64_t i = 0; i < n_reference; i++) {
        if (reference[i] != out_buffer[i]) return static_cast<int64_t>(i);
    }
    return -1;
}

// Key+payload ground truth (build extension; the reference has no payload path): the
// reference's scatter is stable (multi_radixsort.comp:111-122), so pairs must equal
// std::stable_sort by key.
double vrs_stable_sort_pairs_u32(uint32_t *keys, uint32_t *values, uint64_t n) {
    std::vector<uint64_t> packed(n);
    for (uint64_t i = 0; i < n; i++) packed[i] = (static_cast<uint64_t>(keys[i]) << 32) | values[i];
    auto begin = std::chrono::steady_clock::now();
    std::stable_sort(packed.begin(), packed.end(),
                     [](uint64_t a, uint64_t b) { return (a >> 32) < (b >> 32); });
    auto end = std::chrono::steady_clock::now();
    for (uint64_t i = 0; i < n; i++) {
        keys[i] = static_cast<uint32_t>(packed[i] >> 32);
        values[i] = static_cast<uint32_t>(packed[i]);
    }
    return static_cast<double>(std::chrono::duration_cast<std::chrono::microseconds>(end - begin).count()) *
           std::pow(10, -3);
}

// Optional all-core line for the report (clearly labelled, never the headline baseline).
double vrs_parallel_sort_u32(uint32_t *data, uint64_t n) {
    auto begin = std::chrono::steady_clock::now();
#ifdef _OPENMP
    __gnu_parallel::sort(data, data + n);
#else
    std::sort(data, data + n);
#endif
    auto end = std::chrono::steady_clock::now();
    return static_cast<double>(std::chrono::duration_cast<std::chrono::microseconds>(end - begin).count()) *
           std::pow(10, -3);
}

uint32_t vrs_hardware_concurrency() { return std::thread::hardware_concurrency(); }

// first "model name" line of /proc/cpuinfo, for the "1 thread of M (<model>)" report line
int vrs_cpu_model(char *out, uint32_t cap) {
    if (cap == 0) return -1;
    out[0] = 0;
    FILE *f = std::fopen("/proc/cpuinfo", "r");
    if (!f) return -1;
    char line[512];
    while (std::fgets(line, sizeof line, f)) {
        if (std::strncmp(line, "model name", 10) == 0) {
            const char *c = std::strchr(line, ':');
            if (c) {
                c++;
                while (*c == ' ') c++;
                std::snprintf(out, cap, "%s", c);
                size_t len = std::strlen(out);
                if (len && out[len - 1] == '\n') out[len - 1] = 0;
            }
            break;
        }
    }
    std::fclose(f);
    return 0;
}

}  // extern "C"

extern "C" {
// 64-bit flavour of the same verification path (SORT_TYPE uint64_t)
double vrs_stdsort_u64(uint64_t *data, uint64_t n) {
    std::vector<uint64_t> buffer(data, data + n);
    auto begin = std::chrono::steady_clock::now();
    std::sort(buffer.begin(), buffer.end());
    auto end = std::chrono::steady_clock::now();
    std::memcpy(data, buffer.data(), n * sizeof(uint64_t));
    return static_cast<double>(std::chrono::duration_cast<std::chrono::microseconds>(end - begin).count()) *
           std::pow(10, -3);
}
}
