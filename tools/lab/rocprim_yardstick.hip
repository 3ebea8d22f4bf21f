// Independent yardstick, NOT part of the product: rocPRIM's device radix sort (the vendor library shipped in /opt/rocm) on the
// bench workload, timed the way bench.py times ours (K pre-staged batches sorted back to back, keys resident), and our library's
// output compared with rocPRIM's bit for bit.  The product never links or calls rocPRIM ("cross-check /
// perf yardstick only").
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/lab/rocprim_yardstick.hip -Lvkradixsort_amd -lvkradixsort_amd \
//         -Wl,-rpath,$PWD/vkradixsort_amd -o tools/lab/rocprim_yardstick
//   tools/lab/rocprim_yardstick [N] [K] [pairs]
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "vkradixsort_amd.h"

#define CK(x)                                                                           \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            std::exit(1);                                                               \
        }                                                                               \
    } while (0)
#define VK(x)                                                                 \
    do {                                                                      \
        int r_ = (x);                                                         \
        if (r_ != 0) {                                                        \
            std::fprintf(stderr, "%s:%d vrs status %d\n", __FILE__, __LINE__, r_); \
            std::exit(1);                                                     \
        }                                                                     \
    } while (0)

int main(int argc, char **argv) {
    const size_t n = argc > 1 ? static_cast<size_t>(std::atof(argv[1])) : 100000000u;
    const int K = argc > 2 ? std::atoi(argv[2]) : 10;
    const bool pairs = argc > 3 && std::strcmp(argv[3], "pairs") == 0;
    std::vector<uint32_t> h(n), hv;
    std::mt19937 gen(1);  // the reference's generator (MultiRadixSort.cpp:121-133), full 32 bits
    for (auto &k : h) k = gen();
    if (pairs) {
        hv.resize(n);
        for (size_t i = 0; i < n; ++i) hv[i] = static_cast<uint32_t>(i);
    }
    uint32_t *src, *vsrc = nullptr, *out, *vout = nullptr;
    CK(hipMalloc(&src, n * 4));
    CK(hipMalloc(&out, n * 4));
    CK(hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice));
    if (pairs) {
        CK(hipMalloc(&vsrc, n * 4));
        CK(hipMalloc(&vout, n * 4));
        CK(hipMemcpy(vsrc, hv.data(), n * 4, hipMemcpyHostToDevice));
    }
    std::vector<uint32_t *> batch(K), vbatch(K, nullptr);
    for (int i = 0; i < K; ++i) {
        CK(hipMalloc(&batch[i], n * 4));
        if (pairs) CK(hipMalloc(&vbatch[i], n * 4));
    }
    const auto rearm = [&] {
        for (int i = 0; i < K; ++i) {
            CK(hipMemcpy(batch[i], src, n * 4, hipMemcpyDeviceToDevice));
            if (pairs) CK(hipMemcpy(vbatch[i], vsrc, n * 4, hipMemcpyDeviceToDevice));
        }
        CK(hipDeviceSynchronize());
    };
    // ---- rocPRIM: out-of-place radix_sort_keys / radix_sort_pairs over all 32 bits, temporary storage allocated once
    size_t tmp_bytes = 0;
    if (pairs)
        CK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, src, out, vsrc, vout, n, 0, 32, hipStreamDefault));
    else
        CK(rocprim::radix_sort_keys(nullptr, tmp_bytes, src, out, n, 0, 32, hipStreamDefault));
    void *tmp;
    CK(hipMalloc(&tmp, tmp_bytes));
    double best_rp = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        rearm();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < K; ++i) {
            if (pairs)
                CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, batch[i], out, vbatch[i], vout, n, 0, 32, hipStreamDefault));
            else
                CK(rocprim::radix_sort_keys(tmp, tmp_bytes, batch[i], out, n, 0, 32, hipStreamDefault));
        }
        CK(hipDeviceSynchronize());
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rep > 0) best_rp = std::min(best_rp, dt / K);
    }
    std::vector<uint32_t> ref(n), refv;
    CK(hipMemcpy(ref.data(), out, n * 4, hipMemcpyDeviceToHost));
    if (pairs) {
        refv.resize(n);
        CK(hipMemcpy(refv.data(), vout, n * 4, hipMemcpyDeviceToHost));
    }
    // ---- ours, through the C ABI, on the same batches
    vrs_context ctx;
    VK(vrs_context_create(0, &ctx));
    std::vector<vrs_buffer> kb(K), vb(K, nullptr);
    vrs_buffer ktmp, vtmp = nullptr;
    for (int i = 0; i < K; ++i) {
        VK(vrs_buffer_wrap(ctx, batch[i], n * 4, &kb[i]));
        if (pairs) VK(vrs_buffer_wrap(ctx, vbatch[i], n * 4, &vb[i]));
    }
    VK(vrs_buffer_wrap(ctx, out, n * 4, &ktmp));
    if (pairs) VK(vrs_buffer_wrap(ctx, vout, n * 4, &vtmp));
    double best_us = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        rearm();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < K; ++i) {
            if (pairs)
                VK(vrs_sort_pairs_u32(ctx, kb[i], ktmp, vb[i], vtmp, static_cast<uint32_t>(n)));
            else
                VK(vrs_sort_keys_u32(ctx, kb[i], ktmp, static_cast<uint32_t>(n)));
        }
        VK(vrs_queue_wait_idle(ctx));
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rep > 0) best_us = std::min(best_us, dt / K);
    }
    std::vector<uint32_t> mine(n), minev;
    CK(hipMemcpy(mine.data(), batch[K - 1], n * 4, hipMemcpyDeviceToHost));
    bool same = std::memcmp(mine.data(), ref.data(), n * 4) == 0;
    if (pairs) {
        minev.resize(n);
        CK(hipMemcpy(minev.data(), vbatch[K - 1], n * 4, hipMemcpyDeviceToHost));
        same = same && std::memcmp(minev.data(), refv.data(), n * 4) == 0;
    }
    std::printf("%s N=%zu K=%d: rocPRIM radix_sort_%s %.4f ms/sort (%.1f G/s, temp storage %.1f MB) | vkradixsort_amd vrs_sort_%s_u32 %.4f ms/sort "
                "(%.1f G/s) | ratio %.2fx | outputs identical: %s\n",
                pairs ? "pairs" : "keys", n, K, pairs ? "pairs" : "keys", best_rp * 1e3, n / best_rp / 1e9, tmp_bytes / 1e6,
                pairs ? "pairs" : "keys", best_us * 1e3, n / best_us / 1e9, best_rp / best_us, same ? "yes" : "NO");
    return same ? 0 : 2;
}
