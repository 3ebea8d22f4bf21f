// Lab: why is the histogram kernel slower right after the scatter (83 us) than standalone (71 us)?
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../tools/lab/vrs_all_kernels.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main() {
    const uint32_t n = 100000000u, B = 32, W = 12208;
    std::vector<uint32_t> h(n);
    std::mt19937 gen(1);
    for (auto &x : h) x = gen();
    uint32_t *d_a, *d_b, *d_c, *d_hist;
    CK(hipMalloc(&d_a, (size_t)n * 4)); CK(hipMalloc(&d_b, (size_t)n * 4)); CK(hipMalloc(&d_c, (size_t)n * 4));
    CK(hipMemcpy(d_a, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_c, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_hist, (size_t)W * 1024));
    vrs::PrefixScratch sc;
    CK(hipMalloc(&sc.offsets, (size_t)W * 1024)); CK(hipMalloc(&sc.chunk_sums, (size_t)W * 1024));
    vrs::ScatterLaunch cfg; cfg.atomic_rank = true;
    hipEvent_t ev[8];
    for (auto &x : ev) CK(hipEventCreate(&x));
    CK(vrs::launch_histograms(0, d_a, d_hist, n, 0, W, B));
    CK(vrs::launch_prefix(0, d_hist, sc, W));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 4; ++rep) {
        // scatter a -> b, then: hist(b), hist(b) again, hist(c) [an untouched buffer], hist(b)
        CK(vrs::launch_scatter(0, d_a, d_b, nullptr, nullptr, sc.offsets, n, 0, W, B, true, cfg, {ev[0], ev[1]}));
        CK(vrs::launch_histograms(0, d_b, d_hist, n, 8, W, B, {ev[2], ev[3]}));
        CK(vrs::launch_histograms(0, d_b, d_hist, n, 8, W, B, {ev[4], ev[5]}));
        CK(vrs::launch_histograms(0, d_c, d_hist, n, 8, W, B, {ev[6], ev[7]}));
        CK(hipDeviceSynchronize());
        float s, h1, h2, h3;
        CK(hipEventElapsedTime(&s, ev[0], ev[1])); CK(hipEventElapsedTime(&h1, ev[2], ev[3]));
        CK(hipEventElapsedTime(&h2, ev[4], ev[5])); CK(hipEventElapsedTime(&h3, ev[6], ev[7]));
        printf("scatter %.1f us | hist(just written) %.1f | hist(same again) %.1f | hist(other, cold buffer) %.1f\n", s * 1e3, h1 * 1e3,
               h2 * 1e3, h3 * 1e3);
    }
    // which property makes a buffer slow to read: being the scatter's most recent output, or having been scattered into at all?
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t *x = ev;
        CK(vrs::launch_scatter(0, d_a, d_b, nullptr, nullptr, sc.offsets, n, 0, W, B, true, cfg));
        CK(vrs::launch_scatter(0, d_a, d_c, nullptr, nullptr, sc.offsets, n, 0, W, B, true, cfg));
        CK(vrs::launch_histograms(0, d_b, d_hist, n, 8, W, B, {x[0], x[1]}));
        CK(vrs::launch_histograms(0, d_c, d_hist, n, 8, W, B, {x[2], x[3]}));
        CK(vrs::launch_histograms(0, d_a, d_hist, n, 8, W, B, {x[4], x[5]}));
        CK(vrs::launch_histograms(0, d_b, d_hist, n, 8, W, B, {x[6], x[7]}));
        CK(hipDeviceSynchronize());
        float t0, t1, t2, t3;
        CK(hipEventElapsedTime(&t0, x[0], x[1])); CK(hipEventElapsedTime(&t1, x[2], x[3]));
        CK(hipEventElapsedTime(&t2, x[4], x[5])); CK(hipEventElapsedTime(&t3, x[6], x[7]));
        printf("after scatter a->b, a->c: hist(b) %.1f | hist(c) %.1f | hist(a, never scattered into) %.1f | hist(b) %.1f\n", t0 * 1e3, t1 * 1e3, t2 * 1e3, t3 * 1e3);
    }
    // rewrite b sequentially with a plain copy of its own content order-preserved? (copy c->b): is b fast again?
    CK(hipMemcpy(d_b, d_c, (size_t)n * 4, hipMemcpyDeviceToDevice));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep) {
        CK(vrs::launch_histograms(0, d_b, d_hist, n, 8, W, B, {ev[0], ev[1]}));
        CK(vrs::launch_histograms(0, d_c, d_hist, n, 8, W, B, {ev[2], ev[3]}));
        CK(hipDeviceSynchronize());
        float t0, t1;
        CK(hipEventElapsedTime(&t0, ev[0], ev[1])); CK(hipEventElapsedTime(&t1, ev[2], ev[3]));
        printf("after memcpy c->b (same scattered DATA, sequentially written): hist(b) %.1f | hist(c) %.1f\n", t0 * 1e3, t1 * 1e3);
    }
    return 0;
}
