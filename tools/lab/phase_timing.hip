// Lab: where does a scatter workgroup spend its time?  Builds the product kernels with VRS_MARK
// hooks that stamp s_memtime per phase, runs one pass at N keys and prints mean phase durations.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I vkradixsort_amd/csrc tools/lab/phase_timing.hip -o /tmp/phase_timing
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

__device__ unsigned long long *g_marks;  // [W][8]
__shared__ unsigned long long s_marks[8];  // stamped in LDS, flushed once at the end: no global round trips inside the phases
#define VRS_MARK(i)                                                         \
    do {                                                                    \
        if (threadIdx.x == 0) s_marks[(i)] = __builtin_readcyclecounter();  \
    } while (0)
#define VRS_MARK_FLUSH()                                                                     \
    do {                                                                                     \
        if (threadIdx.x == 0) {                                                              \
            s_marks[6] = __builtin_readcyclecounter();                                       \
            for (int i_ = 0; i_ < 7; ++i_) g_marks[(size_t)blockIdx.x * 8 + i_] = s_marks[i_]; \
        }                                                                                    \
    } while (0)
#include "../../tools/lab/vrs_all_kernels.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int ITEMS, int WAVES, int RANK, int OCC>
void run(const char *name, uint32_t n, uint32_t B, uint32_t *d_in, uint32_t *d_out, uint32_t *d_hist, vrs::PrefixScratch sc) {
    uint32_t gis = n / B + (n % B ? 1 : 0), W = (gis + 255) / 256;
    unsigned long long *d_marks;
    CK(hipMalloc(&d_marks, (size_t)W * 8 * 8));
    CK(hipMemset(d_marks, 0, (size_t)W * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_marks), &d_marks, sizeof(d_marks)));
    CK(vrs::launch_histograms(0, d_in, d_hist, n, 0, W, B));
    CK(vrs::launch_prefix(0, d_hist, sc, W));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((vrs::scatter_kernel<uint32_t, ITEMS, WAVES, false, RANK, OCC>), dim3(W), dim3(WAVES * 64), 0, 0, d_in, d_out,
                           (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)sc.offsets, n, 0u, W, B, 1, (const uint32_t *)nullptr, 1u, (const uint32_t *)nullptr, 0u);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> m((size_t)W * 8);
    CK(hipMemcpy(m.data(), d_marks, m.size() * 8, hipMemcpyDeviceToHost));
    double sum[8] = {0}; unsigned long long tmin = ~0ull, tmax = 0;
    for (uint32_t w = 0; w < W; ++w) {
        for (int i = 1; i <= 5; ++i) sum[i] += double(m[(size_t)w * 8 + i] - m[(size_t)w * 8 + i - 1]);
        if (m[(size_t)w * 8] < tmin) tmin = m[(size_t)w * 8];
        if (m[(size_t)w * 8 + 5] > tmax) tmax = m[(size_t)w * 8 + 5];
    }
    printf("%s: kernel %.1f us, W=%u; s_memtime ticks (100MHz => 10ns): span=%llu | load+zero %.0f | rank %.0f | scan %.0f | lds-scatter %.0f | write-out issue %.0f  (sum %.0f)\n",
           name, ms * 1e3, W, tmax - tmin, sum[1] / W, sum[2] / W, sum[3] / W, sum[4] / W, sum[5] / W,
           (sum[1] + sum[2] + sum[3] + sum[4] + sum[5]) / W);
    CK(hipFree(d_marks));
}

int main(int argc, char **argv) {
    uint32_t n = argc > 1 ? (uint32_t)atof(argv[1]) : 100000000u;
    std::vector<uint32_t> h(n);
    std::mt19937 gen(1);
    for (auto &x : h) x = gen();
    uint32_t *d_in, *d_out, *d_hist;
    CK(hipMalloc(&d_in, (size_t)n * 4)); CK(hipMalloc(&d_out, (size_t)n * 4));
    CK(hipMemcpy(d_in, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    uint32_t Wmax = (n / 8 + 255) / 256 + 1;
    CK(hipMalloc(&d_hist, (size_t)Wmax * 1024));
    vrs::PrefixScratch sc;
    CK(hipMalloc(&sc.offsets, (size_t)Wmax * 1024)); CK(hipMalloc(&sc.chunk_sums, (size_t)Wmax * 1024));
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (uint32_t B : {32u, 64u, 16u}) {
            uint32_t gis = n / B + (n % B ? 1 : 0), W = (gis + 255) / 256;
            for (uint32_t shift : {0u, 24u}) {
                float best = 1e9;
                for (int r = 0; r < 6; ++r) {
                    CK(hipEventRecord(e0, 0));
                    CK(vrs::launch_histograms(0, d_in, d_hist, n, shift, W, B));
                    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r > 0 && ms < best) best = ms;
                }
                printf("standalone histogram kernel B=%u shift=%u: %.1f us\n", B, shift, best * 1e3);
                best = 1e9;
                for (int r = 0; r < 6; ++r) {
                    CK(hipEventRecord(e0, 0));
                    CK(vrs::launch_prefix(0, d_hist, sc, W));
                    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r > 0 && ms < best) best = ms;
                }
                printf("standalone prefix kernels  B=%u: %.1f us\n", B, best * 1e3);
            }
        }
    }
    run<32, 4, 1, 3>("32x4 atomic occ3", n, 32, d_in, d_out, d_hist, sc);
    run<32, 4, 0, 3>("32x4 ballot occ3", n, 32, d_in, d_out, d_hist, sc);
    run<32, 4, 1, 4>("32x4 atomic occ4", n, 32, d_in, d_out, d_hist, sc);
    run<32, 4, 0, 4>("32x4 ballot occ4", n, 32, d_in, d_out, d_hist, sc);
    run<16, 8, 1, 4>("16x8 atomic occ4", n, 32, d_in, d_out, d_hist, sc);
    run<16, 8, 0, 4>("16x8 ballot occ4", n, 32, d_in, d_out, d_hist, sc);
    run<16, 4, 1, 4>("16x4 atomic B=16", n, 16, d_in, d_out, d_hist, sc);
    return 0;
}
