// Lab: two-launch prefix time vs chunk size C (W = 12208 rows).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../tools/lab/vrs_all_kernels.hip"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
int main() {
    const uint32_t W = 12208;
    std::vector<uint32_t> h((size_t)W * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 32 + (i * 2654435761u >> 29);
    uint32_t *d_hist, *d_off, *d_cs;
    CK(hipMalloc(&d_hist, h.size() * 4)); CK(hipMalloc(&d_off, h.size() * 4)); CK(hipMalloc(&d_cs, h.size() * 4));
    CK(hipMemcpy(d_hist, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b, c;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&c));
    for (uint32_t C : {16u, 32u, 64u, 128u, 256u}) {
        const uint32_t G = (W + C - 1) / C;
        float best1 = 1e9, best2 = 1e9;
        for (int r = 0; r < 8; ++r) {
            hipExtLaunchKernelGGL(vrs::chunk_sum_kernel, dim3(G), dim3(1024), 0, 0, a, b, 0, (const uint32_t *)d_hist, d_cs, W, C);
            hipExtLaunchKernelGGL(vrs::offsets_kernel, dim3(G), dim3(1024), 0, 0, nullptr, c, 0, (const uint32_t *)d_hist, (const uint32_t *)d_cs, d_off, W, C, G);
            CK(hipDeviceSynchronize());
            float t1, t2;
            CK(hipEventElapsedTime(&t1, a, b)); CK(hipEventElapsedTime(&t2, a, c));
            if (r > 1) { best1 = t1 < best1 ? t1 : best1; best2 = t2 < best2 ? t2 : best2; }
        }
        printf("C=%3u G=%4u: chunk_sum %.1f us, both %.1f us\n", C, G, best1 * 1e3, best2 * 1e3);
    }
    return 0;
}
