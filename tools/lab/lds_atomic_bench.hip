// Lab: LDS atomic / read / write throughput on gfx950 by address pattern.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/lds_atomic_bench.hip -o tools/lab/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// OP: 0 ds_add (no rtn), 1 ds_add_rtn, 2 ds_read, 3 ds_write
// PAT: 0 random digit (256 bins), 1 conflict-free (lane), 2 same address, 3 random over 8192 words, 4 replicated x32 (digit*32 + lane%32)
template <int OP, int PAT>
__global__ __launch_bounds__(1024) void k(uint32_t *out, int iters, unsigned long long *cyc) {
    __shared__ uint32_t s[8192];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += blockDim.x) s[i] = 0;
    __syncthreads();
    uint32_t x = tid * 2654435761u + blockIdx.x * 97u + 12345u;
    uint32_t acc = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        uint32_t a[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            x = x * 1664525u + 1013904223u;
            uint32_t r = x >> 8;
            if (PAT == 0) a[u] = r & 255u;
            else if (PAT == 1) a[u] = lane + ((r & 15u) << 6);
            else if (PAT == 2) a[u] = 7;
            else if (PAT == 3) a[u] = r & 8191u;
            else a[u] = ((r & 255u) << 5) | (lane & 31u);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (OP == 0) __hip_atomic_fetch_add(&s[a[u]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == 1) acc += __hip_atomic_fetch_add(&s[a[u]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (OP == 2) acc += ((volatile uint32_t *)s)[a[u]];
            else ((volatile uint32_t *)s)[a[u]] = x;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (OP == 0 || OP == 3) { __syncthreads(); acc = s[tid]; }
    out[blockIdx.x * blockDim.x + tid] = acc;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP, int PAT>
void run(const char *name, int threads) {
    uint32_t *out; unsigned long long *cyc;
    const int blocks = 256, iters = 200;
    CK(hipMalloc(&out, blocks * 1024 * 4)); CK(hipMalloc(&cyc, blocks * 8));
    hipLaunchKernelGGL((k<OP, PAT>), dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k<OP, PAT>), dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    CK(hipDeviceSynchronize());
    unsigned long long h[256]; CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    const double wave_instrs = double(threads / 64) * iters * 16;  // per CU (1 block per CU)
    printf("%-34s waves/CU=%2d  %7.1f cycles per wave-instruction (CU level)  %6.2f lanes/clk\n", name, threads / 64,
           avg / wave_instrs, 64.0 * wave_instrs / avg);
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
    for (int threads : {256, 1024}) {
        run<0, 0>("ds_add      random 256 bins", threads);
        run<0, 1>("ds_add      conflict-free", threads);
        run<0, 2>("ds_add      same address", threads);
        run<0, 3>("ds_add      random 8192 words", threads);
        run<0, 4>("ds_add      256 bins x32 replicated", threads);
        run<1, 0>("ds_add_rtn  random 256 bins", threads);
        run<1, 1>("ds_add_rtn  conflict-free", threads);
        run<1, 2>("ds_add_rtn  same address", threads);
        run<1, 3>("ds_add_rtn  random 8192 words", threads);
        run<2, 0>("ds_read     random 256 bins", threads);
        run<2, 1>("ds_read     conflict-free", threads);
        run<2, 3>("ds_read     random 8192 words", threads);
        run<3, 1>("ds_write    conflict-free", threads);
        run<3, 3>("ds_write    random 8192 words", threads);
    }
    return 0;
}
