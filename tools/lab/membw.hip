// Lab: achievable HBM bandwidth on this MI355X for the access shapes the sort uses.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/membw.hip -o tools/lab/membw
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// read-only: one workgroup per 32 KiB tile (like the histogram kernel), 8 x 16 B per thread
__global__ __launch_bounds__(256) void read_tile(const uint4 *__restrict__ in, uint32_t *out, size_t nvec) {
    const size_t base = (size_t)blockIdx.x * 2048 + threadIdx.x;
    uint4 q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) q[u] = base + u * 256 < nvec ? in[base + u * 256] : make_uint4(0, 0, 0, 0);
    uint32_t s = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += q[u].x ^ q[u].y ^ q[u].z ^ q[u].w;
    if (s == 0x12345678u) out[0] = s;
}
// read-only: persistent grid-stride, UNROLL x 16 B per thread per step
template <int UNROLL>
__global__ __launch_bounds__(256) void read_stride(const uint4 *__restrict__ in, uint32_t *out, size_t nvec) {
    uint32_t s = 0;
    const size_t step = (size_t)gridDim.x * 256 * UNROLL;
    for (size_t i0 = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i0 < nvec; i0 += step) {
        uint4 q[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) q[u] = i0 + u * 256 < nvec ? in[i0 + u * 256] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) s += q[u].x ^ q[u].y ^ q[u].z ^ q[u].w;
    }
    if (s == 0x12345678u) out[0] = s;
}
// read-only, 4 B per lane (like the scatter's wave-striped loads): 32 x dword per thread per 32 KiB tile
__global__ __launch_bounds__(256) void read_tile_dword(const uint32_t *__restrict__ in, uint32_t *out, size_t n) {
    const size_t base = (size_t)blockIdx.x * 8192 + (threadIdx.x >> 6) * 2048 + (threadIdx.x & 63);
    uint32_t q[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) q[u] = base + u * 64 < n ? in[base + u * 64] : 0u;
    uint32_t s = 0;
#pragma unroll
    for (int u = 0; u < 32; ++u) s ^= q[u];
    if (s == 0x12345678u) out[0] = s;
}
__global__ __launch_bounds__(256) void copy_stride(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t nvec) {
    const size_t step = (size_t)gridDim.x * 256 * 4;
    for (size_t i0 = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i0 < nvec; i0 += step) {
        uint4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = i0 + u * 256 < nvec ? in[i0 + u * 256] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i0 + u * 256 < nvec) out[i0 + u * 256] = q[u];
    }
}
__global__ __launch_bounds__(256) void copy_tile_dword(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, size_t n) {
    const size_t base = (size_t)blockIdx.x * 8192 + (threadIdx.x >> 6) * 2048 + (threadIdx.x & 63);
    uint32_t q[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) q[u] = base + u * 64 < n ? in[base + u * 64] : 0u;
#pragma unroll
    for (int u = 0; u < 32; ++u) if (base + u * 64 < n) out[base + u * 64] = q[u];
}
__global__ __launch_bounds__(256) void write_stride(uint4 *__restrict__ out, size_t nvec) {
    const size_t step = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += step) out[i] = make_uint4(i, 1, 2, 3);
}

// histogram-shaped ablation: MODE 0 = loads only, 1 = + LDS atomics, 2 = + barrier/row store, 3 = LDS atomics on a
// lane-private replicated layout (no bank conflicts), 4 = mode 2 with 2 tiles per workgroup
template <int MODE>
__global__ __launch_bounds__(256) void read_tile_hist(const uint4 *__restrict__ in, uint32_t *out, uint32_t *hist, size_t nvec, uint32_t shift) {
    __shared__ uint32_t s_hist[MODE == 3 ? 8192 : 256];
    const uint32_t tid = threadIdx.x;
    if (MODE == 3) { for (int i = tid; i < 8192; i += 256) s_hist[i] = 0; } else s_hist[tid] = 0;
    __syncthreads();
    const int tiles = MODE == 4 ? 2 : 1;
    uint32_t s = 0;
    for (int t = 0; t < tiles; ++t) {
        const size_t base = ((size_t)blockIdx.x * tiles + t) * 2048 + tid;
        uint4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = base + u * 256 < nvec ? in[base + u * 256] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) s += q[u].x ^ q[u].y ^ q[u].z ^ q[u].w;
            else if (MODE == 3) {
                atomicAdd(&s_hist[(((q[u].x >> shift) & 255u) << 5) | (tid & 31u)], 1u);
                atomicAdd(&s_hist[(((q[u].y >> shift) & 255u) << 5) | (tid & 31u)], 1u);
                atomicAdd(&s_hist[(((q[u].z >> shift) & 255u) << 5) | (tid & 31u)], 1u);
                atomicAdd(&s_hist[(((q[u].w >> shift) & 255u) << 5) | (tid & 31u)], 1u);
            } else {
                atomicAdd(&s_hist[(q[u].x >> shift) & 255u], 1u);
                atomicAdd(&s_hist[(q[u].y >> shift) & 255u], 1u);
                atomicAdd(&s_hist[(q[u].z >> shift) & 255u], 1u);
                atomicAdd(&s_hist[(q[u].w >> shift) & 255u], 1u);
            }
        }
        if (MODE == 2 || MODE == 4) {
            __syncthreads();
            hist[((size_t)blockIdx.x * tiles + t) * 256 + tid] = s_hist[tid];
            if (tiles > 1) { s_hist[tid] = 0; __syncthreads(); }
        }
    }
    if (MODE == 1 || MODE == 3) { __syncthreads(); s = s_hist[tid]; }
    if (s == 0x12345678u) out[0] = s;
}

// MALL probe: write tiles in ascending order, then read them ascending or descending
__global__ __launch_bounds__(256) void write_tile(uint4 *out, size_t nvec, uint32_t salt) {
    const size_t base = (size_t)blockIdx.x * 2048 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 8; ++u) if (base + u * 256 < nvec) out[base + u * 256] = make_uint4(salt, base, u, 3);
}
__global__ __launch_bounds__(256) void read_tile_dir(const uint4 *__restrict__ in, uint32_t *out, size_t nvec, int reverse) {
    const size_t blk = reverse ? (gridDim.x - 1 - blockIdx.x) : blockIdx.x;
    const size_t base = blk * 2048 + threadIdx.x;
    uint4 q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) q[u] = base + u * 256 < nvec ? in[base + u * 256] : make_uint4(0, 0, 0, 0);
    uint32_t s = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += q[u].x ^ q[u].y ^ q[u].z ^ q[u].w;
    if (s == 0x12345678u) out[0] = s;
}

template <typename F>
void timeit(const char *name, double bytes, F launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9, tot = 0;
    for (int r = 0; r < 7; ++r) {
        CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) { best = ms < best ? ms : best; tot += ms; }
    }
    printf("%-44s min %8.1f us  avg %8.1f us  %6.2f TB/s (min)\n", name, best * 1e3, tot / 5 * 1e3, bytes / best / 1e9);
}

int main(int argc, char **argv) {
    size_t n = argc > 1 ? (size_t)atof(argv[1]) : 100000000;  // uint32 elements
    uint32_t *a, *b, *o;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&o, 64));
    CK(hipMemset(a, 1, n * 4)); CK(hipMemset(b, 2, n * 4));
    size_t nvec = n / 4;
    const double B = n * 4.0;
    printf("buffer = %.0f MB\n", B / 1e6);
    timeit("read  tile/WG 8x16B (hist shape)", B, [&] { hipLaunchKernelGGL(read_tile, dim3((nvec + 2047) / 2048), dim3(256), 0, 0, (const uint4 *)a, o, nvec); });
    uint32_t *hh; CK(hipMalloc(&hh, ((nvec + 2047) / 2048 + 2) * 1024));
    {
        // random data for the histogram ablations
        uint32_t *tmp = (uint32_t *)malloc(n * 4); uint32_t x = 12345; for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; tmp[i] = x; }
        CK(hipMemcpy(a, tmp, n * 4, hipMemcpyHostToDevice)); free(tmp);
    }
    const unsigned nt = (nvec + 2047) / 2048;
    timeit("hist-shape 0: loads only", B, [&] { hipLaunchKernelGGL(read_tile_hist<0>, dim3(nt), dim3(256), 0, 0, (const uint4 *)a, o, hh, nvec, 16u); });
    timeit("hist-shape 1: + ds_add per key", B, [&] { hipLaunchKernelGGL(read_tile_hist<1>, dim3(nt), dim3(256), 0, 0, (const uint4 *)a, o, hh, nvec, 16u); });
    timeit("hist-shape 2: + barrier + row store", B, [&] { hipLaunchKernelGGL(read_tile_hist<2>, dim3(nt), dim3(256), 0, 0, (const uint4 *)a, o, hh, nvec, 16u); });
    timeit("hist-shape 3: ds_add on x32 replicated bins", B, [&] { hipLaunchKernelGGL(read_tile_hist<3>, dim3(nt), dim3(256), 0, 0, (const uint4 *)a, o, hh, nvec, 16u); });
    timeit("hist-shape 4: mode 2, 2 tiles per WG", B, [&] { hipLaunchKernelGGL(read_tile_hist<4>, dim3((nt + 1) / 2), dim3(256), 0, 0, (const uint4 *)a, o, hh, nvec, 16u); });
    for (int rev = 0; rev < 2; ++rev) {
        hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
        float bw = 1e9, br = 1e9;
        for (int r = 0; r < 6; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(write_tile, dim3(nt), dim3(256), 0, 0, (uint4 *)b, nvec, (uint32_t)r);
            CK(hipEventRecord(e1, 0));
            hipLaunchKernelGGL(read_tile_dir, dim3(nt), dim3(256), 0, 0, (const uint4 *)b, o, nvec, rev);
            CK(hipEventRecord(e2, 0)); CK(hipEventSynchronize(e2));
            float w, rd; CK(hipEventElapsedTime(&w, e0, e1)); CK(hipEventElapsedTime(&rd, e1, e2));
            if (r > 0) { bw = w < bw ? w : bw; br = rd < br ? rd : br; }
        }
        printf("MALL probe: write tiles ascending %.1f us, then read %s %.1f us\n", bw * 1e3, rev ? "DESCENDING" : "ascending ", br * 1e3);
    }
    timeit("read  tile/WG 32x4B (scatter load shape)", B, [&] { hipLaunchKernelGGL(read_tile_dword, dim3((n + 8191) / 8192), dim3(256), 0, 0, a, o, n); });
    for (int g : {1024, 2048, 4096}) {
        char nm[64];
        snprintf(nm, 64, "read  grid-stride x4, grid=%d", g);
        timeit(nm, B, [&] { hipLaunchKernelGGL(read_stride<4>, dim3(g), dim3(256), 0, 0, (const uint4 *)a, o, nvec); });
        snprintf(nm, 64, "read  grid-stride x8, grid=%d", g);
        timeit(nm, B, [&] { hipLaunchKernelGGL(read_stride<8>, dim3(g), dim3(256), 0, 0, (const uint4 *)a, o, nvec); });
    }
    timeit("copy  grid-stride x4 16B, grid=2048 (r+w)", 2 * B, [&] { hipLaunchKernelGGL(copy_stride, dim3(2048), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, nvec); });
    timeit("copy  grid-stride x4 16B, grid=8192 (r+w)", 2 * B, [&] { hipLaunchKernelGGL(copy_stride, dim3(8192), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, nvec); });
    timeit("copy  tile/WG 32x4B (r+w, scatter shape)", 2 * B, [&] { hipLaunchKernelGGL(copy_tile_dword, dim3((n + 8191) / 8192), dim3(256), 0, 0, a, b, n); });
    timeit("write grid-stride 16B, grid=4096", B, [&] { hipLaunchKernelGGL(write_stride, dim3(4096), dim3(256), 0, 0, (uint4 *)b, nvec); });
    timeit("hipMemcpyDtoD (r+w)", 2 * B, [&] { CK(hipMemcpyAsync(b, a, n * 4, hipMemcpyDeviceToDevice, 0)); });
    return 0;
}
