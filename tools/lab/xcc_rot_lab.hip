// xcc_rot_lab.hip -- does "block b runs on XCC b % 8" hold for EVERY launch, or does the dispatcher carry on where the previous
// kernel of the queue stopped?  Launches a filler kernel of G workgroups, then a probe kernel, for G = 0..17, on one stream.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3u << 11) | 20u); }
__global__ void probe(uint32_t *out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }
__global__ void filler(uint32_t *sink) { if (threadIdx.x == 0 && sink) atomicAdd(sink, 1u); }
int main() {
    hipStream_t st, st2;
    CK(hipStreamCreate(&st));
    CK(hipStreamCreate(&st2));
    uint32_t *d, *sink, h[64];
    CK(hipMalloc(&d, 4 * 64));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(sink, 0, 4));
    for (int threads : {64, 256, 512, 1024}) {
        for (int G = 0; G <= 17; ++G) {
            if (G) hipLaunchKernelGGL(filler, dim3(G), dim3(threads), 0, st, sink);
            hipLaunchKernelGGL(probe, dim3(64), dim3(512), 0, st, d);
            CK(hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            printf("filler %2d x %4d threads, then probe:", G, threads);
            for (int i = 0; i < 10; ++i) printf(" %u", h[i]);
            bool rule = true;
            for (int i = 8; i < 64; ++i) rule &= h[i] == h[i & 7];
            printf("   (b %% 8 rule inside the launch: %s)\n", rule ? "holds" : "BROKEN");
        }
    }
    // two probes back to back without anything between
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(probe, dim3(64), dim3(512), 0, st, d);
        CK(hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        printf("probe alone:");
        for (int i = 0; i < 10; ++i) printf(" %u", h[i]);
        printf("\n");
    }
    // another stream in between
    hipLaunchKernelGGL(filler, dim3(3), dim3(256), 0, st2, sink);
    CK(hipStreamSynchronize(st2));
    hipLaunchKernelGGL(probe, dim3(64), dim3(512), 0, st, d);
    CK(hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    printf("3 filler workgroups on ANOTHER stream, then probe:");
    for (int i = 0; i < 10; ++i) printf(" %u", h[i]);
    printf("\n");
    // a device-to-device copy (blit kernel) in between
    uint32_t *a, *b;
    CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20));
    CK(hipMemcpyAsync(b, a, (1 << 20) - 12, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(probe, dim3(64), dim3(512), 0, st, d);
    CK(hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    printf("a 1 MB device copy on the stream, then probe:");
    for (int i = 0; i < 10; ++i) printf(" %u", h[i]);
    printf("\n");
    CK(hipMemsetAsync(a, 0, 12345, st));
    hipLaunchKernelGGL(probe, dim3(64), dim3(512), 0, st, d);
    CK(hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    printf("a memset on the stream, then probe:");
    for (int i = 0; i < 10; ++i) printf(" %u", h[i]);
    printf("\n");
    return 0;
}
