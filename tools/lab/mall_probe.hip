// lab: what does the 256 MB Infinity Cache give a producer -> consumer pair of kernels?
//   1. read rate over a window of W MB read again and again (W from 32 MB to 1 GB)
//   2. write a region, then read it: whole region at once vs chunk by chunk (producer and consumer alternate)
//   3. in-place update (read + write the same lines) right after a kernel wrote them, chunked vs whole
// build: hipcc --offload-arch=gfx950 -O3 tools/lab/mall_probe.hip -o gpurun_out/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void read_kernel(const uint4 *p, size_t vecs, uint32_t *sink) {
    uint32_t acc = 0;
    const size_t stride = static_cast<size_t>(gridDim.x) * 256 * 4;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 1024 + threadIdx.x; i < vecs; i += stride) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t j = i + r * 256;
            if (j < vecs) { const uint4 v = p[j]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
        }
    }
    if (acc == 0x12345u) *sink = acc;
}
__global__ __launch_bounds__(256) void write_kernel(uint4 *p, size_t vecs, uint32_t seed) {
    const size_t stride = static_cast<size_t>(gridDim.x) * 256 * 4;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 1024 + threadIdx.x; i < vecs; i += stride) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t j = i + r * 256;
            if (j < vecs) p[j] = make_uint4(seed, static_cast<uint32_t>(j), seed ^ 7u, 3u);
        }
    }
}
__global__ __launch_bounds__(256) void update_kernel(uint4 *p, size_t vecs) {
    const size_t stride = static_cast<size_t>(gridDim.x) * 256 * 4;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 1024 + threadIdx.x; i < vecs; i += stride) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t j = i + r * 256;
            if (j < vecs) { uint4 v = p[j]; v.x += 1u; v.y ^= v.x; v.z += v.y; v.w ^= v.z; p[j] = v; }
        }
    }
}
__global__ __launch_bounds__(256) void copy_kernel(const uint4 *s, uint4 *d, size_t vecs) {
    const size_t stride = static_cast<size_t>(gridDim.x) * 256 * 4;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 1024 + threadIdx.x; i < vecs; i += stride) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t j = i + r * 256;
            if (j < vecs) d[j] = s[j];
        }
    }
}

int main() {
    const size_t MB = 1 << 20, total = 1536 * MB;
    uint4 *buf; uint32_t *sink;
    CK(hipMalloc(&buf, total)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, total));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    auto grid_for = [&](size_t bytes) { size_t g = bytes / 16 / 1024; return static_cast<int>(g < static_cast<size_t>(grid) ? (g ? g : 1) : grid); };
    float ms;
    printf("1. read a window again and again (20 reads)\n");
    for (size_t w : {32, 64, 128, 192, 256, 384, 512, 1024}) {
        const size_t bytes = w * MB;
        for (int i = 0; i < 3; ++i) read_kernel<<<grid_for(bytes), 256>>>(buf, bytes / 16, sink);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) read_kernel<<<grid_for(bytes), 256>>>(buf, bytes / 16, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("   window %5zu MB: %7.1f us per read, %6.2f TB/s\n", w, ms * 1e3 / 20, bytes * 20.0 / (ms * 1e-3) / 1e12);
    }
    printf("2. write 400 MB, then read it; chunked = write chunk c, read chunk c, ... (a fresh 400 MB region every repetition)\n");
    const size_t region = 400 * MB;
    for (size_t c : {400, 200, 100, 50, 25, 12}) {
        const size_t chunk = c * MB;
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            uint4 *base = buf + (static_cast<size_t>(rep % 3) * 512 * MB) / 16;
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (size_t off = 0; off < region; off += chunk) {
                const size_t len = region - off < chunk ? region - off : chunk;
                write_kernel<<<grid_for(len), 256>>>(base + off / 16, len / 16, rep);
                read_kernel<<<grid_for(len), 256>>>(base + off / 16, len / 16, sink);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("   chunk %4zu MB: %7.1f us for write + read of 400 MB, %6.2f TB/s of kernel traffic\n", c, best * 1e3, 2.0 * region / (best * 1e-3) / 1e12);
    }
    printf("3. copy 400 MB A -> B, then update B in place (read + write); chunked the same way\n");
    for (size_t c : {400, 200, 100, 50, 25, 12}) {
        const size_t chunk = c * MB;
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            uint4 *a = buf, *b = buf + (512 * MB + static_cast<size_t>(rep % 2) * 512 * MB) / 16;
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (size_t off = 0; off < region; off += chunk) {
                const size_t len = region - off < chunk ? region - off : chunk;
                copy_kernel<<<grid_for(len), 256>>>(a + off / 16, b + off / 16, len / 16);
                update_kernel<<<grid_for(len), 256>>>(b + off / 16, len / 16);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("   chunk %4zu MB: %7.1f us for copy + update of 400 MB, %6.2f TB/s of kernel traffic (16 B/key)\n", c, best * 1e3, 4.0 * region / (best * 1e-3) / 1e12);
    }
    printf("4. the same with the kernels of ALL chunks' copies first, then all updates (what the sort does today)\n");
    {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            uint4 *a = buf, *b = buf + (512 * MB + static_cast<size_t>(rep % 2) * 512 * MB) / 16;
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            copy_kernel<<<grid, 256>>>(a, b, region / 16);
            update_kernel<<<grid, 256>>>(b, region / 16);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("   whole: %7.1f us\n", best * 1e3);
    }
    return 0;
}
