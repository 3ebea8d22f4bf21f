// pool_lab.hip -- times the kernels of the pool form (vrs_msd_pool.hip) one by one on uniform random keys, outside the library:
// lab builds with -D knobs compare variants of ONE kernel in one gpurun call.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I vkradixsort_amd/csrc [-DVRS_POOL_...] tools/lab/pool_lab.hip -o tools/lab/pool_lab_X
//   tools/lab/pool_lab_X [n] [reps]
#include "../../vkradixsort_amd/csrc/vrs_msd_pool.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));                 \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

__global__ void fill_random(uint32_t *k, uint32_t n, uint32_t seed) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t x = i * 0x9E3779B9u + seed;
        x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
        k[i] = x;
    }
}
__global__ void xcc_probe(uint32_t *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = vrs::xcc_id();
}

int main(int argc, char **argv) {
    const uint32_t n = argc > 1 ? static_cast<uint32_t>(atof(argv[1])) : 100000000u;
    const int reps = argc > 2 ? atoi(argv[2]) : 6;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    uint32_t *keys[4], *partner, *home, *ovf, *hist, *xo;
    vrs::PoolPlan *pool;
    vrs::MsdPlan *msd;
    vrs::OnesweepPlanHead *head;
    const uint32_t room = vrs::pool_overflow_capacity(n);
    for (auto &k : keys) CK(hipMalloc(&k, 4ull * n));
    CK(hipMalloc(&partner, 4ull * n));
    CK(hipMalloc(&home, 4ull * n));
    CK(hipMalloc(&ovf, 4ull * room));
    CK(hipMalloc(&hist, 4ull * vrs::kMsdCountWords));
    CK(hipMalloc(&pool, sizeof(vrs::PoolPlan)));
    CK(hipMalloc(&msd, sizeof(vrs::MsdPlan)));
    CK(hipMalloc(&head, sizeof(vrs::OnesweepPlanHead)));
    CK(hipMalloc(&xo, 4 * 64));
    CK(hipMemset(hist, 0, 4ull * vrs::kMsdCountWords));
    CK(hipMemset(pool, 0, sizeof(vrs::PoolPlan)));
    CK(hipMemset(msd, 0, sizeof(vrs::MsdPlan)));
    for (int b = 0; b < 4; ++b) hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, st, keys[b], n, 12345u + b);
    hipLaunchKernelGGL(xcc_probe, dim3(64), dim3(512), 0, st, xo);
    uint32_t hx[64];
    CK(hipMemcpyAsync(hx, xo, sizeof hx, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    unsigned long long xcc_map = 0;
    for (int b = 0; b < 8; ++b) xcc_map |= static_cast<unsigned long long>(hx[b] & 0xFF) << (8 * b);
    const vrs::PoolStreams ps = vrs::pool_streams(n);
#ifdef VRS_POOL_LAB_MARKS
    {
        unsigned long long *dm;
        CK(hipMalloc(&dm, 8ull * 12 * 8 * 64));
        CK(hipMemset(dm, 0, 8ull * 12 * 8 * 64));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(vrs::g_pool_marks), &dm, sizeof dm));
    }
#endif
    hipEvent_t ev[6];
    for (auto &e : ev) CK(hipEventCreate(&e));
    const uint32_t tiles_b_cap = vrs::pool_tiles_b_cap(n, false);
    double sum[5] = {0, 0, 0, 0, 0};
    int counted = 0;
    for (int r = 0; r < reps + 2; ++r) {
        const uint32_t *in = keys[r % 4];
        CK(hipMemsetAsync(reinterpret_cast<char *>(msd) + offsetof(vrs::MsdPlan, cursor_a), 0, vrs::kMsdCursorBytes, st));
        CK(hipEventRecord(ev[0], st));
        CK(vrs::launch_pool_sample(st, in, n, 0, ps, pool, room));
        CK(hipEventRecord(ev[1], st));
        CK(vrs::launch_pool_pass_a(st, in, partner, ovf, n, 0, ps, pool, msd, hist, xcc_map, 256, false, room));
        CK(hipEventRecord(ev[2], st));
        CK(vrs::launch_pool_plan(st, hist, msd, pool, head, nullptr, 1, n, tiles_b_cap, 14333, nullptr));
        CK(hipEventRecord(ev[3], st));
        CK(vrs::launch_pool_pass_b(st, partner, ovf, home, msd, pool, tiles_b_cap, xcc_map, 0));
        CK(hipEventRecord(ev[4], st));
        CK(hipStreamSynchronize(st));
        vrs::MsdPlan hm;
        CK(hipMemcpy(&hm, msd, 16, hipMemcpyDeviceToHost));
        float t[4];
        for (int i = 0; i < 4; ++i) CK(hipEventElapsedTime(&t[i], ev[i], ev[i + 1]));
        std::printf("rep %d: sample %.1f  passA %.1f  plan %.1f  passB %.1f us   ok=%u shift=%u\n", r, t[0] * 1e3, t[1] * 1e3, t[2] * 1e3, t[3] * 1e3, hm.ok, hm.shift);
        if (r >= 2) {
            for (int i = 0; i < 4; ++i) sum[i] += t[i] * 1e3;
            ++counted;
        }
    }
#ifdef VRS_POOL_LAB_MARKS
    {
        const uint32_t wgs = 8u * std::min<uint32_t>(64u, ps.tiles_per_stream);
        std::vector<unsigned long long> m(static_cast<size_t>(wgs) * 12);
        unsigned long long *dm;
        CK(hipMemcpyFromSymbol(&dm, HIP_SYMBOL(vrs::g_pool_marks), sizeof dm));
        CK(hipMemcpy(m.data(), dm, m.size() * 8, hipMemcpyDeviceToHost));
        double s[12] = {};
        for (uint32_t w = 0; w < wgs; ++w)
            for (int k = 0; k < 12; ++k) s[k] += static_cast<double>(m[static_cast<size_t>(w) * 12 + k]);
        const char *nm[12] = {"prev-end->start", "zero+B1", "rank issue", "B2 wait", "scan+atomic+prefetch", "B4 wait", "rebucket+gbase", "B5 wait", "writeout issue", "loop end->flush", "flush", "-"};
        double tot = 0;
        for (int k = 0; k < 11; ++k) tot += s[k];
        for (int k = 0; k < 11; ++k) std::printf("   mark %2d %-22s %10.0f ticks per workgroup  %5.1f %%\n", k, nm[k], s[k] / wgs, 100.0 * s[k] / tot);
        std::printf("   total %.0f ticks per workgroup\n", tot / wgs);
    }
#endif
    std::printf("AVG n=%u: sample %.1f  passA %.1f  plan %.1f  passB %.1f us (events bracket each launch: +~2 us each)\n", n, sum[0] / counted, sum[1] / counted,
                sum[2] / counted, sum[3] / counted);
    return 0;
}
