// pool_lab.hip -- times the kernels of the pool form (vrs_msd_pool.hip) one by one on uniform random keys, outside the library:
// lab builds with -D knobs compare variants of ONE kernel in one gpurun call.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I vkradixsort_amd/csrc [-DVRS_POOL_...] tools/lab/pool_lab.hip -o tools/lab/pool_lab_X
//   tools/lab/pool_lab_X [n] [reps]
#include "../../vkradixsort_amd/csrc/vrs_msd_pool.hip"
#include "../../vkradixsort_amd/csrc/vrs_msd_pool_local.hip"
#include "../../vkradixsort_amd/csrc/vrs_pool_shape.hip"

#include <cstdio>
#include <unistd.h>
#include <cstdlib>
#include <random>
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

__global__ void check_sorted(const uint32_t *k, uint32_t n, unsigned long long *out) {
    unsigned long long bad = 0, sum = 0;
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        if (i + 1 < n && k[i] > k[i + 1]) ++bad;
        sum += k[i];
    }
    if (bad) atomicAdd(&out[0], bad);
    atomicAdd(&out[1], sum);
}

int main(int argc, char **argv) {
    const uint32_t n = argc > 1 ? static_cast<uint32_t>(atof(argv[1])) : 100000000u;
    const int reps = argc > 2 ? atoi(argv[2]) : 6;
    const int mt_seed = argc > 3 ? atoi(argv[3]) : -1;  // >= 0: keys = std::mt19937(seed) outputs (what numpy's RandomState(seed).randint(0, 2**32) draws)
    std::vector<uint32_t> host_keys;
    if (mt_seed >= 0) {
        host_keys.resize(n);
        std::mt19937 gen(static_cast<uint32_t>(mt_seed));
        for (auto &x : host_keys) x = gen();
    }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    uint32_t *keys[4], *partner, *ovf, *slack, *xo;
    vrs::PoolPlan *pool;
    vrs::MsdPlan *msd;
    vrs::OnesweepPlanHead *head;
    unsigned long long *chk;
    const uint32_t room = vrs::pool_overflow_capacity(n);
    for (auto &k : keys) CK(hipMalloc(&k, 4ull * n));
    CK(hipMalloc(&partner, 4ull * n));
    CK(hipMalloc(&ovf, 4ull * room));
    const vrs::PoolShape shape = vrs::pool_shape(n, getenv("POOL_LAB_SUB") ? atoi(getenv("POOL_LAB_SUB")) : 0);
    const uint32_t slack_cap = vrs::pool_slack_capacity(n, shape.sub_bits);
    std::printf("shape: %u bits, local sort %u (capacity %u)\n", shape.sub_bits, shape.local, vrs::pool_local_capacity(shape.local));
    CK(hipMalloc(&slack, 4ull * slack_cap));
    CK(hipMalloc(&pool, sizeof(vrs::PoolPlan)));
    CK(hipMalloc(&msd, sizeof(vrs::MsdPlan)));
    CK(hipMalloc(&head, sizeof(vrs::OnesweepPlanHead)));
    CK(hipMalloc(&xo, 4 * 64));
    CK(hipMalloc(&chk, 32));
    CK(hipMemset(pool, 0, sizeof(vrs::PoolPlan)));
    CK(hipMemset(msd, 0, sizeof(vrs::MsdPlan)));
    hipLaunchKernelGGL(xcc_probe, dim3(64), dim3(512), 0, st, xo);
    uint32_t hx[64];
    CK(hipMemcpyAsync(hx, xo, sizeof hx, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    unsigned long long xcc_map = 0;
    for (int b = 0; b < 8; ++b) xcc_map |= static_cast<unsigned long long>(hx[b] & 0xFF) << (8 * b);
    const vrs::PoolStreams ps = vrs::pool_streams(n);
    hipEvent_t ev[10];
    for (auto &e : ev) CK(hipEventCreate(&e));
    const uint32_t tiles_b = vrs::pool_tiles_b_cap(n);
    double sum[7] = {0, 0, 0, 0, 0, 0, 0};
    int counted = 0;
    for (int r = 0; r < reps + 2; ++r) {
        uint32_t *in = keys[r % 4];
        if (mt_seed >= 0) CK(hipMemcpyAsync(in, host_keys.data(), 4ull * n, hipMemcpyHostToDevice, st));
        else hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, st, in, n, 12345u + r);
        CK(hipMemsetAsync(chk, 0, 32, st));
        hipLaunchKernelGGL(check_sorted, dim3(2048), dim3(256), 0, st, in, n, chk + 2);  // chk[3] = key sum of the input
        CK(hipMemsetAsync(reinterpret_cast<char *>(msd) + offsetof(vrs::MsdPlan, cursor_a), 0, vrs::kMsdCursorBytes, st));
        CK(hipEventRecord(ev[0], st));
        const uint32_t par = r & 1u;
        CK(vrs::launch_pool_sample(st, in, n, 0, ps, pool, room, par));  // (sample + layout kernels)
        CK(hipEventRecord(ev[1], st));
        CK(vrs::launch_pool_pass_a(st, in, partner, ovf, n, 0, ps, pool, msd, xcc_map, false, room, par));
        CK(hipEventRecord(ev[2], st));
        CK(vrs::launch_pool_plan(st, msd, pool, n, tiles_b, slack_cap, partner, ovf, 0, ps, shape.sub_bits, par));
        CK(hipEventRecord(ev[3], st));
        CK(vrs::launch_pool_pass_b(st, partner, ovf, slack, n, msd, pool, tiles_b, 0, vrs::pool_local_capacity(shape.local), slack_cap, xcc_map, 1000u + r, shape.sub_bits, par));
        CK(hipEventRecord(ev[4], st));
        if (getenv("POOL_LAB_ISOLATE")) {  // the local sort on a QUIET chip: whatever the second pass left dirty in the caches has been written back
            CK(hipStreamSynchronize(st));
            usleep(3000);
        }
        CK(hipEventRecord(ev[5], st));
        CK(vrs::launch_pool_local_sort(st, slack, in, n, msd, pool, shape, head, nullptr, 1, par));
        CK(hipEventRecord(ev[6], st));
        float again_us = 0;
        if (getenv("POOL_LAB_ISOLATE")) {  // ... and once more (the same buckets from the same regions to the same places), behind another quiet period
            CK(hipStreamSynchronize(st));
            usleep(3000);
            CK(hipEventRecord(ev[8], st));
            CK(vrs::launch_pool_local_sort(st, slack, in, n, msd, pool, shape, head, nullptr, 1, par));
            CK(hipEventRecord(ev[9], st));
            CK(hipStreamSynchronize(st));
            CK(hipEventElapsedTime(&again_us, ev[8], ev[9]));
            again_us *= 1e3f;
        }
        hipLaunchKernelGGL(check_sorted, dim3(2048), dim3(256), 0, st, in, n, chk);
        CK(hipStreamSynchronize(st));
        vrs::MsdPlan hm;
        vrs::PoolPlan hp;
        unsigned long long hc[4];
        CK(hipMemcpy(&hm, msd, 16, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&hp, pool, 32, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hc, chk, 32, hipMemcpyDeviceToHost));
        if (!hm.ok || getenv("POOL_LAB_DUMP")) {  // why: the fullest bucket against its room and the local sort's capacity
            const uint32_t buckets = 256u << shape.sub_bits;
            std::vector<uint32_t> start(buckets + 4), cur(buckets);
            CK(hipMemcpy(start.data(), reinterpret_cast<char *>(pool) + offsetof(vrs::PoolPlan, sub_start), start.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(cur.data(), reinterpret_cast<char *>(pool) + offsetof(vrs::PoolPlan, sub_cursor), cur.size() * 4, hipMemcpyDeviceToHost));
            uint32_t worst = 0, over = 0, mx = 0;
            double tight = 0;
            for (uint32_t b = 0; b < buckets; ++b) {
                const uint32_t room = start[b + 1] - start[b];
                mx = std::max(mx, cur[b]);
                if (cur[b] > room) { ++over; worst = b; }
                if (room) tight = std::max(tight, double(cur[b]) / room);
            }
            std::printf("   dump: slack used %u of %u, largest bucket %u (cap %u), buckets over their room %u (e.g. %u: %u > %u), tightest fill %.3f\n", start[buckets],
                        slack_cap, mx, vrs::pool_local_capacity(shape.local), over, worst, cur[worst], start[worst + 1] - start[worst], tight);
        }
        float t[6];
        for (int i = 0; i < 6; ++i) CK(hipEventElapsedTime(&t[i], ev[i], ev[i + 1]));
        std::printf("rep %d: sample %.1f  passA %.1f  plan %.1f  passB %.1f  (-) %.1f  local %.1f us   ok_a=%u fail=%u ok=%u shift=%u  descents=%llu sum %s\n", r, t[0] * 1e3,
                    t[1] * 1e3, t[2] * 1e3, t[3] * 1e3, t[4] * 1e3, t[5] * 1e3, hp.ok_a, hp.fail[r & 1], hm.ok, hm.shift, hc[0], hc[1] == hc[3] ? "same" : "DIFFERENT");
        if (again_us != 0) std::printf("        (local sort behind a quiet period: the figure above; once more behind another: %.1f us)\n", again_us);
        if (r >= 2) {
            for (int i = 0; i < 6; ++i) sum[i] += t[i] * 1e3;
            ++counted;
        }
    }
    std::printf("AVG n=%u: sample %.1f  passA %.1f  plan %.1f  passB %.1f  (-) %.1f  local %.1f us (events bracket each launch: +~2 us each); total %.1f\n", n,
                sum[0] / counted, sum[1] / counted, sum[2] / counted, sum[3] / counted, sum[4] / counted, sum[5] / counted,
                (sum[0] + sum[1] + sum[2] + sum[3] + sum[4] + sum[5]) / counted);
    return 0;
}
