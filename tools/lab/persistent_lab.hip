// Lab: one-workgroup-per-tile scatter vs persistent workgroups (grid = CUs x k).
// Result (MI355X, N = 1e8): per-tile 155 us; persistent 212-242 us -- a persistent workgroup has to cross a barrier
// (vmcnt(0): store acknowledgements) between tiles, a finished per-tile workgroup just exits while its stores drain.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../tools/lab/vrs_all_kernels.hip"

namespace vrs {
// ---------------------------------------------------------------------------------------------
// Experiment (tools/lab): the same scatter with PERSISTENT workgroups -- a fixed grid, each workgroup walks its
// XCD range's tiles round-robin -- to see whether 12 208 short-lived workgroups cost launch/tail time.
template <typename K, int ITEMS, int WAVES, int RANK, int OCC>
__global__ __launch_bounds__(WAVES * 64, OCC) void scatter_persistent_lab_kernel(const K *__restrict__ keys_in,
                                                                                 K *__restrict__ keys_out,
                                                                                 const uint32_t *__restrict__ offsets,
                                                                                 uint32_t n, uint32_t shift, uint32_t W) {
    __shared__ ChunkSmem<K, ITEMS, WAVES, false> sm;
    constexpr uint32_t kChunk = ITEMS * WAVES * 64;
    const RadixDigit<K> dg{shift};
    const uint32_t q = W >> 3, r = W & 7u, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t base = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    const uint32_t last = base + (xcd < r ? q + 1u : q);
    const uint32_t stride = gridDim.x >> 3;
    for (uint32_t w = base + slot; w < last; w += stride) {
        const uint64_t tile_begin = static_cast<uint64_t>(w) * kChunk;
        if (tile_begin >= n) break;
        const uint32_t valid = static_cast<uint32_t>(tile_begin + kChunk <= n ? kChunk : n - tile_begin);
        uint32_t run_off = threadIdx.x < kBins ? offsets[static_cast<size_t>(w) * kBins + threadIdx.x] : 0u;
        if (valid == kChunk)
            scatter_chunk<K, ITEMS, WAVES, false, RANK, true>(sm, keys_in + tile_begin, nullptr, keys_out, nullptr, valid, dg, run_off);
        else
            scatter_chunk<K, ITEMS, WAVES, false, RANK, false>(sm, keys_in + tile_begin, nullptr, keys_out, nullptr, valid, dg, run_off);
        __syncthreads();  // LDS is reused by the next tile
    }
}

}  // namespace vrs
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
int main() {
    const uint32_t n = 100000000u, B = 32, W = 12208;
    std::vector<uint32_t> h(n);
    std::mt19937 gen(1);
    for (auto &x : h) x = gen();
    uint32_t *d_a, *d_b, *d_c, *d_hist;
    CK(hipMalloc(&d_a, (size_t)n * 4)); CK(hipMalloc(&d_b, (size_t)n * 4)); CK(hipMalloc(&d_c, (size_t)n * 4));
    CK(hipMemcpy(d_a, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_hist, (size_t)W * 1024));
    vrs::PrefixScratch sc;
    CK(hipMalloc(&sc.offsets, (size_t)W * 1024)); CK(hipMalloc(&sc.chunk_sums, (size_t)W * 1024));
    vrs::ScatterLaunch cfg; cfg.atomic_rank = true;
    CK(vrs::launch_histograms(0, d_a, d_hist, n, 0, W, B));
    CK(vrs::launch_prefix(0, d_hist, sc, W));
    CK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](const char *name, auto launch) {
        float best = 1e9;
        for (int r = 0; r < 6; ++r) {
            CK(hipEventRecord(a, 0)); launch(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r > 0 && ms < best) best = ms;
        }
        printf("%-40s %.1f us\n", name, best * 1e3);
    };
    timeit("one workgroup per tile (product)", [&] { CK(vrs::launch_scatter(0, d_a, d_b, nullptr, nullptr, sc.offsets, n, 0, W, B, true, cfg)); });
    CK(hipMemcpy(h.data(), d_b, 4096, hipMemcpyDeviceToHost));
    for (int per_cu : {1, 2, 3, 4}) {
        char nm[64]; snprintf(nm, 64, "persistent, %d workgroups per CU", per_cu);
        timeit(nm, [&] { hipLaunchKernelGGL((vrs::scatter_persistent_lab_kernel<uint32_t, 16, 8, 1, 4>), dim3(256 * per_cu), dim3(512), 0, 0,
                                            (const uint32_t *)d_a, d_c, (const uint32_t *)sc.offsets, n, 0u, W); });
    }
    std::vector<uint32_t> x(n), y(n);
    CK(hipMemcpy(x.data(), d_b, (size_t)n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), d_c, (size_t)n * 4, hipMemcpyDeviceToHost));
    printf("outputs equal: %d\n", (int)(x == y));
    return 0;
}
