// Lab (round 2): where does a look-back scatter pass spend its time, and what does it wait for?
// Drives the one-call kernels directly (digit tables -> plan -> four look-back passes), with s_memtime phase marks in
// every tile, look-back statistics (polls, rows walked, round trips), and three comparisons on the same box:
// the contract scatter run back to back (what the previous pass's write drain costs), a look-back pass re-run over
// status rows that are already published (no waiting at all), and the counting read by group count.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I vkradixsort_amd/csrc tools/lab/lookback_lab.hip -o tools/lab/lookback_lab
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

__device__ unsigned long long *g_marks;     // [blocks][8]
__device__ unsigned long long g_lbstat[8];  // polls, rows, trips, tiles that polled, tiles
__shared__ unsigned long long s_marks[8];
#define VRS_MARK(i)                                                         \
    do {                                                                    \
        if (threadIdx.x == 0) s_marks[(i)] = __builtin_readcyclecounter();  \
    } while (0)
#define VRS_MARK_FLUSH()                                                                     \
    do {                                                                                     \
        if (threadIdx.x == 0 && g_marks) {                                                   \
            s_marks[6] = __builtin_readcyclecounter();                                       \
            for (int i_ = 0; i_ < 7; ++i_) g_marks[(size_t)blockIdx.x * 8 + i_] = s_marks[i_]; \
        }                                                                                    \
    } while (0)
// per-block slot (no contention): digit 17's thread packs polls | rows walked | round trips into mark 7
#define VRS_LB_STAT(polls, rows, trips)                                                                          \
    do {                                                                                                         \
        if ((threadIdx.x & 255u) == 17u && g_marks)                                                              \
            g_marks[(size_t)blockIdx.x * 8 + 7] =                                                                \
                ((unsigned long long)(polls) << 40) | ((unsigned long long)(rows) << 20) | (unsigned long long)(trips); \
    } while (0)
#include "../../tools/lab/vrs_all_kernels.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Ev {
    hipEvent_t a, b;
    Ev() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    vrs::LaunchEvents le() const { return vrs::LaunchEvents{a, b}; }
    float us() const { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms * 1e3f; }
};

int main(int argc, char **argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atof(argv[1]) : 100000000u;
    const uint32_t G = argc > 2 ? (uint32_t)atoi(argv[2]) : 32u;
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    std::vector<uint32_t> h(n);
    std::mt19937 gen(1);
    for (auto &x : h) x = gen();
    uint32_t *d_src, *d_a, *d_b;
    CK(hipMalloc(&d_src, (size_t)n * 4)); CK(hipMalloc(&d_a, (size_t)n * 4)); CK(hipMalloc(&d_b, (size_t)n * 4));
    CK(hipMemcpy(d_src, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int cus = 256; { hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); cus = p.multiProcessorCount; }
    // xcc map
    unsigned long long xcc_map = 0;
    {
        uint32_t *d; CK(hipMalloc(&d, 64 * 4)); CK(vrs::launch_xcc_probe(st, d, 64)); uint32_t hx[64];
        CK(hipMemcpyAsync(hx, d, sizeof hx, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        for (int b = 0; b < 8; ++b) xcc_map |= (unsigned long long)(hx[b] & 0xFF) << (8 * b);
        CK(hipFree(d));
    }
    const uint32_t S = vrs::kStreams, T = 8192;
    const uint32_t tiles_total = (n + T - 1) / T, group_tiles = (tiles_total + G - 1) / G, group_len = group_tiles * T;
    const vrs::StreamCuts cuts0 = vrs::pass0_stream_cuts(n, group_len, G);
    uint32_t tiles0 = 0;
    for (uint32_t k = 0; k < S; ++k) {
        const uint64_t a = std::min<uint64_t>((uint64_t)cuts0.first_group[k] * group_len, n), b = std::min<uint64_t>((uint64_t)cuts0.first_group[k + 1] * group_len, n);
        tiles0 = std::max<uint32_t>(tiles0, (uint32_t)((b - a + T - 1) / T));
    }
    const uint32_t even = (tiles_total + S - 1) / S, tile_cap = std::max(tiles0, even + even / 4 + 2);
    uint32_t *tables, *status; vrs::OnesweepPlan *plan; vrs::OnesweepPlanHead *host, *host_dev;
    CK(hipMalloc(&tables, vrs::kDigitTableWords * 4)); CK(hipMemset(tables, 0, vrs::kDigitTableWords * 4));
    CK(hipMalloc(&plan, sizeof(vrs::OnesweepPlan)));
    CK(hipHostMalloc((void **)&host, sizeof *host, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void **)&host_dev, host, 0));
    const size_t rows = (size_t)S * tile_cap;
    CK(hipMalloc(&status, rows * 1024));
    unsigned long long *d_marks; const size_t nblocks = (size_t)S * tile_cap;
    CK(hipMalloc(&d_marks, nblocks * 64));
    unsigned long long *null_marks = nullptr;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_marks), &null_marks, sizeof(null_marks)));
    Ev e_dt, e_p[4], e_x;
    uint32_t stamp = 0;
    printf("n=%u groups=%u tile_cap=%u tiles0=%u\n", n, G, tile_cap, tiles0);

    auto one_sort = [&](bool marks_pass, int which) {
        CK(hipMemcpyAsync(d_a, d_src, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        CK(vrs::launch_digit_tables(st, d_a, n, 4, 0, group_len, G, tables, status, rows * 256, cus, e_dt.le()));
        CK(vrs::launch_plan(st, tables, plan, host_dev, ++stamp, n, group_len, G, T, tile_cap, tile_cap, cuts0));
        uint32_t *in = d_a, *out = d_b;
        for (uint32_t i = 0; i < 4; ++i) {
            if (marks_pass) {
                unsigned long long *m = (int)i == which ? d_marks : nullptr;
                CK(hipStreamSynchronize(st));
                CK(hipMemcpyToSymbol(HIP_SYMBOL(g_marks), &m, sizeof(m)));
            }
            CK(vrs::launch_onesweep_scatter(st, in, out, nullptr, nullptr, plan, i, 8 * i, status, i == 0 ? tiles0 : tile_cap, false, true, xcc_map, 4, 4096, -1, e_p[i].le()));
            std::swap(in, out);
        }
        CK(hipStreamSynchronize(st));
    };
    // ---- A: steady state, per-kernel times
    for (int r = 0; r < 2; ++r) one_sort(false, 0);
    double dt = 0, ps[4] = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        one_sort(false, 0);
        dt += e_dt.us(); for (int i = 0; i < 4; ++i) ps[i] += e_p[i].us();
    }
    printf("A steady state: digit_tables %.1f us | passes %.1f %.1f %.1f %.1f us\n", dt / reps, ps[0] / reps, ps[1] / reps, ps[2] / reps, ps[3] / reps);
    {   // verify the last sort
        std::vector<uint32_t> out(n); CK(hipMemcpy(out.data(), d_a, (size_t)n * 4, hipMemcpyDeviceToHost));
        bool ok = true; for (size_t i = 1; i < n && ok; ++i) ok = out[i - 1] <= out[i];
        unsigned long long s0 = 0, s1 = 0; for (auto x : h) s0 += x; for (auto x : out) s1 += x;
        printf("  sorted=%d checksum=%d\n", (int)ok, (int)(s0 == s1));
    }
    // ---- B: phase marks of pass 0 and pass 2
    for (int which : {0, 2}) {
        CK(hipMemset(d_marks, 0, nblocks * 64));
        one_sort(true, which);
        CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> m(nblocks * 8);
        CK(hipMemcpy(m.data(), d_marks, m.size() * 8, hipMemcpyDeviceToHost));
        double sum[8] = {0}; size_t cnt = 0; unsigned long long tmin = ~0ull, tmax = 0;
        for (size_t w = 0; w < nblocks; ++w) {
            if (m[w * 8 + 6] == 0) continue;
            ++cnt;
            for (int i = 1; i <= 6; ++i) sum[i] += double(m[w * 8 + i] - m[w * 8 + i - 1]);
            tmin = std::min(tmin, m[w * 8]); tmax = std::max(tmax, m[w * 8 + 6]);
        }
        printf("B pass %d (%.1f us): %zu tiles, s_memtime ticks (10 ns): span %llu | load+zero %.0f | rank %.0f | scan+publish+fetch %.0f | rebucket+resolve %.0f | write-out issue %.0f | to end %.0f | tile lifetime %.0f\n",
               which, e_p[which].us(), cnt, tmax - tmin, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, sum[4] / cnt, sum[5] / cnt, sum[6] / cnt,
               (sum[1] + sum[2] + sum[3] + sum[4] + sum[5] + sum[6]) / cnt);
        // first-wave tiles vs the rest: lifetime by dispatch order
        double early = 0, late = 0; size_t ne = 0, nl = 0;
        for (size_t w = 0; w < nblocks; ++w) {
            if (m[w * 8 + 6] == 0) continue;
            const double life = double(m[w * 8 + 6] - m[w * 8]);
            if (w < 768) { early += life; ++ne; } else { late += life; ++nl; }
        }
        printf("   lifetime of the first 768 blocks %.0f ticks, of the rest %.0f ticks\n", ne ? early / ne : 0, nl ? late / nl : 0);
        {
            double polls[2] = {0, 0}, rowsw[2] = {0, 0}, trips[2] = {0, 0}; size_t c[2] = {0, 0}, polled[2] = {0, 0}; unsigned long long maxrows = 0;
            for (size_t w = 0; w < nblocks; ++w) {
                if (m[w * 8 + 6] == 0) continue;
                const unsigned long long x = m[w * 8 + 7];
                const int k = w < 768 ? 0 : 1;
                const unsigned long long pl = x >> 40, rw = (x >> 20) & 0xFFFFF, tr = x & 0xFFFFF;
                polls[k] += pl; rowsw[k] += rw; trips[k] += tr; ++c[k]; polled[k] += pl ? 1 : 0; maxrows = std::max(maxrows, rw);
            }
            for (int k = 0; k < 2; ++k)
                printf("   look-back of the %s: tiles %zu, polled %.1f%%, polls/tile %.2f, rows walked/tile %.2f, round trips/tile %.2f\n",
                       k ? "rest" : "first 768 blocks", c[k], c[k] ? 100.0 * polled[k] / c[k] : 0, c[k] ? polls[k] / c[k] : 0, c[k] ? rowsw[k] / c[k] : 0, c[k] ? trips[k] / c[k] : 0);
            printf("   longest walk %llu rows\n", maxrows);
        }
    }
    {
        unsigned long long *m = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_marks), &m, sizeof(m)));
    }
    // ---- C: the contract scatter, alone and back to back (what a scatter costs the kernel after it)
    {
        const uint32_t W = (n / 32 + 255) / 256 + ((n % 8192) ? 1 : 0);
        const uint32_t Wc = (uint32_t)(((uint64_t)n + 8191) / 8192);
        (void)W;
        uint32_t *hist; vrs::PrefixScratch sc;
        CK(hipMalloc(&hist, (size_t)Wc * 1024)); CK(hipMalloc(&sc.offsets, (size_t)Wc * 1024)); CK(hipMalloc(&sc.chunk_sums, (size_t)Wc * 1024));
        vrs::ScatterLaunch cfg; cfg.atomic_rank = true; cfg.compute_units = cus;
        CK(hipMemcpyAsync(d_a, d_src, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        CK(vrs::launch_histograms(st, d_a, hist, n, 0, Wc, 32));
        CK(vrs::launch_prefix(st, hist, sc, Wc));
        CK(hipStreamSynchronize(st));
        Ev ev[6];
        for (int r = 0; r < 3; ++r) {
            for (int i = 0; i < 6; ++i) CK(vrs::launch_scatter(st, d_a, d_b, nullptr, nullptr, sc.offsets, n, 0, Wc, 32, true, cfg, ev[i].le()));
            CK(hipStreamSynchronize(st));
        }
        printf("C contract scatter, six launches back to back (same pass): %.1f %.1f %.1f %.1f %.1f %.1f us\n", ev[0].us(), ev[1].us(), ev[2].us(), ev[3].us(), ev[4].us(), ev[5].us());
        // histogram after scatter, and a second histogram
        Ev eh[3];
        for (int r = 0; r < 2; ++r) {
            CK(vrs::launch_scatter(st, d_a, d_b, nullptr, nullptr, sc.offsets, n, 0, Wc, 32, true, cfg, ev[0].le()));
            for (int i = 0; i < 3; ++i) CK(vrs::launch_histograms(st, d_b, hist, n, 8, Wc, 32, eh[i].le()));
            CK(hipStreamSynchronize(st));
        }
        printf("C scatter %.1f us, then three histogram reads of its output: %.1f %.1f %.1f us\n", ev[0].us(), eh[0].us(), eh[1].us(), eh[2].us());
    }
    // ---- D: look-back pass 1 re-run over rows that are already published (no waiting), back to back
    {
        one_sort(false, 0);
        // after one_sort the status rows hold pass 3's tags; re-run pass 3 (input = buffer b after three swaps, output = a)
        Ev ev[4];
        for (int i = 0; i < 4; ++i)
            CK(vrs::launch_onesweep_scatter(st, d_b, d_a, nullptr, nullptr, plan, 3, 24, status, tile_cap, false, true, xcc_map, 4, 4096, -1, ev[i].le()));
        CK(hipStreamSynchronize(st));
        printf("D look-back pass 3 re-run four times over published rows (no waits): %.1f %.1f %.1f %.1f us (in the sort: %.1f)\n", ev[0].us(), ev[1].us(), ev[2].us(), ev[3].us(), e_p[3].us());
    }
    return 0;
}
