// Lab: does the ORDER in which tiles are visited (MALL / L2 residency across kernels) change the 4-pass sort time?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I vkradixsort_amd/csrc tools/lab/order_lab.hip -o tools/lab/order_lab
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>
#include "../../tools/lab/vrs_all_kernels.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

static std::vector<uint32_t> identity(uint32_t W) { std::vector<uint32_t> p(W); std::iota(p.begin(), p.end(), 0u); return p; }
static std::vector<uint32_t> reversed(std::vector<uint32_t> p) { std::reverse(p.begin(), p.end()); return p; }
// XCD-contiguous ranges (block b -> XCD b%8), ascending or descending inside the range
static std::vector<uint32_t> xcd_range(uint32_t W, bool descending) {
    std::vector<uint32_t> p(W);
    const uint32_t q = W / 8, r = W % 8;
    for (uint32_t b = 0; b < W; ++b) {
        const uint32_t x = b % 8, idx = b / 8;
        const uint32_t base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        const uint32_t cnt = x < r ? q + 1 : q;
        p[b] = base + (descending ? cnt - 1 - idx : idx);
    }
    return p;
}
// "most recently written first" for a buffer produced by a scatter whose XCDs each walked a contiguous tile range:
// every (digit, XCD) sub-segment was filled front to back, so visit tiles by descending position inside their sub-segment
static std::vector<uint32_t> recency(uint32_t W, uint32_t segments) {
    const double Lt = double(W) / segments;
    std::vector<uint32_t> p = identity(W);
    std::vector<double> key(W);
    for (uint32_t t = 0; t < W; ++t) { double f = (t + 0.5) / Lt; key[t] = f - floor(f); }
    std::stable_sort(p.begin(), p.end(), [&](uint32_t a, uint32_t b) { return key[a] > key[b]; });
    return p;
}

// groups of g consecutive tiles per XCD, super-groups of 8g tiles in launch order (what a decoupled look-back would need:
// tile t only depends on tiles dispatched at most 8g blocks earlier)
static std::vector<uint32_t> grouped(uint32_t W, uint32_t g) {
    std::vector<uint32_t> p(W);
    const uint32_t full = W / (8 * g) * (8 * g);
    for (uint32_t b = 0; b < W; ++b) {
        if (b >= full) { p[b] = b; continue; }
        const uint32_t sup = b / (8 * g), x = b % 8, within = (b / 8) % g;
        p[b] = sup * 8 * g + x * g + within;
    }
    return p;
}

struct Order { const char *name; std::vector<uint32_t> hist, scat; bool scat_default; };

int main(int argc, char **argv) {
    uint32_t n = argc > 1 ? (uint32_t)atof(argv[1]) : 100000000u;
    const uint32_t B = 32;
    const uint32_t gis = n / B + (n % B ? 1 : 0), W = (gis + 255) / 256;
    std::vector<uint32_t> h(n);
    std::mt19937 gen(1);
    for (auto &x : h) x = gen();
    std::vector<uint32_t> ref = h;
    std::sort(ref.begin(), ref.end());
    uint32_t *d_src, *d_a, *d_b, *d_hist, *d_oh, *d_os;
    CK(hipMalloc(&d_src, (size_t)n * 4)); CK(hipMalloc(&d_a, (size_t)n * 4)); CK(hipMalloc(&d_b, (size_t)n * 4));
    CK(hipMemcpy(d_src, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_hist, (size_t)W * 1024)); CK(hipMalloc(&d_oh, (size_t)W * 4)); CK(hipMalloc(&d_os, (size_t)W * 4));
    vrs::PrefixScratch sc;
    CK(hipMalloc(&sc.offsets, (size_t)W * 1024)); CK(hipMalloc(&sc.chunk_sums, (size_t)W * 1024));
    vrs::ScatterLaunch cfg; cfg.atomic_rank = true;

    std::vector<Order> orders;
    orders.push_back({"S0 hist asc            | scatter xcd-range asc (current)", identity(W), {}, true});
    orders.push_back({"S1 hist desc           | scatter xcd-range asc", reversed(identity(W)), {}, true});
    orders.push_back({"S2 hist recency(2048)  | scatter xcd-range asc", recency(W, 2048), {}, true});
    orders.push_back({"S3 hist recency(2048)  | scatter reverse(hist)", recency(W, 2048), reversed(recency(W, 2048)), false});
    orders.push_back({"S4 hist asc            | scatter desc (no xcd ranges)", identity(W), reversed(identity(W)), false});
    orders.push_back({"S5 hist xcd-range asc  | scatter xcd-range desc", xcd_range(W, false), xcd_range(W, true), false});
    orders.push_back({"S6 hist xcd-range desc | scatter xcd-range asc", xcd_range(W, true), xcd_range(W, false), false});
    orders.push_back({"S7 hist recency(256)   | scatter xcd-range asc", recency(W, 256), {}, true});
    orders.push_back({"S8 hist recency(2048)  | scatter xcd-range desc", recency(W, 2048), xcd_range(W, true), false});

    if (argc > 2) {
        orders.resize(1);
        orders.push_back({"G1  scatter launch order (round-robin XCDs)", identity(W), identity(W), false});
        static char names[8][64];
        int k = 0;
        for (uint32_t g : {2u, 4u, 8u, 16u, 32u, 64u, 256u}) {
            snprintf(names[k], 64, "G%-3u scatter groups of %u tiles per XCD", g, g);
            orders.push_back({names[k++], identity(W), grouped(W, g), false});
        }
    }
    hipEvent_t ev[4][3][2], t0, t1;
    for (auto &a : ev) for (auto &b : a) for (auto &c : b) CK(hipEventCreate(&c));
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (auto &o : orders) {
        CK(hipMemcpy(d_oh, o.hist.data(), (size_t)W * 4, hipMemcpyHostToDevice));
        if (!o.scat_default) CK(hipMemcpy(d_os, o.scat.data(), (size_t)W * 4, hipMemcpyHostToDevice));
        double best = 1e9, kh = 0, kp = 0, ks = 0;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemcpy(d_a, d_src, (size_t)n * 4, hipMemcpyDeviceToDevice));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, 0));
            uint32_t *in = d_a, *out = d_b;
            for (int p = 0; p < 4; ++p) {
                CK(vrs::launch_histograms(0, in, d_hist, n, 8 * p, W, B, {ev[p][0][0], ev[p][0][1]}, d_oh));
                CK(vrs::launch_prefix(0, d_hist, sc, W, {ev[p][1][0], ev[p][1][1]}));
                CK(vrs::launch_scatter(0, in, out, nullptr, nullptr, sc.offsets, n, 8 * p, W, B, true, cfg,
                                       {ev[p][2][0], ev[p][2][1]}, o.scat_default ? nullptr : d_os));
                std::swap(in, out);
            }
            CK(hipEventRecord(t1, 0));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, t0, t1));
            if (rep > 0 && ms < best) {
                best = ms; kh = kp = ks = 0;
                for (int p = 0; p < 4; ++p) {
                    float a, b, c;
                    CK(hipEventElapsedTime(&a, ev[p][0][0], ev[p][0][1])); CK(hipEventElapsedTime(&b, ev[p][1][0], ev[p][1][1]));
                    CK(hipEventElapsedTime(&c, ev[p][2][0], ev[p][2][1]));
                    kh += a; kp += b; ks += c;
                }
            }
        }
        std::vector<uint32_t> outv(n);
        CK(hipMemcpy(outv.data(), d_a, (size_t)n * 4, hipMemcpyDeviceToHost));
        bool ok = outv == ref;
        printf("%-58s total %.1f us | hist %.1f prefix %.1f scatter %.1f (avg/pass) exact=%d\n", o.name, best * 1e3, kh * 250, kp * 250,
               ks * 250, (int)ok);
    }
    return 0;
}
