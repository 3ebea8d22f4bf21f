// oracle_selftest.cpp -- the CPU oracle (oracle/vrs_oracle.c, vrs_stdsort.cpp: TEST INFRASTRUCTURE, never linked into the product) under
// AddressSanitizer + UndefinedBehaviorSanitizer: every entry point on ragged, empty and tile-edge sizes, against std::sort / std::stable_sort.
// Built by `make -C oracle selftest-asan`, run by tests/test_sanitizers_cpu.py; a checker with an out-of-bounds read would
// pin nothing.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

extern "C" {
uint32_t vrs_oracle_workgroup_count(uint32_t num_elements, uint32_t blocks_per_workgroup);
void vrs_oracle_multi_radixsort_pairs(uint32_t *kbuf0, uint32_t *kbuf1, uint32_t *vbuf0, uint32_t *vbuf1, uint32_t *hist, uint32_t num_elements,
                                      uint32_t blocks_per_workgroup, void (*stage_cb)(void *, uint32_t, uint32_t), void *user);
void vrs_oracle_multi_radixsort(uint32_t *buf0, uint32_t *buf1, uint32_t *hist, uint32_t num_elements, uint32_t blocks_per_workgroup);
void vrs_oracle_single_radixsort(uint32_t *buf0, uint32_t *buf1, uint32_t num_elements);
void vrs_oracle_mt19937_fill(uint32_t seed, uint32_t *out, uint64_t n, uint32_t top_bits_zeroed);
void vrs_oracle_multi_radixsort_u64(uint64_t *kbuf0, uint64_t *kbuf1, uint32_t *vbuf0, uint32_t *vbuf1, uint32_t *hist, uint32_t num_elements,
                                    uint32_t blocks_per_workgroup);
double vrs_stdsort_u32(uint32_t *data, uint64_t n);
int64_t vrs_test_sort(const uint32_t *reference, uint64_t n_reference, const uint32_t *out_buffer, uint64_t n_out);
double vrs_stable_sort_pairs_u32(uint32_t *keys, uint32_t *values, uint64_t n);
}

static int failures = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);  \
            ++failures;                                                    \
        }                                                                  \
    } while (0)

int main() {
    for (uint32_t n : {1u, 2u, 255u, 256u, 257u, 1000u, 4099u, 8192u, 20000u, 65537u})
        for (uint32_t B : {1u, 3u, 4u, 32u, 1000u}) {
            std::vector<uint32_t> keys(n);
            vrs_oracle_mt19937_fill(n + B, keys.data(), n, (n % 3 == 0) ? 4u : 0u);
            std::mt19937 gen(n + B);
            for (uint32_t i = 0; i < n; ++i) {
                uint32_t want = gen();
                if (n % 3 == 0) want >>= 4;
                if (keys[i] != want) {
                    CHECK(keys[i] == want);
                    break;
                }
            }
            if (n % 5 == 0)
                for (auto &k : keys) k &= 0x00FF00FFu;  // many ties
            const uint32_t W = vrs_oracle_workgroup_count(n, B);
            std::vector<uint32_t> b0(keys), b1(n), hist(static_cast<size_t>(W) * 256), ref(keys);
            vrs_oracle_multi_radixsort(b0.data(), b1.data(), hist.data(), n, B);
            vrs_stdsort_u32(ref.data(), n);
            CHECK(vrs_test_sort(ref.data(), n, b0.data(), n) == -1);
            // pairs: stable
            std::vector<uint32_t> k0(keys), k1(n), v0(n), v1(n), rk(keys), rv(n);
            std::iota(v0.begin(), v0.end(), 0u);
            std::iota(rv.begin(), rv.end(), 0u);
            vrs_oracle_multi_radixsort_pairs(k0.data(), k1.data(), v0.data(), v1.data(), hist.data(), n, B, nullptr, nullptr);
            vrs_stable_sort_pairs_u32(rk.data(), rv.data(), n);
            CHECK(k0 == rk && v0 == rv);
            if (n <= 4099) {
                std::vector<uint32_t> s0(keys), s1(n);
                vrs_oracle_single_radixsort(s0.data(), s1.data(), n);
                CHECK(s0 == ref);
            }
            // 64-bit keys
            std::vector<uint64_t> w0(n), w1(n);
            for (uint32_t i = 0; i < n; ++i) w0[i] = (static_cast<uint64_t>(keys[i]) << 32) | keys[n - 1 - i];
            std::vector<uint64_t> wr(w0);
            vrs_oracle_multi_radixsort_u64(w0.data(), w1.data(), nullptr, nullptr, hist.data(), n, B);
            std::sort(wr.begin(), wr.end());
            CHECK(w0 == wr);
        }
    std::printf(failures ? "oracle_selftest: %d FAILED\n" : "oracle_selftest: ok\n", failures);
    return failures ? 1 : 0;
}
