// capi_host_sanity.cpp -- the host-only entry points of the C ABI, run under AddressSanitizer + UndefinedBehaviorSanitizer and under
// ThreadSanitizer (python -m vkradixsort_amd.build --asan / --tsan; tests/test_sanitizers_cpu.py).  No GPU is needed: what needs one must
// fail with a status code, never crash.  The reference's counterpart are the Vulkan validation layers it switches on whenever NDEBUG is not
// defined (engine/include/engine/core/GPUContext.h:84-90): a debug build that checks the host's use of the API.
//   arguments: [hub] = only the loopback hub's rendezvous (the part worth running under TSan)
#include "vkradixsort_amd.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <thread>
#include <vector>

static int failures = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);  \
            ++failures;                                                    \
        }                                                                  \
    } while (0)

static void launch_shapes() {
    // ComputePass.h:24-29: ceil(ceil(N / B) / 256) workgroups of 256 invocations
    for (uint64_t n : {0ull, 1ull, 255ull, 256ull, 257ull, 1000ull, 65536ull, 100000000ull, 4294967295ull})
        for (uint32_t b : {1u, 3u, 32u, 16384u}) {
            const uint32_t w = vrs_workgroup_count(static_cast<uint32_t>(n), b);
            const uint64_t per = 256ull * b;
            CHECK(w == (n + per - 1) / per);
            CHECK(vrs_global_invocation_size(static_cast<uint32_t>(n), b) == (n + b - 1) / b);
        }
    CHECK(std::strstr(vrs_version(), "vkradixsort_amd") != nullptr);
}

static void pool_shapes() {
    uint32_t a = 9, b = 9, cap = 9;
    uint64_t bytes = 9;
    for (uint32_t n : {0u, 1000u, (1u << 22) - 1u, 1u << 22, 10000000u, 100000000u, 130000000u, 224000000u, 224000001u, 4294967295u})
        for (int pairs : {0, 1})
            for (int top : {0, 6, 7, 8}) {
                CHECK(vrs_pool_form_shape_ex(n, pairs, top, &a, &b, &cap, &bytes) == VRS_OK);
                CHECK((a == 0) == (b == 0) && (a == 0) == (cap == 0) && (a == 0) == (bytes == 0));
                if (a) CHECK(a + b >= 14 && a + b <= 16 && bytes > 4ull * n);
            }
    CHECK(vrs_pool_form_shape_ex(100000000u, 0, 5, &a, &b, &cap, &bytes) != VRS_OK);
    CHECK(vrs_pool_form_shape(100000000u, nullptr, nullptr, nullptr) == VRS_OK);
    CHECK(vrs_pool_form_shape(100000000u, &a, &cap, &bytes) == VRS_OK && a == 6 && cap == 7165);
}

static void splitters() {
    std::mt19937 gen(7);
    for (int round = 0; round < 200; ++round) {
        uint64_t counts[256];
        const int kind = round % 4;
        for (int d = 0; d < 256; ++d) counts[d] = kind == 0 ? gen() % 1000 : kind == 1 ? (d == 17 ? 1000000 : 0) : kind == 2 ? 0 : (d % 7 == 0 ? gen() : 0);
        for (int parts = 1; parts <= 32; parts += 1 + round % 5) {
            uint32_t bounds[40];
            std::fill(bounds, bounds + 40, 0xDEADu);
            CHECK(vrs_dist_plan_splitters(counts, parts, bounds) == VRS_OK);
            CHECK(bounds[0] == 0 && bounds[parts] == 256 && bounds[parts + 1] == 0xDEADu);
            for (int q = 0; q < parts; ++q) CHECK(bounds[q] <= bounds[q + 1]);
        }
    }
    uint32_t bounds[4];
    CHECK(vrs_dist_plan_splitters(nullptr, 2, bounds) != VRS_OK);
    uint64_t counts[256] = {1};
    CHECK(vrs_dist_plan_splitters(counts, 0, bounds) != VRS_OK);
    // sampled splitters: the weighted quantiles of the ranks' samples
    for (int world : {1, 2, 3, 8}) {
        const uint32_t per_rank = 64;
        std::vector<uint32_t> samples(static_cast<size_t>(world) * per_rank);
        std::vector<uint64_t> sizes(static_cast<size_t>(world));
        for (int q = 0; q < world; ++q) {
            sizes[static_cast<size_t>(q)] = 1000 + gen() % 100000;
            for (uint32_t i = 0; i < per_rank; ++i) samples[static_cast<size_t>(q) * per_rank + i] = gen();
            std::sort(samples.begin() + q * per_rank, samples.begin() + (q + 1) * per_rank);
        }
        for (int parts : {1, 2, 5, 32}) {
            std::vector<uint32_t> cut(static_cast<size_t>(parts) + 1, 0xDEADu);
            CHECK(vrs_dist_plan_sampled_splitters(samples.data(), sizes.data(), world, per_rank, parts, cut.data()) == VRS_OK);
            for (int p = 1; p + 1 < parts; ++p) CHECK(cut[static_cast<size_t>(p) - 1] <= cut[static_cast<size_t>(p)]);
            CHECK(cut[static_cast<size_t>(parts) - 1] == 0xDEADu && cut[static_cast<size_t>(parts)] == 0xDEADu);  // parts - 1 cut keys, nothing behind them
        }
    }
}

static void errors_without_a_device() {
    // a NULL context / buffer is an argument error everywhere, a missing device a status code
    CHECK(vrs_sort_keys_u32(nullptr, nullptr, nullptr, 10) != VRS_OK);
    CHECK(vrs_sort_pairs_u32(nullptr, nullptr, nullptr, nullptr, nullptr, 10) != VRS_OK);
    CHECK(vrs_queue_wait_idle(nullptr) != VRS_OK);
    CHECK(vrs_buffer_release(nullptr) == VRS_OK);
    CHECK(vrs_context_destroy(nullptr) == VRS_OK);
    CHECK(vrs_set_tuning(nullptr, VRS_TUNE_MSD_POOL, 1) != VRS_OK);
    CHECK(vrs_context_trim_scratch(nullptr, nullptr) != VRS_OK);
    CHECK(vrs_last_error(nullptr) != nullptr);
    CHECK(vrs_dist_last_error(nullptr) != nullptr);
    int count = -1;
    const int rc = vrs_device_count(&count);
    vrs_context ctx = nullptr;
    const int made = vrs_context_create(0, &ctx);
    if (rc != VRS_OK || count == 0) {
        CHECK(made != VRS_OK && ctx == nullptr);
        CHECK(std::strlen(vrs_last_error(nullptr)) > 0);
    } else if (made == VRS_OK) {
        CHECK(vrs_context_destroy(ctx) == VRS_OK);
    }
    vrs_dist_loopback hub = nullptr;
    CHECK(vrs_dist_loopback_create(0, &hub) != VRS_OK && hub == nullptr);
    CHECK(vrs_dist_loopback_create(3, nullptr) != VRS_OK);
    CHECK(vrs_dist_loopback_destroy(nullptr) == VRS_OK);
}

// ---- the loopback hub over host memory: `world` threads, every collective the multi-GPU step uses, then a rank that breaks the rules
static void hub_rendezvous(int world, int rounds) {
    vrs_dist_loopback hub = nullptr;
    CHECK(vrs_dist_loopback_create_host(world, &hub) == VRS_OK);
    if (!hub) return;
    std::atomic<int> bad{0};
    const auto rank_main = [&](int rank) {
        vrs_dist_transport t{};
        if (vrs_dist_loopback_transport(hub, rank, &t) != VRS_OK) {
            ++bad;
            return;
        }
        std::mt19937 gen(1000u + static_cast<uint32_t>(rank));
        for (int r = 0; r < rounds; ++r) {
            const size_t words = 1 + static_cast<size_t>(r % 7) * 5;
            // all-gather: rank q offers q * 1000 + r + i
            std::vector<uint32_t> mine(words), all(words * static_cast<size_t>(world), 0xFFFFFFFFu);
            for (size_t i = 0; i < words; ++i) mine[i] = static_cast<uint32_t>(rank * 1000 + r + static_cast<int>(i));
            if (t.all_gather(t.user, mine.data(), all.data(), words, nullptr) != 0) ++bad;
            for (int q = 0; q < world; ++q)
                for (size_t i = 0; i < words; ++i)
                    if (all[static_cast<size_t>(q) * words + i] != static_cast<uint32_t>(q * 1000 + r + static_cast<int>(i))) ++bad;
            // all-reduce (sum), in place
            std::vector<uint32_t> acc(words);
            for (size_t i = 0; i < words; ++i) acc[i] = static_cast<uint32_t>(rank + 1) * static_cast<uint32_t>(i + 1);
            if (t.all_reduce(t.user, acc.data(), acc.data(), words, nullptr) != 0) ++bad;
            for (size_t i = 0; i < words; ++i)
                if (acc[i] != static_cast<uint32_t>(world * (world + 1) / 2) * static_cast<uint32_t>(i + 1)) ++bad;
            // grouped send / recv: rank q sends (q + p + r) % 5 + 1 words to every peer p with (q + p + r) % 3 != 0, two messages to its right neighbour
            const auto msg_words = [&](int from, int to) { return static_cast<size_t>((from + to + r) % 5 + 1); };
            const auto sends_to = [&](int from, int to) { return from != to && (from + to + r) % 3 != 0; };
            std::vector<std::vector<uint32_t>> out(static_cast<size_t>(world)), in(static_cast<size_t>(world)), out2(1), in2(1);
            if (t.group_start(t.user) != 0) ++bad;
            for (int p = 0; p < world; ++p) {
                if (sends_to(rank, p)) {
                    out[static_cast<size_t>(p)].assign(msg_words(rank, p), static_cast<uint32_t>(rank * 100 + p));
                    if (t.send(t.user, out[static_cast<size_t>(p)].data(), out[static_cast<size_t>(p)].size(), p, nullptr) != 0) ++bad;
                }
                if (sends_to(p, rank)) {
                    in[static_cast<size_t>(p)].assign(msg_words(p, rank), 0u);
                    if (t.recv(t.user, in[static_cast<size_t>(p)].data(), in[static_cast<size_t>(p)].size(), p, nullptr) != 0) ++bad;
                }
            }
            if (world > 1) {  // a second message on one pair: matched in posting order
                const int right = (rank + 1) % world, left = (rank + world - 1) % world;
                out2[0].assign(3, static_cast<uint32_t>(7000 + rank));
                in2[0].assign(3, 0u);
                if (t.send(t.user, out2[0].data(), 3, right, nullptr) != 0) ++bad;
                if (t.recv(t.user, in2[0].data(), 3, left, nullptr) != 0) ++bad;
            }
            if (t.group_end(t.user) != 0) ++bad;
            for (int p = 0; p < world; ++p)
                if (sends_to(p, rank))
                    for (uint32_t v : in[static_cast<size_t>(p)])
                        if (v != static_cast<uint32_t>(p * 100 + rank)) ++bad;
            if (world > 1)
                for (uint32_t v : in2[0])
                    if (v != static_cast<uint32_t>(7000 + (rank + world - 1) % world)) ++bad;
        }
        // a collective whose sizes differ between the ranks: every rank must come back with an error, nobody may hang
        std::vector<uint32_t> mine(8, 1u), all(8u * static_cast<size_t>(world) + 8u);
        const int rc = t.all_gather(t.user, mine.data(), all.data(), rank == 0 && world > 1 ? 7 : 8, nullptr);
        if (world > 1 && rc == 0) ++bad;
        if (world > 1 && t.error_string(t.user, rc) == nullptr) ++bad;
    };
    std::vector<std::thread> threads;
    for (int q = 0; q < world; ++q) threads.emplace_back(rank_main, q);
    for (auto &th : threads) th.join();
    CHECK(bad.load() == 0);
    CHECK(vrs_dist_loopback_destroy(hub) == VRS_OK);
}

int main(int argc, char **argv) {
    const bool hub_only = argc > 1 && std::strcmp(argv[1], "hub") == 0;
    if (!hub_only) {
        launch_shapes();
        pool_shapes();
        splitters();
        errors_without_a_device();
    }
    for (int world : {1, 2, 3, 8}) hub_rendezvous(world, hub_only ? 40 : 12);
    std::printf(failures ? "capi_host_sanity: %d FAILED\n" : "capi_host_sanity: ok\n", failures);
    return failures ? 1 : 0;
}
