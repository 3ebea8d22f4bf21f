// distsortexample [ranks] [keys per rank] [rounds] [seed] [key bits]
// (key bits < 32: keys >> (32 - bits) -- small keys, whose top bytes no byte-aligned cut can balance: the step cuts at sampled keys)
// The multi-GPU step (vrs_dist_sort_keys_u32, BASELINE.json configs[4]) driven from a C++ host: one std::thread per rank, every
// rank its own context on device (rank % device count), the wire the library's in-process transport (vrs_dist_loopback_*) --
// device-to-device copies ordered by events.  With one GPU all ranks share it (how the rank-to-rank bookkeeping is tested here);
// with several GPUs in one process this IS a multi-GPU sort without RCCL.  The reference has no counterpart (single GPU); the
// verification is the reference's own: the concatenation of the ranks' outputs must equal std::sort of all keys
// (MultiRadixSort.cpp:141-161), printed in its words.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "vkradixsort_amd.h"

namespace {
void check(int rc, vrs_context ctx, const char *what) {
    if (rc != VRS_OK) throw std::runtime_error(std::string(what) + ": " + vrs_last_error(ctx));
}
}  // namespace

int main(int argc, char **argv) {
    const int world = argc > 1 ? std::atoi(argv[1]) : 2;
    const uint32_t n = argc > 2 ? static_cast<uint32_t>(std::strtod(argv[2], nullptr)) : 2000000u;
    const int rounds = argc > 3 ? std::atoi(argv[3]) : 2;
    const uint32_t seed = argc > 4 ? static_cast<uint32_t>(std::atoi(argv[4])) : 1000u;
    const int key_bits = argc > 5 ? std::min(32, std::max(1, std::atoi(argv[5]))) : 32;
    try {
        int devices = 0;
        if (vrs_device_count(&devices) != VRS_OK || devices == 0) throw std::runtime_error(std::string("no device: ") + vrs_last_error(nullptr));
        if (world < 1 || world > 32) throw std::runtime_error("ranks must be 1 .. 32");
        std::cout << "[DistSort] Sorting " << world << " x " << n << " 32bit numbers on " << std::min(world, devices) << " device(s), " << rounds
                  << " exchange round(s)." << std::endl;
        std::vector<std::vector<uint32_t>> shards(static_cast<size_t>(world)), outs(static_cast<size_t>(world));
        for (int r = 0; r < world; ++r) {  // shard g uses seed + g
            std::mt19937 gen(seed + static_cast<uint32_t>(r));
            shards[static_cast<size_t>(r)].resize(n);
            for (auto &k : shards[static_cast<size_t>(r)]) k = static_cast<uint32_t>(gen()) >> (32 - key_bits);
        }
        vrs_dist_loopback hub = nullptr;
        if (vrs_dist_loopback_create(world, &hub) != VRS_OK) throw std::runtime_error(vrs_dist_last_error(nullptr));
        std::vector<std::string> errors(static_cast<size_t>(world));
        const auto rank_main = [&](int r) {
            vrs_context ctx = nullptr;
            vrs_dist d = nullptr;
            vrs_buffer keys = nullptr;
            try {
                check(vrs_context_create(r % devices, &ctx), nullptr, "vrs_context_create");
                vrs_dist_transport wire;
                if (vrs_dist_loopback_transport(hub, r, &wire) != VRS_OK) throw std::runtime_error(vrs_dist_last_error(nullptr));
                const uint32_t capacity = static_cast<uint32_t>(n * 1.25) + 70000u;
                if (vrs_dist_create_with_transport(ctx, &wire, r, world, capacity, rounds, &d) != VRS_OK)
                    throw std::runtime_error(vrs_dist_last_error(nullptr));
                check(vrs_buffer_create(ctx, static_cast<size_t>(n) * 4, &keys), ctx, "vrs_buffer_create");
                check(vrs_buffer_upload(ctx, keys, shards[static_cast<size_t>(r)].data(), static_cast<size_t>(n) * 4), ctx, "vrs_buffer_upload");
                vrs_buffer range = nullptr;
                uint32_t count = 0;
                if (vrs_dist_sort_keys_u32(d, keys, n, &range, &count) != VRS_OK) throw std::runtime_error(vrs_dist_last_error(d));
                check(vrs_queue_wait_idle(ctx), ctx, "vrs_queue_wait_idle");
                outs[static_cast<size_t>(r)].resize(count);
                if (count) check(vrs_buffer_download(ctx, range, outs[static_cast<size_t>(r)].data(), static_cast<size_t>(count) * 4), ctx, "vrs_buffer_download");
            } catch (const std::exception &e) {
                errors[static_cast<size_t>(r)] = e.what();
            }
            if (keys) vrs_buffer_release(keys);
            if (d) vrs_dist_destroy(d);
            if (ctx) vrs_context_destroy(ctx);
        };
        std::vector<std::thread> threads;
        for (int r = 0; r < world; ++r) threads.emplace_back(rank_main, r);
        for (auto &t : threads) t.join();
        vrs_dist_loopback_destroy(hub);
        for (int r = 0; r < world; ++r)
            if (!errors[static_cast<size_t>(r)].empty()) throw std::runtime_error("rank " + std::to_string(r) + ": " + errors[static_cast<size_t>(r)]);
        std::vector<uint32_t> all, got;
        for (int r = 0; r < world; ++r) {
            all.insert(all.end(), shards[static_cast<size_t>(r)].begin(), shards[static_cast<size_t>(r)].end());
            got.insert(got.end(), outs[static_cast<size_t>(r)].begin(), outs[static_cast<size_t>(r)].end());
            std::cout << "[DistSort] rank " << r << " holds " << outs[static_cast<size_t>(r)].size() << " keys." << std::endl;
        }
        std::sort(all.begin(), all.end());
        if (got.size() != all.size()) throw std::runtime_error("TEST FAILED.");
        for (size_t i = 0; i < all.size(); ++i)
            if (got[i] != all[i]) throw std::runtime_error("TEST FAILED.");
        std::cout << "[DistSort] Test passed." << std::endl;
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
