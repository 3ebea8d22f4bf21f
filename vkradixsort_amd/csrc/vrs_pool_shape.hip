// vrs_pool_shape.hip -- host-only arithmetic of the pool form (vrs_msd_pool.hip): how an input is cut into slices and tiles, how much
// overflow room and slack its regions may need, which local-sort shape takes the fullest uniform bucket, how the buckets' bits are cut
// between the two passes.  No kernel here: vrs_pool_form_shape(_ex) answers from these without a device.
#include "vrs_local_sort.hpp"

#include <algorithm>
#include <cmath>

namespace vrs {

PoolStreams pool_streams(uint32_t n) {
    PoolStreams ps{};
    ps.tiles_total = (n + kPoolTile - 1u) / kPoolTile;
    ps.tiles_per_stream = std::max<uint32_t>((ps.tiles_total + 7u) / 8u, 1u);
    for (uint32_t s = 0; s < 8u; ++s) {
        const uint64_t a = std::min<uint64_t>(static_cast<uint64_t>(s) * ps.tiles_per_stream * kPoolTile, n);
        const uint64_t b = std::min<uint64_t>(static_cast<uint64_t>(s + 1u) * ps.tiles_per_stream * kPoolTile, n);
        ps.start[s] = static_cast<uint32_t>(a);
        ps.len[s] = static_cast<uint32_t>(b - a);
        const uint32_t full = ps.len[s] / kPoolTile, rest = ps.len[s] % kPoolTile;
        ps.sampled[s] = full * kPoolSampleKeys + std::min(rest, kPoolSampleKeys);
    }
    return ps;
}

uint32_t pool_overflow_capacity(uint32_t n) {
    // sum over the 2048 regions of [six deviations of an estimate scaled up 32-fold + rounding + floor], bounded by
    // Cauchy-Schwarz: sum sqrt(r e_i) <= sqrt(2048 r n); r is 32 but for the slices' ragged last tiles
    const double room = 6.0 * std::sqrt(2048.0 * 33.0 * (static_cast<double>(n) + 2048.0 * 33.0)) + 2048.0 * (kPoolRoomFloor + 64.0);
    return static_cast<uint32_t>(std::min<double>(room, 1u << 28)) & ~31u;
}

uint32_t pool_slack_capacity(uint32_t n, uint32_t sub_bits, uint32_t top_bytes) {
    // the plan kernel gives a top byte of c keys c + 6 sqrt(S R (c + S R)) + S (floor + 4) slots (pool_space, S = 2^sub_bits); over T
    // top bytes with sum c = n that is at most n + 6 sqrt(T S R (n + T S R)) + T S (floor + 4) + rounding (Cauchy-Schwarz), + the dump tile
    const double R = 40.0, B = static_cast<double>(top_bytes) * static_cast<double>(1u << sub_bits);
    const double room = 6.0 * std::sqrt(B * R * (static_cast<double>(n) + B * R)) + B * (kPoolRoomFloor + 4.0) + 256.0 * 4.0;
    return (static_cast<uint32_t>(std::min<double>(static_cast<double>(n) + room, 3.9e9) + 31.0) & ~31u) + kPoolTile;
}

uint32_t pool_tiles_b_cap(uint32_t n) {
    const uint32_t tiles = (n + kPoolTile - 1u) / kPoolTile, even = (tiles + 7u) / 8u;
    // an XCD walks 32 top bytes, each rounded up to whole tiles; the grid is sized before the plan is known (a quarter more than
    // an even split: skewed top bytes)
    return even + even / 4u + 32u + 8u;
}

uint32_t pool_local_capacity(uint32_t local) {
    if (local == 4u || local == 5u) return (local == 4u ? 512u : 1024u) * kPoolPairItems;  // pairs: 6656 / 13312
    if (local == 3u) return 64u * 4u * kLeanMaxVec - 3u;  // one wave per bucket: 1789
    return (local == 2u ? 512u : 256u) * 4u * (local == 0u ? 4u : static_cast<uint32_t>(kLeanMaxVec)) - 3u;
}

PoolShape pool_shape(uint32_t n, int forced_sub_bits) {
    // the fullest of the uniform buckets: 4 to 4.5 deviations above the mean -- 5.5 and a little here
    const auto fits = [&](uint32_t sub_bits, uint32_t local) {
        const double mean = static_cast<double>(n) / (256u << sub_bits);
        return static_cast<uint64_t>(mean + 5.5 * std::sqrt(mean)) + 32u <= pool_local_capacity(local);
    };
    PoolShape sh{};
    // Six bits as long as the 16384 buckets fit a 256-thread workgroup; beyond (about 1.1e8 uniform keys) seven bits keep them there
    // (2e8 keys: 1.05 instead of 1.08 ms with the 512-thread shape).  Smaller buckets are NOT better: a workgroup's fixed work -- five
    // counter tables to zero and scan, two memory round trips -- is a third of its life at 3000 keys (10^8 keys by seven bits: the
    // local sort 205 instead of 176 us, with five workgroups per CU), a sixth at 6100.
    sh.sub_bits = forced_sub_bits >= 6 && forced_sub_bits <= 8 ? static_cast<uint32_t>(forced_sub_bits) : (fits(6, 1) ? 6u : 7u);
    sh.local = fits(sh.sub_bits, 3) ? 3u : fits(sh.sub_bits, 0) ? 0u : fits(sh.sub_bits, 1) ? 1u : 2u;
    return sh;
}

PoolShape pool_shape_pairs(uint32_t n) {
    // pairs: the local sort's shape by the fullest uniform bucket (pool_shape's rule); local 4 = 512 threads, 5 = 1024.  Six bits while
    // the 16384 buckets fit the 512-thread workgroup (about 1.05e8 pairs); beyond, seven bits keep them there (32768 buckets: 2e8 pairs
    // 2.35 -> 2.0 ms); the 1024-thread workgroup only where even those do not fit
    const auto fits = [&](uint32_t sub_bits, uint32_t local) {
        const double mean = static_cast<double>(n) / (256u << sub_bits);
        return static_cast<uint64_t>(mean + 5.5 * std::sqrt(mean)) + 32u <= pool_local_capacity(local);
    };
    if (fits(6, 4)) return PoolShape{6u, 4u};
    if (fits(7, 4)) return PoolShape{7u, 4u};
    return PoolShape{6u, 5u};
}

PoolCut pool_cut(uint32_t n, bool pairs, int top_bits_setting, int forced_sub_bits) {
    // the shape by size in the 8 + S naming (256 << S buckets), then the cut of those bits between the passes: the setting (default 7 + 7)
    // only where the usual cut would be 8 + 6 -- sorts whose second pass takes 7 or 8 bits of 256 top bytes keep that cut
    PoolShape shape = pairs ? pool_shape_pairs(n) : pool_shape(n, forced_sub_bits);
    PoolCut cut{};
    cut.top_bits = (top_bits_setting != 8 && shape.sub_bits == 6u) ? static_cast<uint32_t>(top_bits_setting) : 8u;
    cut.sub_bits = shape.sub_bits + 8u - cut.top_bits;
    cut.local = shape.local;
    return cut;
}

uint32_t pool_max_pairs() {
    // the largest n whose fullest uniform bucket (pool_shape_pairs' rule) fits the shape that function returns: beyond it the second pass
    // flags every sort and no larger pairs shape exists -- such sorts must not be candidates at all (found by bisection, once)
    static const uint32_t limit = [] {
        const auto fits = [](uint32_t n) {
            const PoolShape sh = pool_shape_pairs(n);
            const double mean = static_cast<double>(n) / (256u << sh.sub_bits);
            return static_cast<uint64_t>(mean + 5.5 * std::sqrt(mean)) + 32u <= pool_local_capacity(sh.local);
        };
        uint32_t lo = 1u << 22, hi = 300000000u;  // fits(lo), !fits(hi)
        while (hi - lo > 1u) {
            const uint32_t mid = lo + (hi - lo) / 2u;
            (fits(mid) ? lo : hi) = mid;
        }
        return lo;
    }();
    return limit;
}

PoolShape pool_grouped_shape(uint32_t n, uint32_t top_bytes) {
    // S bits below the top byte for the second pass, 24 - S <= 18 for the local sort, top_bytes << S buckets within the plan's tables:
    // the smallest S whose (uniform) buckets fit a 256-thread local sort, else the largest that fits at all; sub_bits 0 = none does
    PoolShape best{0u, 0u};
    for (uint32_t s = 6u; s <= 8u; ++s) {
        if ((top_bytes << s) > kPoolMaxBuckets) break;
        const double mean = static_cast<double>(n) / (static_cast<double>(top_bytes) * (1u << s));
        const uint64_t need = static_cast<uint64_t>(mean + 5.5 * std::sqrt(mean)) + 32u;
        for (uint32_t local : {3u, 0u, 1u, 2u}) {
            if (need > pool_local_capacity(local)) continue;
            if (best.sub_bits == 0u || (best.local == 2u && local != 2u)) best = PoolShape{s, local};
            break;
        }
        if (best.sub_bits != 0u && best.local != 2u) break;
    }
    return best;
}

}  // namespace vrs
