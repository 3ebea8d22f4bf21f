// vrs_dist.hip -- the multi-GPU step behind the C ABI: key-range sharded sort over an RCCL communicator, one process
// per GPU (BASELINE.json configs[4]).  The reference has no multi-GPU code (no collective call
// site anywhere under /root/reference); north_star defines the path: shard by key range across the GPUs of a node,
// one all-to-all over xGMI between the local step and the local sorts.
//
// Host orchestration only -- every device step goes through the public C ABI of this library (the same entry points
// a C++ host would call): a top-byte partition pass (vrs_multi_radixsort_histograms + vrs_multi_radixsort with
// g_shift = 24), ONE all-gather of every rank's 256 top-byte counts, splitters, R rounds of grouped send/recv, and a
// vrs_sort_keys_u32 per received sub-range while the next round is on the wire.  The Python twin is
// vkradixsort_amd/distributed.py (RangeShardedSort.step, main path); the two plan_splitters must agree bit for bit
// (tests/test_capi_cpu.py).
//
// RCCL is bound at run time (dlopen / dlsym): the library has no link-time dependency on it, and a process that
// already carries an RCCL (PyTorch ships its own librccl.so) keeps using THAT one -- a communicator is only valid
// inside the copy of the library that made it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "vkradixsort_amd.h"

namespace {

// the slice of the RCCL API the step needs (rccl.h: ncclResult_t == int, ncclSuccess == 0, ncclUint32 == 3)
struct Rccl {
    void *handle = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok() const { return AllGather && Send && Recv && GroupStart && GroupEnd; }
};
constexpr int kNcclUint32 = 3;

thread_local std::string g_dist_error;

bool load_rccl(Rccl &r) {
    // an RCCL already mapped into the process first (RTLD_NOLOAD), then the system one
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (int pass = 0; pass < 2 && !r.handle; ++pass)
        for (const char *nm : names) {
            r.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (r.handle) break;
        }
    void *src = r.handle ? r.handle : RTLD_DEFAULT;  // RTLD_DEFAULT: symbols of whatever copy the process already holds
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(src, "ncclAllGather"));
    r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(src, "ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(src, "ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(src, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(src, "ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(src, "ncclGetErrorString"));
    return r.ok();
}

}  // namespace

struct vrs_dist_t {
    vrs_context ctx = nullptr;
    void *comm = nullptr;  // ncclComm_t; NULL only at world size 1 (no exchange partner: every transfer is a device copy)
    int rank = 0, world = 1, rounds = 1;
    uint32_t capacity = 0;  // keys: shard size and receive capacity
    Rccl rccl;
    hipStream_t sort_stream = nullptr;  // the context's stream
    hipStream_t comm_stream = nullptr;  // exchange rounds run here, beside the sorts
    std::vector<hipEvent_t> round_done;  // round r has landed in the receive buffer
    hipEvent_t grouped_ready = nullptr, sorts_done = nullptr;
    vrs_buffer grouped = nullptr, recv = nullptr, scratch = nullptr, hist = nullptr, prefix = nullptr, table = nullptr;
    std::vector<uint32_t> host_table;  // world x 257: every rank's top-byte prefix row and shard size
    std::string last_error;
    double max_imbalance = 1.15;
};

namespace {

int dfail(vrs_dist d, int code, const std::string &msg) {
    if (d) d->last_error = msg; else g_dist_error = msg;
    return code;
}
int dfail_ctx(vrs_dist d, int code, const char *what) {
    return dfail(d, code, std::string(what) + ": " + vrs_last_error(d->ctx));
}
#define VRS_D(d, call)                                        \
    do {                                                      \
        const int rc__ = (call);                              \
        if (rc__ != VRS_OK) return dfail_ctx((d), rc__, #call); \
    } while (0)
#define VRS_DHIP(d, call)                                                                             \
    do {                                                                                              \
        const hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) return dfail((d), VRS_ERROR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)
#define VRS_DNCCL(d, call)                                                                                   \
    do {                                                                                                     \
        const int r__ = (call);                                                                              \
        if (r__ != 0)                                                                                        \
            return dfail((d), VRS_ERROR_HIP,                                                                 \
                         std::string(#call) + ": " + ((d)->rccl.GetErrorString ? (d)->rccl.GetErrorString(r__) : "RCCL error")); \
    } while (0)

}  // namespace

extern "C" {

// Byte boundaries b[0] = 0 <= b[1] <= ... <= b[parts] = 256: part q owns top bytes [b[q], b[q+1]).  Greedy: boundary q
// is the byte at which the cumulative count first reaches q / parts of the total, snapped to whichever side is closer.
// Deterministic, identical on every rank.  (Python twin: distributed.plan_splitters.)
int vrs_dist_plan_splitters(const uint64_t *counts256, int parts, uint32_t *bounds) {
    if (!counts256 || !bounds || parts < 1) return dfail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "counts, bounds or parts invalid");
    uint64_t cum[257];
    cum[0] = 0;
    for (int i = 0; i < 256; ++i) cum[i + 1] = cum[i] + counts256[i];
    const double total = static_cast<double>(cum[256]);
    bounds[0] = 0;
    bounds[parts] = 256;
    for (int q = 1; q < parts; ++q) {
        const double target = total * q / parts;
        // first index with cum[idx] >= target (numpy.searchsorted(cum, target, side="left"))
        int hi = static_cast<int>(std::lower_bound(cum, cum + 257, target, [](uint64_t a, double t) { return static_cast<double>(a) < t; }) - cum);
        hi = std::min(std::max(hi, 0), 256);
        const int lo = std::max(hi - 1, 0);
        const int pick = std::fabs(static_cast<double>(cum[lo]) - target) <= std::fabs(static_cast<double>(cum[hi]) - target) ? lo : hi;
        bounds[q] = std::max<uint32_t>(static_cast<uint32_t>(pick), bounds[q - 1]);
    }
    return VRS_OK;
}

const char *vrs_dist_last_error(vrs_dist d) { return d ? d->last_error.c_str() : g_dist_error.c_str(); }

int vrs_dist_create(vrs_context ctx, void *nccl_comm, int rank, int world, uint32_t capacity_keys, int rounds, vrs_dist *out) {
    if (!out) return dfail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (!ctx || world < 1 || rank < 0 || rank >= world || capacity_keys == 0)
        return dfail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context, rank / world or capacity invalid");
    if (world > 1 && !nccl_comm) return dfail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "an RCCL communicator is required at world size > 1");
    vrs_dist d = new (std::nothrow) vrs_dist_t();
    if (!d) return dfail(nullptr, VRS_ERROR_OUT_OF_MEMORY, "host allocation failed");
    d->ctx = ctx;
    d->comm = nccl_comm;
    d->rank = rank;
    d->world = world;
    d->rounds = std::max(1, std::min(rounds, 256 / world));
    d->capacity = capacity_keys;
    d->sort_stream = static_cast<hipStream_t>(vrs_context_stream(ctx));
    if (nccl_comm && !load_rccl(d->rccl)) {
        delete d;
        return dfail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "no RCCL (librccl.so) could be bound in this process");
    }
    const auto cleanup = [&](int code, const std::string &msg) {
        vrs_dist_destroy(d);
        return dfail(nullptr, code, msg);
    };
    if (hipStreamCreateWithFlags(&d->comm_stream, hipStreamNonBlocking) != hipSuccess) return cleanup(VRS_ERROR_HIP, "hipStreamCreate failed");
    d->round_done.resize(static_cast<size_t>(d->rounds));
    for (auto &e : d->round_done)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return cleanup(VRS_ERROR_HIP, "hipEventCreate failed");
    if (hipEventCreateWithFlags(&d->grouped_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&d->sorts_done, hipEventDisableTiming) != hipSuccess)
        return cleanup(VRS_ERROR_HIP, "hipEventCreate failed");
    const size_t kb = static_cast<size_t>(capacity_keys) * sizeof(uint32_t);
    const uint32_t W = vrs_workgroup_count(capacity_keys, 32);
    if (vrs_buffer_create(ctx, kb, &d->grouped) || vrs_buffer_create(ctx, kb, &d->recv) || vrs_buffer_create(ctx, kb, &d->scratch) ||
        vrs_buffer_create(ctx, static_cast<size_t>(W) * 256 * 4, &d->hist) || vrs_buffer_create(ctx, 257 * 4, &d->prefix) ||
        vrs_buffer_create(ctx, static_cast<size_t>(world) * 257 * 4, &d->table))
        return cleanup(VRS_ERROR_OUT_OF_MEMORY, std::string("buffer allocation failed: ") + vrs_last_error(ctx));
    d->host_table.resize(static_cast<size_t>(world) * 257);
    *out = d;
    return VRS_OK;
}

int vrs_dist_destroy(vrs_dist d) {
    if (!d) return VRS_OK;
    if (d->comm_stream) (void)hipStreamSynchronize(d->comm_stream);
    for (vrs_buffer b : {d->grouped, d->recv, d->scratch, d->hist, d->prefix, d->table})
        if (b) (void)vrs_buffer_release(b);
    for (auto e : d->round_done)
        if (e) (void)hipEventDestroy(e);
    if (d->grouped_ready) (void)hipEventDestroy(d->grouped_ready);
    if (d->sorts_done) (void)hipEventDestroy(d->sorts_done);
    if (d->comm_stream) (void)hipStreamDestroy(d->comm_stream);
    delete d;  // the RCCL handle stays mapped: the process may hold communicators made by it
    return VRS_OK;
}

int vrs_dist_sort_keys_u32(vrs_dist d, vrs_buffer keys, uint32_t n, vrs_buffer *out_keys, uint32_t *out_count) {
    if (!d || !out_keys || !out_count) return dfail(d, VRS_ERROR_INVALID_ARGUMENT, "dist, out_keys or out_count is NULL");
    *out_keys = nullptr;
    *out_count = 0;
    if (n > d->capacity) return dfail(d, VRS_ERROR_INVALID_ARGUMENT, "shard larger than the capacity given at creation");
    if (n && !keys) return dfail(d, VRS_ERROR_INVALID_ARGUMENT, "keys is NULL");
    vrs_context ctx = d->ctx;
    const int world = d->world, me = d->rank, R = d->rounds;
    uint32_t *grouped = static_cast<uint32_t *>(vrs_buffer_device_ptr(d->grouped));
    uint32_t *recv = static_cast<uint32_t *>(vrs_buffer_device_ptr(d->recv));
    uint32_t *prefix = static_cast<uint32_t *>(vrs_buffer_device_ptr(d->prefix));
    uint32_t *table = static_cast<uint32_t *>(vrs_buffer_device_ptr(d->table));

    // the receive buffer may still be read by the sorts of the previous step's last round (they ran on the sort stream)
    VRS_DHIP(d, hipEventRecord(d->sorts_done, d->sort_stream));
    VRS_DHIP(d, hipStreamWaitEvent(d->comm_stream, d->sorts_done, 0));

    // 1. local step: one radix pass on the top byte groups the shard so that every key range is a contiguous slice;
    //    row 0 of the offset table is the exclusive prefix of the top-byte counts
    if (n) {
        vrs_push_constants pc{n, 24, vrs_workgroup_count(n, 32), 32};
        VRS_D(d, vrs_multi_radixsort_histograms(ctx, keys, d->hist, &pc));
        VRS_D(d, vrs_multi_radixsort(ctx, keys, d->grouped, d->hist, &pc));
        VRS_D(d, vrs_multi_radixsort_digit_offsets_device(ctx, d->prefix));
    } else {
        VRS_DHIP(d, hipMemsetAsync(prefix, 0, 256 * 4, d->sort_stream));  // an empty shard: nothing was launched
    }
    VRS_DHIP(d, hipMemcpyAsync(prefix + 256, &n, 4, hipMemcpyHostToDevice, d->sort_stream));  // word 256 = shard size
    VRS_DHIP(d, hipEventRecord(d->grouped_ready, d->sort_stream));

    // 2. ONE small collective: every rank learns every rank's prefix row (world x 1 KiB), from which each derives the
    //    same splitters, its send counts and its receive counts without further traffic
    if (world > 1) {
        VRS_DNCCL(d, d->rccl.AllGather(prefix, table, 257, kNcclUint32, d->comm, d->sort_stream));
    } else {
        VRS_DHIP(d, hipMemcpyAsync(table, prefix, 257 * 4, hipMemcpyDeviceToDevice, d->sort_stream));
    }
    VRS_DHIP(d, hipMemcpyAsync(d->host_table.data(), table, d->host_table.size() * 4, hipMemcpyDeviceToHost, d->sort_stream));
    VRS_DHIP(d, hipStreamSynchronize(d->sort_stream));
    std::vector<std::vector<uint64_t>> base(static_cast<size_t>(world), std::vector<uint64_t>(257));
    uint64_t counts[256] = {};
    uint64_t grand_total = 0;
    for (int q = 0; q < world; ++q) {
        const uint32_t *row = &d->host_table[static_cast<size_t>(q) * 257];
        for (int t = 0; t < 256; ++t) base[q][t] = row[t];
        base[q][256] = row[256];  // shard size: the end of top byte 255
        for (int t = 0; t < 256; ++t) counts[t] += base[q][t + 1] - base[q][t];
        grand_total += row[256];
    }
    const int P = world * R;  // part q * R + r = rank q, round r
    std::vector<uint32_t> parts(static_cast<size_t>(P) + 1);
    int rc = vrs_dist_plan_splitters(counts, P, parts.data());
    if (rc) return rc;
    // how many keys a rank receives; byte-aligned cuts cannot balance keys whose top bytes are too concentrated
    uint64_t worst = 0;
    for (int q = 0; q < world; ++q) {
        uint64_t s = 0;
        for (uint32_t t = parts[static_cast<size_t>(q) * R]; t < parts[static_cast<size_t>(q + 1) * R]; ++t) s += counts[t];
        worst = std::max(worst, s);
    }
    const double ideal = std::max(static_cast<double>(grand_total) / world, 1.0);
    if (static_cast<double>(worst) > d->max_imbalance * ideal || worst > d->capacity)
        return dfail(d, VRS_ERROR_UNBALANCED,
                     "top bytes too concentrated for byte-aligned key ranges (small or clustered keys): use the sampled-splitter "
                     "path of vkradixsort_amd.distributed.RangeShardedSort, or a larger capacity");

    // what I receive in round r from source s, and where it lands: rounds ascending, sources ascending
    std::vector<uint64_t> round_off(static_cast<size_t>(R) + 1, 0);
    std::vector<std::vector<uint64_t>> recv_rs(static_cast<size_t>(R), std::vector<uint64_t>(static_cast<size_t>(world)));
    for (int r = 0; r < R; ++r) {
        const uint32_t lo = parts[static_cast<size_t>(me) * R + r], hi = parts[static_cast<size_t>(me) * R + r + 1];
        uint64_t tot = 0;
        for (int s = 0; s < world; ++s) {
            recv_rs[r][s] = base[s][hi] - base[s][lo];
            tot += recv_rs[r][s];
        }
        round_off[r + 1] = round_off[r] + tot;
    }
    const uint64_t total = round_off[R];
    if (total > d->capacity) return dfail(d, VRS_ERROR_UNBALANCED, "this rank's key range holds more keys than the capacity");

    // 3. exchange, round by round on the exchange stream: grouped send/recv with every peer; my own slice is a device copy
    VRS_DHIP(d, hipStreamWaitEvent(d->comm_stream, d->grouped_ready, 0));
    for (int r = 0; r < R; ++r) {
        if (world > 1) VRS_DNCCL(d, d->rccl.GroupStart());
        uint64_t off = round_off[r];
        for (int s = 0; s < world; ++s) {
            const uint64_t cnt = recv_rs[r][s];
            if (cnt && s != me) VRS_DNCCL(d, d->rccl.Recv(recv + off, cnt, kNcclUint32, s, d->comm, d->comm_stream));
            if (cnt && s == me) {
                const uint64_t a = base[me][parts[static_cast<size_t>(me) * R + r]];
                VRS_DHIP(d, hipMemcpyAsync(recv + off, grouped + a, cnt * 4, hipMemcpyDeviceToDevice, d->comm_stream));
            }
            off += cnt;
        }
        for (int dst = 0; dst < world; ++dst) {
            if (dst == me) continue;
            const uint64_t a = base[me][parts[static_cast<size_t>(dst) * R + r]], b = base[me][parts[static_cast<size_t>(dst) * R + r + 1]];
            if (b > a) VRS_DNCCL(d, d->rccl.Send(grouped + a, b - a, kNcclUint32, dst, d->comm, d->comm_stream));
        }
        if (world > 1) VRS_DNCCL(d, d->rccl.GroupEnd());
        VRS_DHIP(d, hipEventRecord(d->round_done[static_cast<size_t>(r)], d->comm_stream));
    }
    // 4. round r's keys are sorted (the library's four-pass sort) while the later rounds are still on the wire; the
    //    sub-ranges are disjoint and ascending, so their concatenation is the sorted range: nothing to merge
    for (int r = 0; r < R; ++r) {
        const uint64_t cnt = round_off[r + 1] - round_off[r];
        VRS_DHIP(d, hipStreamWaitEvent(d->sort_stream, d->round_done[static_cast<size_t>(r)], 0));
        if (!cnt) continue;
        vrs_buffer view = nullptr;
        VRS_D(d, vrs_buffer_wrap(ctx, recv + round_off[r], cnt * 4, &view));
        rc = vrs_sort_keys_u32(ctx, view, d->scratch, static_cast<uint32_t>(cnt));
        (void)vrs_buffer_release(view);
        if (rc) return dfail_ctx(d, rc, "vrs_sort_keys_u32 (received sub-range)");
    }
    *out_keys = d->recv;
    *out_count = static_cast<uint32_t>(total);
    return VRS_OK;
}

}  // extern "C"
