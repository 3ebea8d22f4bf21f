// vrs_dist.hip -- the multi-GPU step behind the C ABI: key-range sharded sort, one rank per GPU (BASELINE.json configs[4];
// SURVEY.md section 8e).  The reference has no multi-GPU code (no collective call site anywhere under /root/reference);
// north_star defines the path: shard by key range across the GPUs of a node, one all-to-all over xGMI between the local
// step and the local sorts.
//
// Host orchestration only -- every device step goes through the public C ABI of this library (the entry points a C++ host
// would call).  Two shapes of the step:
//
//   hybrid shape (default; the single-GPU hybrid sort with the exchange between its two MSD passes, 28 B/key per GPU):
//     1. vrs_msd_partition_u32 of the shard: one counting read (top-14-bit bucket histogram + top-byte counts) and the
//        first MSD pass -- the shard grouped by the top byte of the key range;
//     2. ONE all-gather of every rank's row (top-byte counts of its eight input slices, shard size, bucket shift, status)
//        and ONE all-reduce of the 16384-bin bucket histograms (64 KB); every rank derives the same byte-aligned
//        splitters, all send / receive counts and where every message lands from the gathered rows;
//     3. R rounds of grouped send / recv on a second stream: one message per (sender, top byte), landing in top-byte
//        order -- the receive buffer of a round IS the first MSD pass's output for that round's key sub-range;
//     4. per round, while the later rounds are on the wire: vrs_msd_finish_u32 (second MSD pass + LDS-local sort) with the
//        all-reduced histogram masked to the round's buckets.  A round whose plan refuses (a bucket beyond the local
//        sort's capacity) is sorted by vrs_sort_keys_u32 instead -- a rank-local matter.
//   byte shape (the DEFAULT since round 5; under VRS_DIST_SHAPE=hybrid what ALL ranks fall back to together, from the gathered
//     rows: a total whose global top-14-bit buckets would not fit the local sort -- about 2e8 keys, so every step of 8 x 1e8 --,
//     a key range below 27 bits, ranks that probed different ranges; 28 + 8 (1 - 1 / world) B/key per GPU): contract partition
//     pass by the top byte (12 B/key), the same exchange -- but the rank's own keys stay where the partition pass wrote them --,
//     and per received sub-range the pool form's second half (vrs_msd_finish_grouped_split_u32: 16 B/key, nothing is read to be
//     counted; VRS_DIST_POOL_FINISH=0: vrs_msd_finish_grouped_u32 -- ONE counting read, the second MSD pass by the next 8
//     bits, the local sort: 20 B/key).  A sub-range it cannot take (a top byte with more keys than its 256 sub-buckets
//     hold: then one message per (sender, round), decided by all ranks from the summed top-byte counts; or a refused plan)
//     is sorted whole by vrs_sort_keys_u32_ranged (28 B/key).  A total too large for the hybrid shape is remembered: the
//     following steps start here (no counting read + first pass for nothing, one all-gather), every 64th looks again.
//   In both shapes the collectives run on the exchange stream beside the local step's last kernel, and every round's second
//   half is enqueued before any of their plans is looked at: the host waits once per step, for the collectives.
//
// Every decision to leave the step is made by all ranks from the same gathered data (a rank that cannot take part says so
// in its row and still joins the collectives), so no rank is left waiting inside a collective.
//
// The wire is a table of five functions (vrs_dist_transport): RCCL bound at run time by dlopen (vrs_dist_create: no
// link-time dependency, and a process that already carries an RCCL -- PyTorch ships its own -- keeps using THAT copy), or
// anything the caller supplies (vrs_dist_create_with_transport).  vrs_dist_loopback_* is an in-process transport -- the
// ranks are host threads of one process, every transfer a device copy ordered by events -- for one process driving several
// GPUs, and for running the rank-to-rank bookkeeping of this file on a single GPU (tests/test_gpu_dist.py).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "vkradixsort_amd.h"

namespace {

// The bucket histograms of a step's rounds, one table per round in vrs_msd_finish_u32's layout, made in ONE launch before the
// rounds start (five small fills and copies per round were 25 us of every round): round r's table = the all-reduced
// histogram masked to the round's top bytes [lo, hi), slice counts and flag zero, the shift word set.
struct RoundCuts {
    uint32_t lo[32], hi[32];
    uint32_t shift;
};
__global__ __launch_bounds__(1024) void round_tables_kernel(const uint32_t *__restrict__ reduced, uint32_t *__restrict__ tables, RoundCuts cuts) {
    const uint32_t r = blockIdx.x;
    uint32_t *t = tables + static_cast<size_t>(r) * VRS_MSD_COUNT_WORDS;
    for (uint32_t c = threadIdx.x; c < VRS_MSD_COUNT_WORDS; c += 1024u) {
        uint32_t v = 0;
        if (c < 16384u && (c >> 6) >= cuts.lo[r] && (c >> 6) < cuts.hi[r]) v = reduced[c];
        if (c == VRS_MSD_SHIFT_WORD) v = cuts.shift;
        t[c] = v;
    }
}

// Sampled-splitter steps: `count` keys of the shard at evenly spaced positions (an empty shard adds zeros nobody weighs).
__global__ __launch_bounds__(256) void sample_shard_kernel(const uint32_t *__restrict__ keys, uint32_t n, uint32_t *__restrict__ out, uint32_t count) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < count) out[i] = n ? keys[static_cast<uint64_t>(i) * (n - 1u) / (count - 1u)] : 0u;
}

// ---- RCCL, bound at run time (rccl.h: ncclResult_t == int, ncclSuccess == 0, ncclUint32 == 3, ncclSum == 0)
struct Rccl {
    void *handle = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok() const { return AllGather && AllReduce && Send && Recv && GroupStart && GroupEnd; }
};
constexpr int kNcclUint32 = 3, kNcclSum = 0;

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
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(src, "ncclAllReduce"));
    r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(src, "ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(src, "ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(src, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(src, "ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(src, "ncclGetErrorString"));
    return r.ok();
}

// the RCCL flavour of the transport table
struct RcclEndpoint {
    Rccl rccl;
    void *comm = nullptr;
};
int rccl_all_gather(void *u, const void *s, void *r, size_t words, void *st) {
    auto *e = static_cast<RcclEndpoint *>(u);
    return e->rccl.AllGather(s, r, words, kNcclUint32, e->comm, static_cast<hipStream_t>(st));
}
int rccl_all_reduce(void *u, const void *s, void *r, size_t words, void *st) {
    auto *e = static_cast<RcclEndpoint *>(u);
    return e->rccl.AllReduce(s, r, words, kNcclUint32, kNcclSum, e->comm, static_cast<hipStream_t>(st));
}
int rccl_group_start(void *u) { return static_cast<RcclEndpoint *>(u)->rccl.GroupStart(); }
int rccl_group_end(void *u) { return static_cast<RcclEndpoint *>(u)->rccl.GroupEnd(); }
int rccl_send(void *u, const void *b, size_t words, int peer, void *st) {
    auto *e = static_cast<RcclEndpoint *>(u);
    return e->rccl.Send(b, words, kNcclUint32, peer, e->comm, static_cast<hipStream_t>(st));
}
int rccl_recv(void *u, void *b, size_t words, int peer, void *st) {
    auto *e = static_cast<RcclEndpoint *>(u);
    return e->rccl.Recv(b, words, kNcclUint32, peer, e->comm, static_cast<hipStream_t>(st));
}
const char *rccl_error_string(void *u, int code) {
    auto *e = static_cast<RcclEndpoint *>(u);
    return e->rccl.GetErrorString ? e->rccl.GetErrorString(code) : "RCCL error";
}

// the row every rank contributes to the one all-gather of a step
constexpr int kRowSlices = 8 * 256;         // top-byte counts of the shard's eight input slices (hybrid shape) / [0, 256): the shard's top-byte counts (byte shape)
constexpr int kRowN = kRowSlices;           // shard size
constexpr int kRowStatus = kRowSlices + 1;  // 0, or the VRS_ERROR_* that keeps this rank from taking part
constexpr int kRowCapacity = kRowSlices + 2;
constexpr int kRowShift = kRowSlices + 3;   // hybrid shape: the probed bucket shift ...
constexpr int kRowOver = kRowSlices + 4;    // ... and != 0: a key of the shard lies above the probed range
constexpr int kRowWords = kRowSlices + 8;
constexpr uint32_t kSamplesPerRank = 2048;  // sampled-splitter steps: what every rank adds to the pool (<= kRowSlices: the pool travels in the row / table buffers)
constexpr uint64_t kHybridShapeMaxBucket = 14333;  // msd_local_capacity of bare uint32 keys (vrs_msd_hybrid.hip); VRS_DIST_HYBRID_MAX_BUCKET (tests) lowers it

}  // namespace

struct vrs_dist_t {
    vrs_context ctx = nullptr;
    int device = 0;
    int rank = 0, world = 1, rounds = 1;
    uint32_t capacity = 0;  // keys: shard size and receive capacity
    bool has_transport = false;
    vrs_dist_transport tr{};
    RcclEndpoint *rccl = nullptr;  // owned; the transport's `user` when vrs_dist_create made it
    bool byte_shape_only = true;   // unless VRS_DIST_SHAPE=hybrid
    uint64_t hybrid_max_bucket = kHybridShapeMaxBucket;
    bool too_large_for_hybrid = false;  // the last step that tried found N_total / 16384 beyond the local sort: the same on every rank
    uint64_t steps = 0;
    hipStream_t sort_stream = nullptr;  // the context's stream
    hipStream_t comm_stream = nullptr;  // exchange rounds run here, beside the sorts
    std::vector<hipEvent_t> round_done;  // round r has landed in the receive buffer
    hipEvent_t grouped_ready = nullptr, sorts_done = nullptr, counts_ready = nullptr;
    vrs_buffer grouped = nullptr, recv = nullptr, scratch = nullptr, hist = nullptr, row = nullptr, table = nullptr;
    vrs_buffer counts = nullptr, reduced = nullptr, round_counts = nullptr;
    std::vector<uint32_t> host_table;  // world x kRowWords
    uint32_t host_row_tail[8] = {};  // host words on their way into device rows: [0..2] shard size, status, capacity
    std::string last_error;
    double max_imbalance = 1.15;
    uint64_t hybrid_rounds = 0, fallback_rounds = 0, byte_steps = 0, grouped_rounds = 0;
    bool no_grouped_finish = false;  // VRS_DIST_GROUPED_FINISH=0: the byte shape with whole ranged sorts (tests, A/B)
    vrs_buffer splitters = nullptr;  // sampled-splitter steps: the P - 1 cut keys on the device
    uint32_t host_splitters[256] = {};
    uint64_t splitter_steps = 0;
    bool no_sampled_splitters = false;  // VRS_DIST_SAMPLED_SPLITTERS=0: concentrated top bytes return VRS_ERROR_UNBALANCED (tests)
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

std::string transport_error(vrs_dist d, const char *what, int code) {
    const char *s = d->tr.error_string ? d->tr.error_string(d->tr.user, code) : nullptr;
    return std::string(what) + " failed: " + (s ? s : "transport error") + " (" + std::to_string(code) + ")";
}
#define VRS_DTR(d, what, call)                                                        \
    do {                                                                              \
        const int r__ = (call);                                                       \
        if (r__ != 0) return dfail((d), VRS_ERROR_HIP, transport_error((d), (what), r__)); \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// In-process transport: the ranks are host threads of one process that share a hub.  A collective is a rendezvous on the
// host (every rank publishes what it offers and an event that says when its stream has produced it), after which every
// rank enqueues, on its own stream, the device copies that bring ITS data in; a second rendezvous hands every rank the
// events behind which its own buffers are free again.  Same GPU or peer GPUs of one process (device-to-device copies).
struct LoopOp {
    const void *src;
    void *dst;
    size_t words;
    int peer;
};
struct LoopEndpoint {
    struct vrs_dist_loopback_t *hub;
    int rank;
    bool grouping = false;
    std::vector<LoopOp> sends, recvs;  // of the open group
    hipStream_t group_stream = nullptr;
};
}  // namespace

struct vrs_dist_loopback_t {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool broken = false;  // a rank failed inside a collective: everyone leaves with an error instead of waiting
    // A wire that costs something (vrs_dist_loopback_set_wire; VRS_LOOPBACK_LINK_GBPS / VRS_LOOPBACK_LATENCY_US at creation): every group of
    // sends / receives holds the receiver's stream for (the most bytes it gets from ONE peer) / link_gbps -- the peers' links carry their
    // messages side by side, a peer's messages share its link -- plus latency_us, every gather / reduce for latency_us.  A model of xGMI's
    // point-to-point links for the schedule's sake (what of the wire is exposed, per R), not a measurement of anything.
    double link_gbps = 0.0;    // 0: the wire is free (device copies only)
    double latency_us = 0.0;
    bool host_memory = false;  // vrs_dist_loopback_create_host: the buffers are HOST memory, every transfer a memcpy at the rendezvous, no HIP call
                               // anywhere (the hub's matching and barriers run without a device: the sanitizer builds' CPU tests)
    std::vector<LoopEndpoint> ends;
    // what the ranks publish for the collective in flight
    std::vector<const void *> src;
    std::vector<size_t> words;
    std::vector<hipEvent_t> ready, done;  // per rank: "my data is produced" / "my copies have run"
    std::vector<std::vector<LoopOp>> sends;
};

namespace {

// host barrier of the hub's ranks; false = the hub is broken (a peer failed)
bool loop_barrier(vrs_dist_loopback_t *h) {
    std::unique_lock<std::mutex> lk(h->m);
    if (h->broken) return false;
    const uint64_t gen = h->generation;
    if (++h->arrived == h->world) {
        h->arrived = 0;
        ++h->generation;
        h->cv.notify_all();
        return true;
    }
    h->cv.wait(lk, [&] { return h->generation != gen || h->broken; });
    return !h->broken;
}
int loop_break(vrs_dist_loopback_t *h, int code) {
    std::lock_guard<std::mutex> lk(h->m);
    h->broken = true;
    h->cv.notify_all();
    return code;
}
constexpr int kLoopErrHip = 1, kLoopErrPeer = 2, kLoopErrUsage = 3;
__global__ void wire_delay_kernel(unsigned long long ticks) {  // holds its stream for `ticks` of the 100 MHz wall clock
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
bool loop_wire(vrs_dist_loopback_t *h, hipStream_t st, size_t bytes_on_busiest_link) {
    if (h->host_memory || st == nullptr || (h->link_gbps <= 0.0 && h->latency_us <= 0.0)) return true;
    const double us = h->latency_us + (h->link_gbps > 0.0 ? static_cast<double>(bytes_on_busiest_link) / (h->link_gbps * 1e3) : 0.0);
    if (us <= 0.0) return true;
    hipLaunchKernelGGL(wire_delay_kernel, dim3(1), dim3(64), 0, st, static_cast<unsigned long long>(us * 100.0));
    return hipGetLastError() == hipSuccess;
}
// the hub's three device operations; in host-memory mode the barriers alone order the ranks (a copy is done when memcpy returns)
bool loop_record(vrs_dist_loopback_t *h, hipEvent_t ev, hipStream_t st) { return h->host_memory || hipEventRecord(ev, st) == hipSuccess; }
bool loop_wait(vrs_dist_loopback_t *h, hipStream_t st, hipEvent_t ev) { return h->host_memory || hipStreamWaitEvent(st, ev, 0) == hipSuccess; }
bool loop_copy(vrs_dist_loopback_t *h, void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t st, bool sync) {
    if (h->host_memory) {
        std::memcpy(dst, src, bytes);
        return true;
    }
    if (hipMemcpyAsync(dst, src, bytes, kind, st) != hipSuccess) return false;
    return !sync || hipStreamSynchronize(st) == hipSuccess;
}

// after the copies of a collective: every rank waits (on its stream) until all ranks' copies have run, so that whatever it
// enqueues next may overwrite the buffers it offered
int loop_release(LoopEndpoint *e, hipStream_t st) {
    vrs_dist_loopback_t *h = e->hub;
    if (!loop_record(h, h->done[e->rank], st)) return loop_break(h, kLoopErrHip);
    if (!loop_barrier(h)) return kLoopErrPeer;
    for (int s = 0; s < h->world; ++s)
        if (s != e->rank && !loop_wait(h, st, h->done[s])) return loop_break(h, kLoopErrHip);
    if (!loop_barrier(h)) return kLoopErrPeer;  // nobody re-records its events before everybody has waited on them
    return 0;
}

int loop_gather_like(void *u, const void *send, void *recv, size_t words, void *stream, bool reduce) {
    auto *e = static_cast<LoopEndpoint *>(u);
    vrs_dist_loopback_t *h = e->hub;
    hipStream_t st = static_cast<hipStream_t>(stream);
    h->src[e->rank] = send;
    h->words[e->rank] = words;
    if (!loop_record(h, h->ready[e->rank], st)) return loop_break(h, kLoopErrHip);
    if (!loop_barrier(h)) return kLoopErrPeer;
    for (int s = 0; s < h->world; ++s)
        if (h->words[s] != words) return loop_break(h, kLoopErrUsage);
    if (h->world > 1 && !loop_wire(h, st, words * 4)) return loop_break(h, kLoopErrHip);
    if (!reduce) {
        for (int s = 0; s < h->world; ++s) {
            if (s != e->rank && !loop_wait(h, st, h->ready[s])) return loop_break(h, kLoopErrHip);
            if (!loop_copy(h, static_cast<uint32_t *>(recv) + static_cast<size_t>(s) * words, h->src[s], words * 4, hipMemcpyDeviceToDevice, st, false))
                return loop_break(h, kLoopErrHip);
        }
    } else {
        // sum of the ranks' buffers through the host (a few KB: this transport is not the fast path of anything)
        std::vector<uint32_t> acc(words, 0), tmp(words);
        for (int s = 0; s < h->world; ++s) {
            if (s != e->rank && !loop_wait(h, st, h->ready[s])) return loop_break(h, kLoopErrHip);
            if (!loop_copy(h, tmp.data(), h->src[s], words * 4, hipMemcpyDeviceToHost, st, true)) return loop_break(h, kLoopErrHip);
            for (size_t i = 0; i < words; ++i) acc[i] += tmp[i];
        }
        // every rank has read every offer before anyone's result may land in a buffer that is also an offer (in-place use)
        if (!loop_barrier(h)) return kLoopErrPeer;
        if (!loop_copy(h, recv, acc.data(), words * 4, hipMemcpyHostToDevice, st, true)) return loop_break(h, kLoopErrHip);
    }
    return loop_release(e, st);
}
int loop_all_gather(void *u, const void *s, void *r, size_t words, void *st) { return loop_gather_like(u, s, r, words, st, false); }
int loop_all_reduce(void *u, const void *s, void *r, size_t words, void *st) { return loop_gather_like(u, s, r, words, st, true); }
int loop_group_start(void *u) {
    auto *e = static_cast<LoopEndpoint *>(u);
    if (e->grouping) return loop_break(e->hub, kLoopErrUsage);  // (a usage error of one rank: its peers leave their collectives too)
    e->grouping = true;
    e->sends.clear();
    e->recvs.clear();
    e->group_stream = nullptr;
    return 0;
}
int loop_send(void *u, const void *b, size_t words, int peer, void *st) {
    auto *e = static_cast<LoopEndpoint *>(u);
    if (!e->grouping || peer < 0 || peer >= e->hub->world || peer == e->rank) return loop_break(e->hub, kLoopErrUsage);
    e->sends.push_back(LoopOp{b, nullptr, words, peer});
    e->group_stream = static_cast<hipStream_t>(st);
    return 0;
}
int loop_recv(void *u, void *b, size_t words, int peer, void *st) {
    auto *e = static_cast<LoopEndpoint *>(u);
    if (!e->grouping || peer < 0 || peer >= e->hub->world || peer == e->rank) return loop_break(e->hub, kLoopErrUsage);
    e->recvs.push_back(LoopOp{nullptr, b, words, peer});
    e->group_stream = static_cast<hipStream_t>(st);
    return 0;
}
// the k-th receive a rank posted from peer p takes the k-th send p posted to that rank (the order RCCL matches them in)
int loop_group_end(void *u) {
    auto *e = static_cast<LoopEndpoint *>(u);
    vrs_dist_loopback_t *h = e->hub;
    if (!e->grouping) return loop_break(h, kLoopErrUsage);
    e->grouping = false;
    hipStream_t st = e->group_stream;  // nullptr: this rank has nothing to move in this group (it still takes part)
    if (h->host_memory && (!e->sends.empty() || !e->recvs.empty())) st = reinterpret_cast<hipStream_t>(h);  // (host memory: any non-null token; never handed to HIP)
    h->sends[e->rank] = e->sends;
    if (st && !loop_record(h, h->ready[e->rank], st)) return loop_break(h, kLoopErrHip);
    // a rank without operations records nothing: nobody will wait on its event, because nobody receives from it
    if (!loop_barrier(h)) return kLoopErrPeer;
    {   // the wire: this group's receives, per peer (a peer's messages share one link, the peers' links run side by side)
        std::vector<size_t> from(static_cast<size_t>(h->world), 0);
        size_t busiest = 0;
        for (const LoopOp &r : e->recvs) busiest = std::max(busiest, from[static_cast<size_t>(r.peer)] += r.words * 4);
        if (!e->recvs.empty() && !loop_wire(h, st, busiest)) return loop_break(h, kLoopErrHip);
    }
    std::vector<size_t> next(static_cast<size_t>(h->world), 0);
    for (const LoopOp &r : e->recvs) {
        const std::vector<LoopOp> &theirs = h->sends[r.peer];
        size_t &k = next[static_cast<size_t>(r.peer)];
        while (k < theirs.size() && theirs[k].peer != e->rank) ++k;
        if (k == theirs.size() || theirs[k].words != r.words) return loop_break(h, kLoopErrUsage);  // unmatched receive
        if (!loop_wait(h, st, h->ready[r.peer]) || !loop_copy(h, r.dst, theirs[k].src, r.words * 4, hipMemcpyDeviceToDevice, st, false))
            return loop_break(h, kLoopErrHip);
        ++k;
    }
    // release: senders may reuse what they offered once the receivers' copies have run
    if (st) {
        if (!loop_record(h, h->done[e->rank], st)) return loop_break(h, kLoopErrHip);
    }
    if (!loop_barrier(h)) return kLoopErrPeer;
    if (st)
        for (const LoopOp &s : e->sends)
            if (!loop_wait(h, st, h->done[s.peer])) return loop_break(h, kLoopErrHip);
    if (!loop_barrier(h)) return kLoopErrPeer;
    return 0;
}
const char *loop_error_string(void *, int code) {
    switch (code) {
        case kLoopErrHip: return "loopback transport: a HIP call failed";
        case kLoopErrPeer: return "loopback transport: another rank failed inside a collective";
        case kLoopErrUsage: return "loopback transport: mismatched collective (sizes, peers or grouping differ between the ranks)";
        default: return "loopback transport error";
    }
}

}  // namespace

extern "C" {

int vrs_dist_loopback_create(int world, vrs_dist_loopback *out) {
    if (!out || world < 1 || world > 64) return dfail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "out is NULL or world is not in 1 .. 64");
    *out = nullptr;
    auto *h = new (std::nothrow) vrs_dist_loopback_t();
    if (!h) return dfail(nullptr, VRS_ERROR_OUT_OF_MEMORY, "host allocation failed");
    h->world = world;
    h->ends.resize(static_cast<size_t>(world));
    h->src.assign(static_cast<size_t>(world), nullptr);
    h->words.assign(static_cast<size_t>(world), 0);
    h->sends.resize(static_cast<size_t>(world));
    h->ready.assign(static_cast<size_t>(world), nullptr);
    h->done.assign(static_cast<size_t>(world), nullptr);
    for (int r = 0; r < world; ++r) {
        h->ends[static_cast<size_t>(r)].hub = h;
        h->ends[static_cast<size_t>(r)].rank = r;
    }
    if (const char *v = std::getenv("VRS_LOOPBACK_LINK_GBPS")) h->link_gbps = std::max(0.0, std::atof(v));
    if (const char *v = std::getenv("VRS_LOOPBACK_LATENCY_US")) h->latency_us = std::max(0.0, std::atof(v));
    *out = h;
    return VRS_OK;
}

int vrs_dist_loopback_set_wire(vrs_dist_loopback hub, double link_gbps, double latency_us) {
    if (!hub || link_gbps < 0.0 || latency_us < 0.0) return dfail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "hub is NULL or a rate / latency is negative");
    std::lock_guard<std::mutex> lk(hub->m);  // (between steps: no collective is in flight)
    hub->link_gbps = link_gbps;
    hub->latency_us = latency_us;
    return VRS_OK;
}

// The same hub over HOST memory: send / recv / gather buffers are host pointers, every transfer a memcpy made at the rendezvous, the
// `hip_stream` arguments ignored -- no HIP call anywhere, so the hub's matching, barriers and failure paths run (and can be put under
// ThreadSanitizer) on a machine without a GPU.  Not a transport for vrs_dist_create_with_transport (its buffers are device memory).
int vrs_dist_loopback_create_host(int world, vrs_dist_loopback *out) {
    const int rc = vrs_dist_loopback_create(world, out);
    if (rc == VRS_OK) (*out)->host_memory = true;
    return rc;
}

// Fills `out` with rank `rank`'s end of the hub.  Call on the thread (and with the device current) that will drive this rank:
// the rank's events are made here.
int vrs_dist_loopback_transport(vrs_dist_loopback hub, int rank, vrs_dist_transport *out) {
    if (!hub || !out || rank < 0 || rank >= hub->world) return dfail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "hub, out or rank invalid");
    const size_t r = static_cast<size_t>(rank);
    if (!hub->host_memory && !hub->ready[r] && (hipEventCreateWithFlags(&hub->ready[r], hipEventDisableTiming) != hipSuccess ||
                           hipEventCreateWithFlags(&hub->done[r], hipEventDisableTiming) != hipSuccess))
        return dfail(nullptr, VRS_ERROR_HIP, "hipEventCreate failed");
    out->user = &hub->ends[r];
    out->all_gather = loop_all_gather;
    out->all_reduce = loop_all_reduce;
    out->group_start = loop_group_start;
    out->send = loop_send;
    out->recv = loop_recv;
    out->group_end = loop_group_end;
    out->error_string = loop_error_string;
    return VRS_OK;
}

int vrs_dist_loopback_destroy(vrs_dist_loopback hub) {
    if (!hub) return VRS_OK;
    for (auto e : hub->ready)
        if (e) (void)hipEventDestroy(e);
    for (auto e : hub->done)
        if (e) (void)hipEventDestroy(e);
    delete hub;
    return VRS_OK;
}

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

// The cut keys of a sampled-splitter step: `samples` holds per_rank keys of every rank's shard (rank-major), taken at evenly spaced
// positions; a sample of rank q stands for shard_sizes[q] / per_rank keys (an empty shard's samples count for nothing).  Cut key p
// (1 <= p < parts) is the first sample, in key order, at which p / parts of all keys have gone by; range r = number of cut keys <= key.
// Host only, deterministic, the same on every rank.
int vrs_dist_plan_sampled_splitters(const uint32_t *samples, const uint64_t *shard_sizes, int world, uint32_t per_rank, int parts,
                                    uint32_t *splitters) {
    if (!samples || !shard_sizes || !splitters || world < 1 || per_rank == 0 || parts < 1)
        return dfail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "samples, shard sizes, splitters, world, per_rank or parts invalid");
    std::vector<std::pair<uint32_t, double>> pool;  // (key, keys it stands for)
    pool.reserve(static_cast<size_t>(world) * per_rank);
    uint64_t total = 0;
    for (int q = 0; q < world; ++q) {
        total += shard_sizes[q];
        if (!shard_sizes[q]) continue;
        const double w = static_cast<double>(shard_sizes[q]) / per_rank;
        for (uint32_t i = 0; i < per_rank; ++i) pool.emplace_back(samples[static_cast<size_t>(q) * per_rank + i], w);
    }
    std::sort(pool.begin(), pool.end(), [](const std::pair<uint32_t, double> &a, const std::pair<uint32_t, double> &b) { return a.first < b.first; });
    double cum = 0;
    size_t i = 0;
    for (int p = 1; p < parts; ++p) {
        const double target = static_cast<double>(total) * p / parts;
        while (i < pool.size() && cum + pool[i].second < target) cum += pool[i++].second;
        splitters[p - 1] = i < pool.size() ? pool[i].first : 0xFFFFFFFFu;
    }
    return VRS_OK;
}

const char *vrs_dist_last_error(vrs_dist d) { return d ? d->last_error.c_str() : g_dist_error.c_str(); }

int vrs_dist_create_with_transport(vrs_context ctx, const vrs_dist_transport *transport, int rank, int world,
                                   uint32_t capacity_keys, int rounds, vrs_dist *out) {
    if (!out) return dfail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (!ctx || world < 1 || rank < 0 || rank >= world || capacity_keys == 0)
        return dfail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "context, rank / world or capacity invalid");
    if (world > 1 && !transport) return dfail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "a transport is required at world size > 1");
    if (transport && !(transport->all_gather && transport->all_reduce && transport->group_start && transport->send && transport->recv && transport->group_end))
        return dfail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "the transport table has an empty slot");
    vrs_dist d = new (std::nothrow) vrs_dist_t();
    if (!d) return dfail(nullptr, VRS_ERROR_OUT_OF_MEMORY, "host allocation failed");
    d->ctx = ctx;
    d->device = vrs_context_device(ctx);
    d->rank = rank;
    d->world = world;
    d->rounds = std::max(1, std::min(rounds, 32 / std::min(world, 32)));  // world * rounds <= 32 parts: a round keeps every XCD of the second pass busy
    if (world > 32) d->rounds = 1;
    d->capacity = capacity_keys;
    if (transport) {
        d->tr = *transport;
        d->has_transport = true;
    }
    // The byte shape is the default since round 5: with its rounds finished by the pool form's second half and the rank's own keys left
    // in place it moves 28 + 8 (1 - 1 / world) bytes per key where the hybrid shape moves 28 + 8, needs one collective less, and takes
    // every total (world size 1, 10^8 keys: 0.63 against 0.82 ms).  VRS_DIST_SHAPE=hybrid: the hybrid shape wherever its buckets fit.
    const char *shape = std::getenv("VRS_DIST_SHAPE");
    d->byte_shape_only = !(shape && std::strcmp(shape, "hybrid") == 0);
    if (const char *gf = std::getenv("VRS_DIST_GROUPED_FINISH")) d->no_grouped_finish = std::strcmp(gf, "0") == 0;
    if (const char *ss = std::getenv("VRS_DIST_SAMPLED_SPLITTERS")) d->no_sampled_splitters = std::strcmp(ss, "0") == 0;
    if (const char *mb = std::getenv("VRS_DIST_HYBRID_MAX_BUCKET")) {  // test knob: the "total too large for the hybrid shape" path at test sizes
        const long v = std::atol(mb);
        if (v > 0 && static_cast<uint64_t>(v) < kHybridShapeMaxBucket) d->hybrid_max_bucket = static_cast<uint64_t>(v);
    }
    d->sort_stream = static_cast<hipStream_t>(vrs_context_stream(ctx));
    const auto cleanup = [&](int code, const std::string &msg) {
        vrs_dist_destroy(d);
        return dfail(nullptr, code, msg);
    };
    // the exchange stream and every event belong to the CONTEXT's device, whatever device the calling thread had current
    if (hipSetDevice(d->device) != hipSuccess) return cleanup(VRS_ERROR_HIP, "hipSetDevice failed");
    if (hipStreamCreateWithFlags(&d->comm_stream, hipStreamNonBlocking) != hipSuccess) return cleanup(VRS_ERROR_HIP, "hipStreamCreate failed");
    d->round_done.resize(static_cast<size_t>(d->rounds));
    for (auto &e : d->round_done)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return cleanup(VRS_ERROR_HIP, "hipEventCreate failed");
    if (hipEventCreateWithFlags(&d->grouped_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&d->sorts_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&d->counts_ready, hipEventDisableTiming) != hipSuccess)
        return cleanup(VRS_ERROR_HIP, "hipEventCreate failed");
    const size_t kb = static_cast<size_t>(capacity_keys) * sizeof(uint32_t);
    const uint32_t W = vrs_workgroup_count(capacity_keys, 32);
    const size_t cw = static_cast<size_t>(VRS_MSD_COUNT_WORDS) * 4;
    if (vrs_buffer_create(ctx, kb, &d->grouped) || vrs_buffer_create(ctx, kb, &d->recv) || vrs_buffer_create(ctx, kb, &d->scratch) ||
        vrs_buffer_create(ctx, static_cast<size_t>(W) * 256 * 4, &d->hist) || vrs_buffer_create(ctx, kRowWords * 4, &d->row) ||
        vrs_buffer_create(ctx, static_cast<size_t>(world) * kRowWords * 4, &d->table) || vrs_buffer_create(ctx, cw, &d->counts) ||
        vrs_buffer_create(ctx, cw, &d->reduced) || vrs_buffer_create(ctx, cw * static_cast<size_t>(d->rounds), &d->round_counts) ||
        vrs_buffer_create(ctx, 256 * 4, &d->splitters))
        return cleanup(VRS_ERROR_OUT_OF_MEMORY, std::string("buffer allocation failed: ") + vrs_last_error(ctx));
    d->host_table.resize(static_cast<size_t>(world) * kRowWords);
    *out = d;
    return VRS_OK;
}

int vrs_dist_create(vrs_context ctx, void *nccl_comm, int rank, int world, uint32_t capacity_keys, int rounds, vrs_dist *out) {
    if (!out) return dfail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (world > 1 && !nccl_comm) return dfail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "an RCCL communicator is required at world size > 1");
    if (!nccl_comm) return vrs_dist_create_with_transport(ctx, nullptr, rank, world, capacity_keys, rounds, out);
    auto *ep = new (std::nothrow) RcclEndpoint();
    if (!ep) return dfail(nullptr, VRS_ERROR_OUT_OF_MEMORY, "host allocation failed");
    if (!load_rccl(ep->rccl)) {
        delete ep;
        return dfail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "no RCCL (librccl.so) could be bound in this process");
    }
    ep->comm = nccl_comm;
    const vrs_dist_transport t{ep, rccl_all_gather, rccl_all_reduce, rccl_group_start, rccl_send, rccl_recv, rccl_group_end, rccl_error_string};
    const int rc = vrs_dist_create_with_transport(ctx, &t, rank, world, capacity_keys, rounds, out);
    if (rc != VRS_OK) {
        delete ep;
        return rc;
    }
    (*out)->rccl = ep;
    return VRS_OK;
}

int vrs_dist_destroy(vrs_dist d) {
    if (!d) return VRS_OK;
    (void)hipSetDevice(d->device);
    if (d->comm_stream) (void)hipStreamSynchronize(d->comm_stream);
    for (vrs_buffer b : {d->grouped, d->recv, d->scratch, d->hist, d->row, d->table, d->counts, d->reduced, d->round_counts, d->splitters})
        if (b) (void)vrs_buffer_release(b);
    for (auto e : d->round_done)
        if (e) (void)hipEventDestroy(e);
    if (d->grouped_ready) (void)hipEventDestroy(d->grouped_ready);
    if (d->sorts_done) (void)hipEventDestroy(d->sorts_done);
    if (d->counts_ready) (void)hipEventDestroy(d->counts_ready);
    if (d->comm_stream) (void)hipStreamDestroy(d->comm_stream);
    delete d->rccl;  // the RCCL handle itself stays mapped: the process may hold communicators made by it
    delete d;
    return VRS_OK;
}

int vrs_dist_stats(vrs_dist d, uint64_t *hybrid_rounds, uint64_t *fallback_rounds, uint64_t *byte_shape_steps) {
    if (!d) return dfail(nullptr, VRS_ERROR_INVALID_ARGUMENT, "dist is NULL");
    if (hybrid_rounds) *hybrid_rounds = d->hybrid_rounds;
    if (fallback_rounds) *fallback_rounds = d->fallback_rounds;
    if (byte_shape_steps) *byte_shape_steps = d->byte_steps;
    return VRS_OK;
}

int vrs_dist_splitter_steps(vrs_dist d, uint64_t *splitter_steps) {
    if (!d || !splitter_steps) return dfail(d, VRS_ERROR_INVALID_ARGUMENT, "dist or splitter_steps is NULL");
    *splitter_steps = d->splitter_steps;
    return VRS_OK;
}

int vrs_dist_grouped_rounds(vrs_dist d, uint64_t *grouped_rounds) {
    if (!d || !grouped_rounds) return dfail(d, VRS_ERROR_INVALID_ARGUMENT, "dist or grouped_rounds is NULL");
    *grouped_rounds = d->grouped_rounds;
    return VRS_OK;
}

int vrs_dist_sort_keys_u32(vrs_dist d, vrs_buffer keys, uint32_t n, vrs_buffer *out_keys, uint32_t *out_count) {
    if (!d || !out_keys || !out_count) return dfail(d, VRS_ERROR_INVALID_ARGUMENT, "dist, out_keys or out_count is NULL");
    *out_keys = nullptr;
    *out_count = 0;
    vrs_context ctx = d->ctx;
    const int world = d->world, me = d->rank, R = d->rounds;
    VRS_DHIP(d, hipSetDevice(d->device));
    uint32_t *grouped = static_cast<uint32_t *>(vrs_buffer_device_ptr(d->grouped));
    uint32_t *recv = static_cast<uint32_t *>(vrs_buffer_device_ptr(d->recv));
    uint32_t *row = static_cast<uint32_t *>(vrs_buffer_device_ptr(d->row));
    uint32_t *table = static_cast<uint32_t *>(vrs_buffer_device_ptr(d->table));
    uint32_t *counts = static_cast<uint32_t *>(vrs_buffer_device_ptr(d->counts));
    uint32_t *reduced = static_cast<uint32_t *>(vrs_buffer_device_ptr(d->reduced));

    // What keeps THIS rank from taking part goes into its row; it still joins the collectives, and all ranks leave together.
    int my_status = VRS_OK;
    std::string my_error;
    if (n > d->capacity) {
        my_status = VRS_ERROR_INVALID_ARGUMENT;
        my_error = "shard larger than the capacity given at creation";
    } else if (n && !keys) {
        my_status = VRS_ERROR_INVALID_ARGUMENT;
        my_error = "keys is NULL";
    }
    const uint32_t n_eff = my_status == VRS_OK ? n : 0u;

    // the receive buffer may still be read by the sorts of the previous step's last round (they ran on the sort stream; one of
    // them may still owe its second half: enqueue-only sorts)
    VRS_D(d, vrs_sort_settle(ctx));
    VRS_DHIP(d, hipEventRecord(d->sorts_done, d->sort_stream));
    VRS_DHIP(d, hipStreamWaitEvent(d->comm_stream, d->sorts_done, 0));

    // 1. local step.  Hybrid shape: counting read + first MSD pass (the shard grouped by the top byte of its key range).
    //    The byte shape's contract partition pass, if the ranks settle on it, runs after the all-gather (the keys are
    //    untouched until then).
    // A total the hybrid shape cannot take (about 2e8 keys: every step of 8 x 1e8) is remembered: the next steps go straight to
    // the byte shape -- no counting read and first MSD pass for nothing, ONE all-gather instead of two -- and every 64th looks
    // again.  The memory is the same on every rank (it comes from the gathered table), so the ranks still issue the same
    // collectives.
    const bool byte_first = d->byte_shape_only || (d->too_large_for_hybrid && (d->steps % 64u) != 0u);
    d->steps++;
    const bool try_hybrid = !byte_first;
    bool partitioned = false;
    if (!byte_first) {
        VRS_DHIP(d, hipMemsetAsync(row, 0, kRowWords * 4, d->sort_stream));
        VRS_DHIP(d, hipMemsetAsync(counts, 0, static_cast<size_t>(VRS_MSD_COUNT_WORDS) * 4, d->sort_stream));
        // The collectives need the counts, not the partitioned keys: they run on the exchange stream from the moment the counts
        // are out (counts_ready), beside the first MSD pass, and the host waits for them only.
        if (try_hybrid && n_eff >= (1u << 16)) {
            const int rc = vrs_msd_partition_signal_u32(ctx, keys, d->grouped, d->counts, n_eff, d->counts_ready);
            if (rc == VRS_OK) {
                partitioned = true;
            } else {
                my_status = rc;
                my_error = std::string("vrs_msd_partition_u32: ") + vrs_last_error(ctx);
            }
        }
        if (!partitioned) VRS_DHIP(d, hipEventRecord(d->counts_ready, d->sort_stream));  // (the memsets above)
        VRS_DHIP(d, hipStreamWaitEvent(d->comm_stream, d->counts_ready, 0));
        if (partitioned) {
            // the row's slice counts = words [16384, 16384 + 2048) of the counts
            VRS_DHIP(d, hipMemcpyAsync(row, counts + 16384, kRowSlices * 4, hipMemcpyDeviceToDevice, d->comm_stream));
            VRS_DHIP(d, hipMemcpyAsync(row + kRowShift, counts + VRS_MSD_SHIFT_WORD, 2 * 4, hipMemcpyDeviceToDevice, d->comm_stream));  // shift word + range flag
        }
        d->host_row_tail[0] = my_status == VRS_OK ? n : 0u;
        d->host_row_tail[1] = static_cast<uint32_t>(my_status);
        d->host_row_tail[2] = d->capacity;
        VRS_DHIP(d, hipMemcpyAsync(row + kRowN, d->host_row_tail, 3 * 4, hipMemcpyHostToDevice, d->comm_stream));

        // 2. the collectives: every rank learns every rank's row; the bucket histograms are summed
        if (d->has_transport) {
            VRS_DTR(d, "all-gather of the shard rows", d->tr.all_gather(d->tr.user, row, table, kRowWords, d->comm_stream));
            if (try_hybrid) VRS_DTR(d, "all-reduce of the bucket histograms", d->tr.all_reduce(d->tr.user, counts, reduced, 16384, d->comm_stream));
        } else {
            VRS_DHIP(d, hipMemcpyAsync(table, row, kRowWords * 4, hipMemcpyDeviceToDevice, d->comm_stream));
            if (try_hybrid) VRS_DHIP(d, hipMemcpyAsync(reduced, counts, 16384 * 4, hipMemcpyDeviceToDevice, d->comm_stream));
        }
        VRS_DHIP(d, hipMemcpyAsync(d->host_table.data(), table, d->host_table.size() * 4, hipMemcpyDeviceToHost, d->comm_stream));
        VRS_DHIP(d, hipStreamSynchronize(d->comm_stream));  // the collectives; the first MSD pass may still be running
    }

    // ---- from here on every rank holds the same table: every decision below is the same on all of them
    uint32_t min_capacity = 0xFFFFFFFFu;
    uint64_t grand_total = 0;
    bool hybrid = try_hybrid;
    uint32_t shift = 0xFFFFFFFFu;  // of the first non-empty shard; empty shards have nothing to say (and nothing to send)
    uint64_t bucket_expect = 0;
    if (!byte_first) {
        for (int q = 0; q < world; ++q) {
            const uint32_t *r = &d->host_table[static_cast<size_t>(q) * kRowWords];
            if (r[kRowStatus] != 0u) {
                if (q == me) return dfail(d, my_status, my_error);
                return dfail(d, VRS_ERROR_PEER, "rank " + std::to_string(q) + " could not take part in the step (its status " + std::to_string(r[kRowStatus]) + ")");
            }
            min_capacity = std::min(min_capacity, r[kRowCapacity]);
            grand_total += r[kRowN];
        }
        // hybrid shape iff every non-empty shard was partitioned with the same bucket shift of a 27..32-bit key range and no key
        // of any shard lies above the range its rank probed
        for (int q = 0; q < world && hybrid; ++q) {
            const uint32_t *r = &d->host_table[static_cast<size_t>(q) * kRowWords];
            if (r[kRowN] == 0u) continue;
            if (shift == 0xFFFFFFFFu) shift = r[kRowShift];
            if (r[kRowN] < (1u << 16) || r[kRowShift] != shift || r[kRowOver] != 0u) hybrid = false;  // too small to have been partitioned, another key range, a stray key
        }
        if (hybrid && (shift < 13u || shift > 18u)) hybrid = false;  // (no keys at all: 0xFFFFFFFF)
        // the hybrid shape's buckets are the top 14 bits of the GLOBAL key range: N_total / 16384 keys each, whatever the number
        // of ranks -- they must fit the local sort (14333 keys: about 2e8 keys in total).  Larger totals take the byte shape,
        // whose per-range sorts bucket each received sub-range on its own (vrs_sort_keys_u32_ranged).
        bucket_expect = grand_total / 16384u + grand_total / 16384u / 8u + 64u;
        d->too_large_for_hybrid = bucket_expect > d->hybrid_max_bucket;
        if (hybrid && d->too_large_for_hybrid) hybrid = false;
    }
    (void)partitioned;

    // top-byte counts per rank (hybrid shape: sums over the eight slices; byte shape: filled in below) and their prefixes
    std::vector<std::vector<uint64_t>> base(static_cast<size_t>(world), std::vector<uint64_t>(257, 0));
    uint64_t byte_counts[256] = {};
    if (!hybrid) {
        // byte shape: contract partition pass by the top byte now (12 B/key), then an all-gather of the real rows (the only one
        // of a step that went straight here)
        d->byte_steps++;
        uint32_t *prefix = row;  // row words [0, 256): exclusive prefix of my top-byte counts; [256]: my shard size, [257] status, [258] capacity
        // the all-gather needs the prefix of my top-byte counts, not the partitioned keys: the sort stage hands the prefix out (and
        // signals counts_ready) before its scatter kernel, and the collective runs on the exchange stream beside that kernel
        bool signalled = false;
        if (n_eff) {
            vrs_push_constants pc{n_eff, 24, vrs_workgroup_count(n_eff, 32), 32};
            int rc = vrs_multi_radixsort_histograms(ctx, keys, d->hist, &pc);
            if (rc == VRS_OK) {
                vrs_buffer pv = nullptr;
                rc = vrs_buffer_wrap(ctx, prefix, 256 * 4, &pv);
                if (rc == VRS_OK) rc = vrs_multi_radixsort_offsets_hook(ctx, pv, d->counts_ready);
                if (rc == VRS_OK) {
                    rc = vrs_multi_radixsort(ctx, keys, d->grouped, d->hist, &pc);
                    signalled = rc == VRS_OK;
                    if (!signalled) (void)vrs_multi_radixsort_offsets_hook(ctx, nullptr, nullptr);  // (a stage that failed before its prefix)
                }
                if (pv) (void)vrs_buffer_release(pv);
            }
            if (rc != VRS_OK) {
                my_status = rc;
                my_error = std::string("top-byte partition pass: ") + vrs_last_error(ctx);
            }
        } else {
            VRS_DHIP(d, hipMemsetAsync(prefix, 0, 256 * 4, d->sort_stream));
        }
        if (!signalled) VRS_DHIP(d, hipEventRecord(d->counts_ready, d->sort_stream));
        VRS_DHIP(d, hipStreamWaitEvent(d->comm_stream, d->counts_ready, 0));
        constexpr size_t kByteRow = 259;
        d->host_row_tail[0] = my_status == VRS_OK ? n_eff : 0u;
        d->host_row_tail[1] = static_cast<uint32_t>(my_status);
        d->host_row_tail[2] = d->capacity;
        VRS_DHIP(d, hipMemcpyAsync(row + 256, d->host_row_tail, 3 * 4, hipMemcpyHostToDevice, d->comm_stream));
        if (d->has_transport) {
            VRS_DTR(d, "all-gather of the top-byte prefixes", d->tr.all_gather(d->tr.user, row, table, kByteRow, d->comm_stream));
        } else {
            VRS_DHIP(d, hipMemcpyAsync(table, row, kByteRow * 4, hipMemcpyDeviceToDevice, d->comm_stream));
        }
        VRS_DHIP(d, hipMemcpyAsync(d->host_table.data(), table, static_cast<size_t>(world) * kByteRow * 4, hipMemcpyDeviceToHost, d->comm_stream));
        VRS_DHIP(d, hipStreamSynchronize(d->comm_stream));  // the collective; the scatter may still be running
        min_capacity = 0xFFFFFFFFu;
        grand_total = 0;
        for (int q = 0; q < world; ++q) {
            const uint32_t *r = &d->host_table[static_cast<size_t>(q) * kByteRow];
            if (r[257] != 0u) {
                if (q == me) return dfail(d, my_status, my_error);
                return dfail(d, VRS_ERROR_PEER, "rank " + std::to_string(q) + " could not take part in the step or failed in its partition pass (status " + std::to_string(r[257]) + ")");
            }
            min_capacity = std::min(min_capacity, r[258]);
            grand_total += r[256];
            for (int t = 0; t < 256; ++t) base[static_cast<size_t>(q)][static_cast<size_t>(t)] = r[t];
            base[static_cast<size_t>(q)][256] = r[256];
            for (int t = 0; t < 256; ++t) byte_counts[t] += base[static_cast<size_t>(q)][static_cast<size_t>(t) + 1] - base[static_cast<size_t>(q)][static_cast<size_t>(t)];
        }
    } else {
        for (int q = 0; q < world; ++q) {
            const uint32_t *r = &d->host_table[static_cast<size_t>(q) * kRowWords];
            uint64_t run = 0;
            for (int t = 0; t < 256; ++t) {
                base[static_cast<size_t>(q)][static_cast<size_t>(t)] = run;
                uint64_t c = 0;
                for (int g = 0; g < 8; ++g) c += r[g * 256 + t];
                run += c;
                byte_counts[t] += c;
            }
            base[static_cast<size_t>(q)][256] = run;  // == the shard size
        }
    }

    const int P = world * R;  // part q * R + r = rank q, round r
    std::vector<uint32_t> parts(static_cast<size_t>(P) + 1);
    int rc = vrs_dist_plan_splitters(byte_counts, P, parts.data());
    if (rc) return rc;
    // how many keys every rank receives; byte-aligned cuts cannot balance keys whose top bytes are too concentrated.
    // The same verdict on every rank: the gathered capacities, not this rank's own, decide.
    uint64_t worst = 0;
    for (int q = 0; q < world; ++q) {
        uint64_t s = 0;
        for (uint32_t t = parts[static_cast<size_t>(q) * R]; t < parts[static_cast<size_t>(q + 1) * R]; ++t) s += byte_counts[t];
        worst = std::max(worst, s);
    }
    const double ideal = std::max(static_cast<double>(grand_total) / world, 1.0);
    // first key value of every part: what its ranged sort may assume of its keys (byte-aligned parts: the first top byte)
    std::vector<uint32_t> floors(static_cast<size_t>(P));
    for (int p = 0; p < P; ++p) floors[static_cast<size_t>(p)] = std::min<uint32_t>(parts[static_cast<size_t>(p)], 255u) << 24;
    bool by_splitters = false;
    if (static_cast<double>(worst) > d->max_imbalance * ideal || worst > min_capacity) {
        if (d->no_sampled_splitters)
            return dfail(d, VRS_ERROR_UNBALANCED, "top bytes too concentrated for byte-aligned key ranges (small or clustered keys) and VRS_DIST_SAMPLED_SPLITTERS=0");
        // Sampled splitters (the same decision on every rank: it depends on the gathered table only).  Every rank adds 2048 keys
        // of its shard, taken at evenly spaced positions, to a pool (one all-gather of 8 KiB rows); the P - 1 cut keys are the
        // pool's quantiles, every sample weighing (its shard's size / 2048) keys; the shard -- the caller's buffer is still
        // untouched -- is grouped again, by range this time (vrs_range_partition: range = number of cut keys <= key, one stable
        // 12 B/key pass), and a third all-gather hands out the range prefixes.  Every part is then one message per source and
        // one ranged sort.  Only keys with massive ties (one value holding more than a rank's share) still cannot be balanced.
        by_splitters = true;
        hybrid = false;
        d->splitter_steps++;
        std::vector<uint64_t> shard(static_cast<size_t>(world));
        for (int q = 0; q < world; ++q) shard[static_cast<size_t>(q)] = base[static_cast<size_t>(q)][256];
        const uint32_t S = kSamplesPerRank;
        hipLaunchKernelGGL(sample_shard_kernel, dim3(S / 256u), dim3(256), 0, d->sort_stream, static_cast<const uint32_t *>(n_eff ? vrs_buffer_device_ptr(keys) : nullptr), n_eff, row, S);
        VRS_DHIP(d, hipGetLastError());
        VRS_DHIP(d, hipEventRecord(d->counts_ready, d->sort_stream));
        VRS_DHIP(d, hipStreamWaitEvent(d->comm_stream, d->counts_ready, 0));
        if (d->has_transport) {
            VRS_DTR(d, "all-gather of the key samples", d->tr.all_gather(d->tr.user, row, table, S, d->comm_stream));
        } else {
            VRS_DHIP(d, hipMemcpyAsync(table, row, S * 4, hipMemcpyDeviceToDevice, d->comm_stream));
        }
        VRS_DHIP(d, hipMemcpyAsync(d->host_table.data(), table, static_cast<size_t>(world) * S * 4, hipMemcpyDeviceToHost, d->comm_stream));
        VRS_DHIP(d, hipStreamSynchronize(d->comm_stream));
        if ((rc = vrs_dist_plan_sampled_splitters(d->host_table.data(), shard.data(), world, S, P, d->host_splitters))) return rc;
        if (P > 1) VRS_DHIP(d, hipMemcpyAsync(vrs_buffer_device_ptr(d->splitters), d->host_splitters, static_cast<size_t>(P - 1) * 4, hipMemcpyHostToDevice, d->sort_stream));
        uint32_t *prefix = row;  // as in the byte shape: [0, 256) exclusive prefix of my range counts, [256] shard size, [257] status, [258] capacity
        if (n_eff) {
            rc = vrs_range_partition(ctx, keys, d->grouped, d->splitters, static_cast<uint32_t>(P - 1), n_eff);
            if (rc == VRS_OK) {
                vrs_buffer pv = nullptr;
                rc = vrs_buffer_wrap(ctx, prefix, 256 * 4, &pv);
                if (rc == VRS_OK) rc = vrs_multi_radixsort_digit_offsets_device(ctx, pv);
                if (pv) (void)vrs_buffer_release(pv);
            }
            if (rc != VRS_OK) {
                my_status = rc;
                my_error = std::string("range partition pass: ") + vrs_last_error(ctx);
            }
        } else {
            VRS_DHIP(d, hipMemsetAsync(prefix, 0, 256 * 4, d->sort_stream));
        }
        VRS_DHIP(d, hipEventRecord(d->counts_ready, d->sort_stream));
        VRS_DHIP(d, hipStreamWaitEvent(d->comm_stream, d->counts_ready, 0));
        constexpr size_t kByteRow = 259;
        d->host_row_tail[0] = my_status == VRS_OK ? n_eff : 0u;
        d->host_row_tail[1] = static_cast<uint32_t>(my_status);
        d->host_row_tail[2] = d->capacity;
        VRS_DHIP(d, hipMemcpyAsync(row + 256, d->host_row_tail, 3 * 4, hipMemcpyHostToDevice, d->comm_stream));
        if (d->has_transport) {
            VRS_DTR(d, "all-gather of the range prefixes", d->tr.all_gather(d->tr.user, row, table, kByteRow, d->comm_stream));
        } else {
            VRS_DHIP(d, hipMemcpyAsync(table, row, kByteRow * 4, hipMemcpyDeviceToDevice, d->comm_stream));
        }
        VRS_DHIP(d, hipMemcpyAsync(d->host_table.data(), table, static_cast<size_t>(world) * kByteRow * 4, hipMemcpyDeviceToHost, d->comm_stream));
        VRS_DHIP(d, hipStreamSynchronize(d->comm_stream));
        std::vector<uint64_t> part_counts(static_cast<size_t>(P), 0);
        for (int q = 0; q < world; ++q) {
            const uint32_t *r = &d->host_table[static_cast<size_t>(q) * kByteRow];
            if (r[257] != 0u) {
                if (q == me) return dfail(d, my_status, my_error);
                return dfail(d, VRS_ERROR_PEER, "rank " + std::to_string(q) + " failed in its range partition pass (status " + std::to_string(r[257]) + ")");
            }
            // the prefix of ranges [0, P): entry P (no such range: every later entry of the 256 holds the shard size) closes it
            std::vector<uint64_t> &bq = base[static_cast<size_t>(q)];
            for (int t = 0; t < P; ++t) bq[static_cast<size_t>(t)] = r[t];
            bq[static_cast<size_t>(P)] = r[256];
            for (int t = 0; t < P; ++t) part_counts[static_cast<size_t>(t)] += bq[static_cast<size_t>(t) + 1] - bq[static_cast<size_t>(t)];
        }
        for (int p = 0; p <= P; ++p) parts[static_cast<size_t>(p)] = static_cast<uint32_t>(p);  // part p IS range p
        for (int p = 0; p < P; ++p) floors[static_cast<size_t>(p)] = p ? d->host_splitters[p - 1] : 0u;
        worst = 0;
        for (int q = 0; q < world; ++q) {
            uint64_t s = 0;
            for (int t = q * R; t < (q + 1) * R; ++t) s += part_counts[static_cast<size_t>(t)];
            worst = std::max(worst, s);
        }
        if (static_cast<double>(worst) > d->max_imbalance * ideal || worst > min_capacity)
            return dfail(d, VRS_ERROR_UNBALANCED,
                         "too many equal keys: no cut between key VALUES gives every rank at most 15 % over the even share (and no more than "
                         "the smallest capacity of all ranks)");
    }

    // Byte shape: if no top byte holds more keys than its 256 sub-buckets can take (the same verdict on every rank: the summed
    // top-byte counts decide), the keys also land grouped by top byte and every round is finished by ONE counting read, the
    // second MSD pass by the next 8 bits and the local sort (vrs_msd_finish_grouped_u32: 20 B/key instead of a whole ranged sort's 28).
    bool grouped_finish = false;
    if (!hybrid && !by_splitters && !d->no_grouped_finish) {
        uint64_t fullest = 0;
        for (int t = 0; t < 256; ++t) fullest = std::max(fullest, byte_counts[t]);
        grouped_finish = fullest <= 256u * 12500u && grand_total >= (1u << 16);
    }
    const bool by_top_byte = hybrid || grouped_finish;  // one message per (top byte, source), the result built in the scratch buffer
    const char *pf_env = std::getenv("VRS_DIST_POOL_FINISH");
    const bool pool_finish = !(pf_env && pf_env[0] == '0');
    // Byte shape finished by the pool form's second half: the keys this rank keeps for itself are NOT copied beside the received ones --
    // they stay where the partition pass wrote them (the grouped buffer) and the finish reads every top byte as two pieces
    // (vrs_msd_finish_grouped_split_u32): 1 / world of the exchange's 8 bytes per key never moves.  The receive buffer keeps the layout
    // it would have had, with every top byte's own part LAST: a hole, filled only if the round has to be sorted another way.
    // (VRS_DIST_COPY_OWN=1: the copies as before)
    const char *co_env = std::getenv("VRS_DIST_COPY_OWN");
    const bool own_in_place = grouped_finish && pool_finish && !(co_env && co_env[0] == '1');
    struct OwnHole { uint64_t dst, src, len; };
    std::vector<std::vector<OwnHole>> holes(static_cast<size_t>(R));

    // what I receive in round r: one message per (top byte, source) in that order -- the round's keys land grouped by top byte --
    // or (byte shape with whole ranged sorts) one message per source.  round_off: where round r starts in the receive buffer.
    std::vector<uint64_t> round_off(static_cast<size_t>(R) + 1, 0);
    for (int r = 0; r < R; ++r) {
        const uint32_t lo = parts[static_cast<size_t>(me) * R + r], hi = parts[static_cast<size_t>(me) * R + r + 1];
        uint64_t tot = 0;
        for (int s = 0; s < world; ++s) tot += base[static_cast<size_t>(s)][hi] - base[static_cast<size_t>(s)][lo];
        round_off[static_cast<size_t>(r) + 1] = round_off[static_cast<size_t>(r)] + tot;
    }
    const uint64_t total = round_off[static_cast<size_t>(R)];  // <= worst <= min_capacity <= my capacity

    // 3. exchange, round by round on the exchange stream: grouped send/recv with every peer; my own slices are device copies
    VRS_DHIP(d, hipEventRecord(d->grouped_ready, d->sort_stream));
    VRS_DHIP(d, hipStreamWaitEvent(d->comm_stream, d->grouped_ready, 0));
    for (int r = 0; r < R; ++r) {
        int failed = 0;
        std::string what;
        const auto tr = [&](const char *w, int code) {
            if (code != 0 && failed == 0) {
                failed = code;
                what = w;
            }
        };
        if (d->has_transport) tr("group start", d->tr.group_start(d->tr.user));
        const uint32_t lo = parts[static_cast<size_t>(me) * R + r], hi = parts[static_cast<size_t>(me) * R + r + 1];
        uint64_t off = round_off[static_cast<size_t>(r)];
        hipError_t he = hipSuccess;
        // my own slices are device copies; neighbouring ones are merged (a copy of a megabyte is launch-bound: at world size 1
        // the 256 top bytes of the hybrid shape would be 256 copies of 5 us each instead of one of 0.15 ms)
        uint64_t run_src = 0, run_dst = 0, run_len = 0;
        const auto flush_own = [&] {
            if (run_len && he == hipSuccess)
                he = hipMemcpyAsync(recv + run_dst, grouped + run_src, run_len * 4, hipMemcpyDeviceToDevice, d->comm_stream);
            run_len = 0;
        };
        const auto land = [&](int s, uint32_t t0, uint32_t t1) {  // source s's keys with top bytes [t0, t1) land at `off`
            const uint64_t a = base[static_cast<size_t>(s)][t0], b = base[static_cast<size_t>(s)][t1];
            if (b > a) {
                if (s == me) {
                    if (run_len && run_src + run_len == a && run_dst + run_len == off) {
                        run_len += b - a;
                    } else {
                        flush_own();
                        run_src = a;
                        run_dst = off;
                        run_len = b - a;
                    }
                } else if (failed == 0) {
                    tr("recv", d->tr.recv(d->tr.user, recv + off, b - a, s, d->comm_stream));
                }
            }
            off += b - a;
        };
        if (own_in_place) {
            for (uint32_t t = lo; t < hi; ++t) {
                for (int s = 0; s < world; ++s)
                    if (s != me) land(s, t, t + 1);
                const uint64_t a = base[static_cast<size_t>(me)][t], b = base[static_cast<size_t>(me)][t + 1];
                if (b > a) {  // (neighbouring holes whose sources follow each other too are one copy, should it come to that)
                    std::vector<OwnHole> &h = holes[static_cast<size_t>(r)];
                    if (!h.empty() && h.back().dst + h.back().len == off && h.back().src + h.back().len == a) h.back().len += b - a;
                    else h.push_back(OwnHole{off, a, b - a});
                }
                off += b - a;
            }
        } else if (by_top_byte) {
            for (uint32_t t = lo; t < hi; ++t)
                for (int s = 0; s < world; ++s) land(s, t, t + 1);
        } else {
            for (int s = 0; s < world; ++s) land(s, lo, hi);
        }
        flush_own();
        for (int dst = 0; dst < world && failed == 0; ++dst) {
            if (dst == me) continue;
            const uint32_t dlo = parts[static_cast<size_t>(dst) * R + r], dhi = parts[static_cast<size_t>(dst) * R + r + 1];
            if (by_top_byte) {
                for (uint32_t t = dlo; t < dhi && failed == 0; ++t) {
                    const uint64_t a = base[static_cast<size_t>(me)][t], b = base[static_cast<size_t>(me)][t + 1];
                    if (b > a) tr("send", d->tr.send(d->tr.user, grouped + a, b - a, dst, d->comm_stream));
                }
            } else {
                const uint64_t a = base[static_cast<size_t>(me)][dlo], b = base[static_cast<size_t>(me)][dhi];
                if (b > a) tr("send", d->tr.send(d->tr.user, grouped + a, b - a, dst, d->comm_stream));
            }
        }
        // a group that was opened is closed whatever happened inside it
        if (d->has_transport) tr("group end", d->tr.group_end(d->tr.user));
        if (failed != 0) return dfail(d, VRS_ERROR_HIP, transport_error(d, ("exchange round " + std::to_string(r) + ", " + what).c_str(), failed));
        if (he != hipSuccess) return dfail(d, VRS_ERROR_HIP, std::string("device copy of this rank's own slice: ") + hipGetErrorString(he));
        VRS_DHIP(d, hipEventRecord(d->round_done[static_cast<size_t>(r)], d->comm_stream));
    }

    // 4. round r's keys are finished while the later rounds are still on the wire; the sub-ranges are disjoint and ascending,
    //    so their concatenation is the sorted range: nothing to merge.  Hybrid shape: the result is built in the scratch
    //    buffer (second MSD pass: receive buffer -> scratch; local sort in place there), at the round's own offset -- the
    //    step's output then IS the scratch buffer.
    uint32_t *scratch = static_cast<uint32_t *>(vrs_buffer_device_ptr(d->scratch));
    uint32_t *round_counts = static_cast<uint32_t *>(vrs_buffer_device_ptr(d->round_counts));
    if (hybrid) {
        RoundCuts cuts{};
        for (int r = 0; r < R; ++r) {
            cuts.lo[r] = parts[static_cast<size_t>(me) * R + r];
            cuts.hi[r] = parts[static_cast<size_t>(me) * R + r + 1];
        }
        cuts.shift = shift;
        hipLaunchKernelGGL(round_tables_kernel, dim3(static_cast<uint32_t>(R)), dim3(1024), 0, d->sort_stream, reduced, round_counts, cuts);
        VRS_DHIP(d, hipGetLastError());
    }
    // one round sorted with vrs_sort_keys_u32_ranged: every key of the round is >= its first top byte << 24, so the sort buckets the
    // sub-range as if it were a whole key range (hybrid shape: the step's output is the scratch buffer)
    const auto ranged_round = [&](int r, uint64_t cnt) -> int {
        // (own keys left in place: into their holes first -- on the sort stream, behind the round's exchange)
        for (const OwnHole &h : holes[static_cast<size_t>(r)])
            if (hipMemcpyAsync(recv + h.dst, grouped + h.src, h.len * 4, hipMemcpyDeviceToDevice, d->sort_stream) != hipSuccess)
                return VRS_ERROR_HIP;
        holes[static_cast<size_t>(r)].clear();
        vrs_buffer view = nullptr, sview = nullptr;
        int e = vrs_buffer_wrap(ctx, recv + round_off[static_cast<size_t>(r)], cnt * 4, &view);
        if (e == VRS_OK) e = vrs_buffer_wrap(ctx, scratch + round_off[static_cast<size_t>(r)], cnt * 4, &sview);
        if (e == VRS_OK) e = vrs_sort_keys_u32_ranged(ctx, view, sview, static_cast<uint32_t>(cnt), floors[static_cast<size_t>(me) * R + r]);
        if (e == VRS_OK && by_top_byte) e = vrs_buffer_copy(ctx, sview, view, cnt * 4);
        if (view) (void)vrs_buffer_release(view);
        if (sview) (void)vrs_buffer_release(sview);
        return e;
    };
    // Hybrid shape: every round's second half is enqueued before any of their plans is looked at -- a host wait per round would
    // leave the GPU idle while the next round is enqueued -- and a round its plan refused (its kernels left at once, the keys are
    // still in the receive buffer) is sorted whole afterwards.
    std::vector<uint32_t> ticket(static_cast<size_t>(R), 0u);
    for (int r = 0; r < R; ++r) {
        const uint64_t cnt = round_off[static_cast<size_t>(r) + 1] - round_off[static_cast<size_t>(r)];
        // (an enqueue-only sort of the round before may still owe its second half: that goes on the stream BEFORE the wait for this
        // round's keys, or it would be ordered behind the exchange it was meant to overlap)
        if ((rc = vrs_sort_settle(ctx))) return dfail_ctx(d, rc, "vrs_sort_settle (the round before)");
        VRS_DHIP(d, hipStreamWaitEvent(d->sort_stream, d->round_done[static_cast<size_t>(r)], 0));
        if (!cnt) continue;
        if (by_top_byte && cnt >= (1u << 16)) {
            vrs_buffer view = nullptr, sview = nullptr;
            VRS_D(d, vrs_buffer_wrap(ctx, recv + round_off[static_cast<size_t>(r)], cnt * 4, &view));
            rc = vrs_buffer_wrap(ctx, scratch + round_off[static_cast<size_t>(r)], cnt * 4, &sview);
            if (rc != VRS_OK) {
                (void)vrs_buffer_release(view);
                return dfail_ctx(d, rc, "vrs_buffer_wrap");
            }
            if (hybrid) {
                // the bucket histogram of exactly this round's keys: table r (the plan kernel reads it in place)
                vrs_buffer table = nullptr;
                rc = vrs_buffer_wrap(ctx, round_counts + static_cast<size_t>(r) * VRS_MSD_COUNT_WORDS, static_cast<size_t>(VRS_MSD_COUNT_WORDS) * 4, &table);
                if (rc == VRS_OK) rc = vrs_msd_finish_u32(ctx, view, sview, table, static_cast<uint32_t>(cnt), static_cast<uint32_t>(bucket_expect));
                if (table) (void)vrs_buffer_release(table);
            } else {
                const uint32_t lo = parts[static_cast<size_t>(me) * R + r], hi = parts[static_cast<size_t>(me) * R + r + 1];
                // every top byte's keys landed in one piece and every rank knows how many there are (the gathered table): nothing
                // needs to be read to be counted -- the pool form's second half, 16 bytes per key instead of 20
                // (VRS_DIST_POOL_FINISH=0: the counted finish)
                uint32_t per_byte[256], own_byte[256];
                for (uint32_t t = lo; t < hi; ++t) {
                    per_byte[t - lo] = static_cast<uint32_t>(byte_counts[t]);
                    own_byte[t - lo] = static_cast<uint32_t>(base[static_cast<size_t>(me)][t + 1] - base[static_cast<size_t>(me)][t]);
                }
                if (own_in_place)  // (this round's own keys: the grouped buffer from the first of its top bytes on)
                    rc = vrs_msd_finish_grouped_split_u32(ctx, view, d->grouped, base[static_cast<size_t>(me)][lo], sview, static_cast<uint32_t>(cnt), lo,
                                                          hi - lo, per_byte, own_byte);
                else
                    rc = pool_finish ? vrs_msd_finish_grouped_counts_u32(ctx, view, sview, static_cast<uint32_t>(cnt), lo, hi - lo, per_byte)
                                     : vrs_msd_finish_grouped_u32(ctx, view, sview, static_cast<uint32_t>(cnt), lo, hi - lo);
            }
            if (rc == VRS_OK) rc = vrs_msd_finish_ticket(ctx, &ticket[static_cast<size_t>(r)]);
            (void)vrs_buffer_release(view);
            (void)vrs_buffer_release(sview);
            if (rc != VRS_OK) return dfail_ctx(d, rc, hybrid ? "vrs_msd_finish_u32 (received sub-range)" : "vrs_msd_finish_grouped_u32 (received sub-range)");
        } else {
            if ((rc = ranged_round(r, cnt))) return dfail_ctx(d, rc, "vrs_sort_keys_u32_ranged (received sub-range)");
        }
    }
    for (int r = 0; r < R; ++r) {
        if (!ticket[static_cast<size_t>(r)]) continue;
        int took = 0;
        // waits for that round's plan (the round has landed by then), never for its sort
        if ((rc = vrs_msd_finish_status_at(ctx, ticket[static_cast<size_t>(r)], &took))) return dfail_ctx(d, rc, "vrs_msd_finish_status_at");
        if (took) {
            if (hybrid) d->hybrid_rounds++;
            else d->grouped_rounds++;
            continue;
        }
        d->fallback_rounds++;
        const uint64_t cnt = round_off[static_cast<size_t>(r) + 1] - round_off[static_cast<size_t>(r)];
        if ((rc = ranged_round(r, cnt))) return dfail_ctx(d, rc, "vrs_sort_keys_u32_ranged (a round the hybrid form refused)");
    }
    *out_keys = by_top_byte ? d->scratch : d->recv;
    *out_count = static_cast<uint32_t>(total);
    return VRS_OK;
}

}  // extern "C"
