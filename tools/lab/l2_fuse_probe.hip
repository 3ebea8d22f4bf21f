// lab: can a producer phase and a consumer phase meet in ONE XCD's 4 MB L2?
// 256 units of 1.5 MB (a top-byte's worth of 10^8 keys).  Every XCD works through its own queue of units with persistent
// workgroups: B items (48 tiles of 32 KB per unit: read from A, written SCATTERED over the unit's 64 sub-ranges in B, like the
// second MSD pass) and L items (64 sub-ranges of 24 KB: read + rewritten in place, like the local sort).  An L item waits until
// all B items of its unit have finished.  `lag` = how many units of B items are queued between B(j) and L(j):
//   lag 0 : B(0) L(0) B(1) L(1) ...        lag 1 : B(0) B(1) L(0) B(2) L(1) ...       lag 1000 : all B, then all L (= two kernels)
// build: hipcc --offload-arch=gfx950 -O3 tools/lab/l2_fuse_probe.hip -o tools/lab/libs/l2_fuse_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr uint32_t kTile = 32768, kSub = 24576, kSubStride = 24576 + 256;  // bytes; the stride is an odd number of 256-byte blocks: regular
// sub-range bases would land every piece of a tile in one L2 channel, which the sort's data-dependent bucket bases do not

struct Queue { uint32_t ticket; uint32_t pad0[31]; uint32_t done[64]; uint32_t pad[32]; };  // per XCD

__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3u << 11) | 20u); }

// item t of an XCD's queue -> (unit j of this XCD, kind, index); units_x units per XCD, T tiles and S sub-ranges per unit
__device__ bool decode(uint32_t t, uint32_t units_x, uint32_t T, uint32_t S, uint32_t lag, uint32_t &j, bool &is_l, uint32_t &idx) {
    if (lag == 3000u) {  // only the L items, no waiting
        if (t < units_x * S) { j = t / S; idx = t % S; is_l = true; return true; }
        return false;
    }
    if (lag >= units_x) {  // all B, then all L (2000: only the B items)
        if (lag == 2000u && t >= units_x * T) return false;
        if (t < units_x * T) { j = t / T; idx = t % T; is_l = false; return true; }
        t -= units_x * T;
        if (t < units_x * S) { j = t / S; idx = t % S; is_l = true; return true; }
        return false;
    }
    // rounds r = 0 .. units_x + lag - 1: B(r) if r < units_x, then L(r - lag) if r >= lag
    const uint32_t per = T + S;
    // the first `lag` rounds hold only B items
    if (t < lag * T) { j = t / T; idx = t % T; is_l = false; return true; }
    t -= lag * T;
    const uint32_t full = units_x - lag;  // rounds with both
    if (t < full * per) {
        const uint32_t r = t / per, o = t % per;
        if (o < T) { j = r + lag; idx = o; is_l = false; } else { j = r; idx = o - T; is_l = true; }
        return true;
    }
    t -= full * per;
    if (t < lag * S) { j = full + t / S; idx = t % S; is_l = true; return true; }
    return false;
}

template <bool NT, bool STATIC>
__global__ __launch_bounds__(256) void fused_kernel(const uint4 *a, uint4 *b, Queue *queues, uint32_t units, uint32_t T, uint32_t S,
                                                    uint32_t lag, uint32_t unit_bytes) {
    __shared__ uint32_t s_t;
    const uint32_t x = xcc_id() & 7u;
    Queue *q = queues + x;
    const uint32_t units_x = units / 8;
    uint32_t mine = 0;
    const uint32_t step = gridDim.x / 8;  // STATIC: workgroups per XCD (the lab launches a grid the chip holds at once)
    for (uint32_t round = 0;; ++round) {
        // workgroup scope: the atomic runs in THIS XCD's L2 (an agent-scope one is performed memory-side)
        if (STATIC && round) { mine += step; if (threadIdx.x == 0) s_t = mine; }
        else if (threadIdx.x == 0) s_t = __hip_atomic_fetch_add(&q->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        const uint32_t t = s_t;
        mine = t;
        __syncthreads();
        uint32_t j, idx; bool is_l;
        if (!decode(t, units_x, T, S, lag, j, is_l, idx)) return;
        const size_t unit = (static_cast<size_t>(j) * 8 + x) * unit_bytes / 16;  // in uint4 (A)
        const size_t unit_b = (static_cast<size_t>(j) * 8 + x) * (static_cast<size_t>(S) * kSubStride) / 16;
        if (!is_l) {
            const uint4 *src = a + unit + static_cast<size_t>(idx) * (kTile / 16);
            uint4 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (NT) {
                    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                    const u4 w = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(src + r * 256 + threadIdx.x));
                    v[r] = make_uint4(w.x, w.y, w.z, w.w);
                }
                else v[r] = src[r * 256 + threadIdx.x];
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint32_t p = (r * 256 + threadIdx.x) * 16;     // byte in the tile
                const uint32_t piece = kTile / S;                    // bytes of this tile per sub-range (512)
                const uint32_t s = p / piece, within = p % piece;
                b[unit_b + (static_cast<size_t>(s) * kSubStride + idx * piece + within) / 16] = v[r];
            }
            // NOT an agent-scope release (that writes the whole L2 back): the consumers sit behind the same L2, so the stores only
            // have to have reached it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add(&q->done[j], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            if (threadIdx.x == 0) {
                uint32_t spins = 0;
                while (lag != 3000u && __hip_atomic_load(&q->done[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < T) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > 4000000u) { atomicAdd(&q->pad[0], 1u); break; }  // lab: never hang the box
                }
            }
            __syncthreads();
            // NOT an agent-scope acquire either (buffer_inv sc1 also drops the L2's lines: 505 us for the L items alone)
            uint4 *p = b + unit_b + static_cast<size_t>(idx) * (kSubStride / 16);
            uint4 v[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) v[r] = p[r * 256 + threadIdx.x];
#pragma unroll
            for (int r = 0; r < 6; ++r) { v[r].x += 1u; v[r].y ^= v[r].x; v[r].z += v[r].y; v[r].w ^= v[r].z; p[r * 256 + threadIdx.x] = v[r]; }
        }
    }
}

int main(int argc, char **argv) {
    const uint32_t units = 256, T = 48, S = 64, unit_bytes = T * kTile;  // 1.5 MB
    const size_t total = static_cast<size_t>(units) * unit_bytes;        // 384 MB
    uint4 *a, *b[2]; Queue *q;
    CK(hipMalloc(&a, total)); const size_t total_b = static_cast<size_t>(units) * S * kSubStride;
    CK(hipMalloc(&b[0], total_b)); CK(hipMalloc(&b[1], total_b)); CK(hipMalloc(&q, 8 * sizeof(Queue)));
    CK(hipMemset(a, 1, total)); CK(hipMemset(b[0], 2, total_b)); CK(hipMemset(b[1], 2, total_b));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("384 MB in 256 units of 1.5 MB; B = read A + scattered write into B, L = in-place update of B: 16 B/key of kernel traffic\n");
    for (int nt = 0; nt < 1; ++nt)
    for (int wgs_per_cu : {4, 8})
    for (uint32_t lag : {0u, 1u, 1000u, 2000u, 3000u}) {
        float best = 1e9f, ms;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemsetAsync(q, 0, 8 * sizeof(Queue)));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            if (nt) fused_kernel<false, true><<<256 * wgs_per_cu, 256>>>(a, b[rep & 1], q, units, T, S, lag, unit_bytes);
            else fused_kernel<false, false><<<256 * wgs_per_cu, 256>>>(a, b[rep & 1], q, units, T, S, lag, unit_bytes);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        Queue h[8];
        CK(hipMemcpy(h, q, sizeof(h), hipMemcpyDeviceToHost));
        uint32_t gave_up = 0;
        for (int x = 0; x < 8; ++x) gave_up += h[x].pad[0];
        printf("  static items %d, %d workgroups per CU, lag %4u: %7.1f us  (%5.2f TB/s of kernel traffic)  tickets %u %u .. gave up %u\n", nt, wgs_per_cu, lag, best * 1e3,
               4.0 * total / (best * 1e-3) / 1e12, h[0].ticket, h[7].ticket, gave_up);
        fflush(stdout);
    }
    return 0;
}
