// Device-wide exclusive prefix sum over a *computed* int sequence (reduce-then-scan, three
// launches, deterministic).  `In` is a functor `int operator()(int64_t i)`; `Out` is a functor
// `void operator()(int64_t i, int value_i, int exclusive_prefix_i)`.
// Used for: first-point flags -> voxel ids (voxelize.hip) and popcount(words) -> ranks
// (rulebook.hip).  wave = 64 lanes.
#pragma once
#include "s2d_common.h"

namespace s2d {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ int wave_inclusive_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan of one int per thread across a 256-thread block; returns the block total in *total
__device__ __forceinline__ int block_exclusive_scan(int v, int *total, int *lds /*[4]*/) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = wave_inclusive_scan(v);
    if (lane == 63) lds[wid] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) {
        int s = lds[w];
        if (w < wid) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

template <typename In>
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(In in, int64_t n, int *block_sums) {
    __shared__ int lds[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int t = 0; t < SCAN_ITEMS; ++t) {
        int64_t i = base + t;
        if (i < n) s += in(i);
    }
    int tot;
    block_exclusive_scan(s, &tot, lds);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// single block: exclusive scan of block_sums[nb] in place, total -> *total_out (may be null)
static __global__ __launch_bounds__(SCAN_THREADS) void scan_sums_kernel(int *block_sums, int nb, int *total_out) {
    __shared__ int lds[4];
    int carry = 0;
    for (int base = 0; base < nb; base += SCAN_THREADS) {
        int i = base + threadIdx.x;
        int v = i < nb ? block_sums[i] : 0;
        int tot;
        int ex = block_exclusive_scan(v, &tot, lds);
        if (i < nb) block_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

template <typename In, typename Out>
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(In in, Out out, int64_t n,
                                                                  const int *block_sums) {
    __shared__ int lds[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int t = 0; t < SCAN_ITEMS; ++t) {
        int64_t i = base + t;
        v[t] = i < n ? in(i) : 0;
        s += v[t];
    }
    int tot;
    int ex = block_exclusive_scan(s, &tot, lds) + block_sums[blockIdx.x];
#pragma unroll
    for (int t = 0; t < SCAN_ITEMS; ++t) {
        int64_t i = base + t;
        if (i < n) out(i, v[t], ex);
        ex += v[t];
    }
}

static inline int scan_num_blocks(int64_t n) { return (int)ceil_div(n > 0 ? n : 1, SCAN_TILE); }

// ---- single-pass variant (r04): one launch instead of three ------------------------------------------------------------------
// Every tile publishes its sum as an 8-byte granule {tag = 1, sum} (one relaxed agent-scope store: MI355X_MICROARCH.md "R2", the
// data is the flag) and then adds up the granules of ALL earlier tiles - they are published before anybody looks back, so there is
// no serial chain, and earlier tiles are dispatched earlier, so a resident workgroup never waits on one that cannot start.
// `flags` = one uint64 per tile whose upper half is != 1 before the launch (the callers' memsets write 0x00 or 0x7F bytes there).
// The spin is bounded: on a time-out *err is set to 1 (never observed) and the sums are wrong rather than the device hung.
constexpr int SCAN1_ITEMS = 16;
constexpr int SCAN1_TILE = SCAN_THREADS * SCAN1_ITEMS;
static inline int scan1_num_blocks(int64_t n) { return (int)ceil_div(n > 0 ? n : 1, SCAN1_TILE); }

template <typename In, typename Out>
__global__ __launch_bounds__(SCAN_THREADS) void scan_onepass_kernel(In in, Out out, int64_t n, unsigned long long *flags, int *total_out,
                                                                     int *err) {
    __shared__ int lds[4];
    const int tile = (int)blockIdx.x;
    const int64_t base = (int64_t)tile * SCAN1_TILE + (int64_t)threadIdx.x * SCAN1_ITEMS;
    int v[SCAN1_ITEMS];
    int s = 0;
#pragma unroll
    for (int t = 0; t < SCAN1_ITEMS; ++t) {
        const int64_t i = base + t;
        v[t] = i < n ? in(i) : 0;
        s += v[t];
    }
    int tot;
    const int ex = block_exclusive_scan(s, &tot, lds);
    if (threadIdx.x == 0) __hip_atomic_store(&flags[tile], (1ull << 32) | (unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int before = 0;
    for (int t = threadIdx.x; t < tile; t += SCAN_THREADS) {
        unsigned spins = 0;
        unsigned long long f;
        while (((f = __hip_atomic_load(&flags[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != 1ull) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) {
                if (err) atomicExch(err, 1);
                f = 0;
                break;
            }
        }
        before += (int)(unsigned)f;
    }
    int prev;
    block_exclusive_scan(before, &prev, lds);
    if (tile == (int)gridDim.x - 1 && threadIdx.x == 0 && total_out) *total_out = prev + tot;
    int run = prev + ex;
#pragma unroll
    for (int t = 0; t < SCAN1_ITEMS; ++t) {
        const int64_t i = base + t;
        if (i < n) out(i, v[t], run);
        run += v[t];
    }
}

// flags must hold scan1_num_blocks(n) uint64 (see above for their initial state)
template <typename In, typename Out>
static inline int device_exclusive_scan_onepass(In in, Out out, int64_t n, unsigned long long *flags, int *total_out, int *err, hipStream_t st) {
    hipLaunchKernelGGL((scan_onepass_kernel<In, Out>), dim3(scan1_num_blocks(n)), dim3(SCAN_THREADS), 0, st, in, out, n, flags, total_out, err);
    S2D_LAUNCH_CHECK();
    return 0;
}

// block_sums must hold scan_num_blocks(n) ints
template <typename In, typename Out>
static inline int device_exclusive_scan(In in, Out out, int64_t n, int *block_sums, int *total_out,
                                        hipStream_t st) {
    const int nb = scan_num_blocks(n);
    hipLaunchKernelGGL(scan_reduce_kernel<In>, dim3(nb), dim3(SCAN_THREADS), 0, st, in, n, block_sums);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, block_sums, nb, total_out);
    hipLaunchKernelGGL((scan_apply_kernel<In, Out>), dim3(nb), dim3(SCAN_THREADS), 0, st, in, out, n, block_sums);
    S2D_LAUNCH_CHECK();
    return 0;
}

}  // namespace s2d
