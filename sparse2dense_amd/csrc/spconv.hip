// Sparse 3-D convolution (submanifold and strided) as an OUTPUT-STATIONARY implicit GEMM on the
// gfx950 matrix cores.  Replaces spconv's indice_conv / indice_conv_backward (27x gather kernel +
// cuBLAS mm + scatter-add kernel with HBM round trips between them) by one kernel per layer:
//
//   forward / data-gradient  (spconv_fwd_mfma)
//     a wave owns 16*MT output rows and a 16*NT-wide slab of output channels; it walks the K
//     kernel offsets, gathers the neighbour rows named by nbr[k][rows] straight into MFMA A
//     fragments (float4 per lane, rows are L2/MALL resident), multiplies by W[k] staged per
//     workgroup in LDS with global_load_lds (double buffered, one barrier per offset) and keeps
//     the fp32 accumulators in registers until the single, vectorised store.  No atomics, no
//     intermediate buffers, deterministic.  Offsets with no active row in the wave are skipped.
//   weight-gradient (spconv_wgrad_mfma)
//     grid = (offset k, row split); M = cin, N = cout, K = rows.  A/B fragments are float4 row
//     segments of in[nbr[k][o]] and dout[o]; partial tiles go to a [split][K][cin][cout] slab
//     that a second kernel reduces in a fixed order (deterministic).
//
// Arithmetic: v_mfma_f32_16x16x4_f32 — exact fp32 FMA chains (fp32 in, fp32 accumulate).
// Channel counts that are not multiples of 16 (the 5-channel input layer) take the VALU kernels
// at the bottom; they are bandwidth-trivial.
#include "s2d_common.h"
#include <cstdlib>

namespace s2d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int N>
struct VecF;
template <>
struct VecF<1> { typedef float type; };
template <>
struct VecF<2> { typedef float2 type; };
template <>
struct VecF<4> { typedef float4 type; };

// bf16 bit patterns (low 16 bits of a 32-bit value) of N consecutive elements at p; T = float (rounded here) or __bf16
typedef unsigned u32v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned bf16_bits(float v) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v); }
template <typename T, int N>
__device__ __forceinline__ void load_bf16_bits(const T *__restrict__ p, unsigned (&out)[N]) {
    if constexpr (sizeof(T) == 4) {
        if constexpr (N == 1) {
            out[0] = bf16_bits(*reinterpret_cast<const float *>(p));
        } else if constexpr (N == 2) {
            const float2 v = *reinterpret_cast<const float2 *>(p);
            out[0] = bf16_bits(v.x); out[1] = bf16_bits(v.y);
        } else {
            const float4 v = *reinterpret_cast<const float4 *>(p);
            out[0] = bf16_bits(v.x); out[1] = bf16_bits(v.y); out[2] = bf16_bits(v.z); out[3] = bf16_bits(v.w);
        }
    } else {
        if constexpr (N == 1) {   // the aligned 32-bit word holding the element, then a shift (no 16-bit loads)
            const uintptr_t a = reinterpret_cast<uintptr_t>(p);
            const unsigned w = *reinterpret_cast<const unsigned *>(a & ~(uintptr_t)3);
            out[0] = (a & 2) ? (w >> 16) : (w & 0xffffu);
        } else if constexpr (N == 2) {
            const unsigned w = *reinterpret_cast<const unsigned *>(p);
            out[0] = w & 0xffffu; out[1] = w >> 16;
        } else {
            const uint2 w = *reinterpret_cast<const uint2 *>(p);
            out[0] = w.x & 0xffffu; out[1] = w.x >> 16; out[2] = w.y & 0xffffu; out[3] = w.y >> 16;
        }
    }
}
__device__ __forceinline__ float vec_get(const float &v, int) { return v; }
__device__ __forceinline__ float vec_get(const float2 &v, int i) { return i == 0 ? v.x : v.y; }
__device__ __forceinline__ float vec_get(const float4 &v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// Stage a [CIN][CT] fp32 slab of W[k] (row stride `cout` floats in global memory) into LDS,
// lane-linear, with the LDS-DMA path (16 B per lane, 1 KiB per wave instruction).
template <int CIN, int CT>
__device__ __forceinline__ void stage_weights(float *lds, const float *__restrict__ wk, int cout, int wid, int lane) {
    constexpr int UNITS = CIN * CT / 256;  // wave-instructions of 64 lanes x 4 floats
#pragma unroll
    for (int u = 0; u < (UNITS + 3) / 4; ++u) {
        const int unit = u * 4 + wid;
        if (unit < UNITS) {
            const int chunk = unit * 64 + lane;  // 16-byte chunk id inside the slab
            const int row = (chunk * 4) / CT;
            const int col = (chunk * 4) % CT;
            const float *src = wk + (int64_t)row * cout + col;
            float *dst = lds + unit * 256;  // wave-uniform base; hardware adds lane*16
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    }
}

template <int CIN, int NT, int MT>
__global__ __launch_bounds__(256) void spconv_fwd_mfma(const float *__restrict__ in, const float *__restrict__ w,
                                                       const float *__restrict__ bias, const int32_t *__restrict__ nbr,
                                                       int n_out, int kvol, int cout, float *__restrict__ out) {
    constexpr int CT = 16 * NT;   // output channels per workgroup
    constexpr int KT = CIN / 16;  // float4 A loads per row
    typedef typename VecF<NT>::type vecn;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *wbuf0 = reinterpret_cast<float *>(smem);
    float *wbuf1 = wbuf0 + CIN * CT;

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int co0 = blockIdx.y * CT;
    const int row0 = (blockIdx.x * 4 + wid) * (16 * MT);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage_weights<CIN, CT>(wbuf0, w + co0, cout, wid, lane);

    for (int k = 0; k < kvol; ++k) {
        float *wb = (k & 1) ? wbuf1 : wbuf0;
        if (k + 1 < kvol)
            stage_weights<CIN, CT>((k & 1) ? wbuf0 : wbuf1, w + (int64_t)(k + 1) * CIN * cout + co0, cout, wid, lane);

        int j[MT];
        bool mine = false;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int row = row0 + 16 * m + r;
            j[m] = row < n_out ? nbr[(int64_t)k * n_out + row] : -1;
            mine = mine || (j[m] >= 0);
        }
        const bool any = __ballot(mine) != 0ull;  // wave-uniform
        float4 a[MT][KT];
        if (any) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float4 *src = reinterpret_cast<const float4 *>(in + (int64_t)(j[m] >= 0 ? j[m] : 0) * CIN) + q;
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    a[m][t] = j[m] >= 0 ? src[t * 4] : float4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        if (k == 0) __syncthreads();  // W[0] landed (the barrier's release drains the LDS-DMA)
        if (any) {
#pragma unroll
            for (int t = 0; t < KT; ++t) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kk = 16 * t + 4 * q + e;  // K index of this lane's A element
                    const vecn b = *reinterpret_cast<const vecn *>(wb + kk * CT + r * NT);
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const float av = vec_get(a[m][t], e);
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, vec_get(b, n), acc[m][n], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();  // everyone is done with wb; W[k+1] has landed
    }

    // C/D layout: column = lane&15 (-> channels r*NT .. r*NT+NT-1), row = (lane>>4)*4 + reg
    vecn bv;
    {
        float tmp[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) tmp[n] = bias ? bias[co0 + r * NT + n] : 0.f;
        bv = *reinterpret_cast<vecn *>(tmp);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = row0 + 16 * m + 4 * q + reg;
            if (row < n_out) {
                float tmp[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) tmp[n] = acc[m][n][reg] + vec_get(bv, n);
                *reinterpret_cast<vecn *>(out + (int64_t)row * cout + co0 + r * NT) = *reinterpret_cast<vecn *>(tmp);
            }
        }
    }
}

// ---- weight gradient -----------------------------------------------------------------------------
template <int CIN, int COUT>
struct WgradCfg {
    static constexpr int VA = CIN / 16 < 4 ? CIN / 16 : 4;     // floats per A load
    static constexpr int LA = CIN / (16 * VA);                 // A loads per pair
    static constexpr int VB = COUT >= 32 ? 2 : 1;              // floats per B load = N tiles per wave
    static constexpr int WCO = COUT / (16 * VB);               // waves across cout
    static constexpr int WROW = 4 / WCO;                       // waves across rows
};

// (kernel offset, row-split block) of this workgroup.  Workgroups go to the 8 XCDs round-robin in launch order; when the
// number of row-split blocks is a multiple of 8, XCD x gets every offset of the row blocks = x (mod 8), so that the rows
// its L2 sees are 1/8 of the tensor (the 27 offsets of a row block re-read the same input / gradient rows; spread
// over all XCDs each 4 MB L2 streamed the whole tensor).
__device__ __forceinline__ void wgrad_block(int kvol, int &k, int &by) {
    const int gy = gridDim.y, b = blockIdx.x + kvol * blockIdx.y;
    if ((gy & 7) == 0) {
        const int xcd = b & 7, q = b >> 3;
        k = q % kvol;
        by = (q / kvol) * 8 + xcd;
    } else {
        k = blockIdx.x;
        by = blockIdx.y;
    }
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void spconv_wgrad_mfma(const float *__restrict__ in, const float *__restrict__ dout,
                                                         const int32_t *__restrict__ nbr, int n_out, int kvol,
                                                         int rows_per_split, float *__restrict__ partial) {
    typedef WgradCfg<CIN, COUT> C;
    typedef typename VecF<C::VA>::type veca;
    typedef typename VecF<C::VB>::type vecb;
    constexpr int U = 4;  // pair groups (of 4 pairs) whose loads are in flight together
    __shared__ int2 pair_lds[4][64];  // per wave: compacted (o, j) of the current 64-row window
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    int k, by;
    wgrad_block(kvol, k, by);
    const int wco = wid % C::WCO, wrow = wid / C::WCO;
    const int split = by * C::WROW + wrow;
    const int co_base = wco * 16 * C::VB;
    const int r_begin = split * rows_per_split;
    const int r_end = min(n_out, r_begin + rows_per_split);
    const int32_t *nk = nbr + (int64_t)k * n_out;
    int2 *mypairs = pair_lds[wid];

    f32x4 acc[C::LA][C::VA][C::VB];
#pragma unroll
    for (int la = 0; la < C::LA; ++la)
#pragma unroll
        for (int e = 0; e < C::VA; ++e)
#pragma unroll
            for (int f = 0; f < C::VB; ++f) acc[la][e][f] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int base = r_begin; base < r_end; base += 64) {
        // 64-row window: coalesced read of the gather map, wave-level compaction of the active pairs
        const int o_l = base + lane;
        const int j_l = o_l < r_end ? nk[o_l] : -1;
        const unsigned long long mask = __ballot(j_l >= 0);
        const int cnt = __popcll(mask);
        if (cnt == 0) continue;  // wave-uniform
        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
        if (j_l >= 0) mypairs[rank] = make_int2(o_l, j_l);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave: LDS ops are ordered, this pins the compiler
        const int groups = (cnt + 3) >> 2;
        for (int g0 = 0; g0 < groups; g0 += U) {
            veca av[U][C::LA];
            vecb bv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = 4 * (g0 + u) + q;
                const bool ok = idx < cnt;
                const int2 pr = mypairs[min(idx, cnt - 1)];   // branch-free gather, zeroed by select below
#pragma unroll
                for (int la = 0; la < C::LA; ++la) {
                    veca v = *reinterpret_cast<const veca *>(in + (int64_t)pr.y * CIN + 64 * la + C::VA * i16);
                    float tmp[C::VA];
#pragma unroll
                    for (int e = 0; e < C::VA; ++e) tmp[e] = ok ? vec_get(v, e) : 0.f;
                    av[u][la] = *reinterpret_cast<veca *>(tmp);
                }
                bv[u] = *reinterpret_cast<const vecb *>(dout + (int64_t)pr.x * COUT + co_base + C::VB * i16);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (4 * (g0 + u) < cnt) {  // wave-uniform
#pragma unroll
                    for (int la = 0; la < C::LA; ++la)
#pragma unroll
                        for (int e = 0; e < C::VA; ++e)
#pragma unroll
                            for (int f = 0; f < C::VB; ++f)
                                acc[la][e][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(vec_get(av[u][la], e), vec_get(bv[u], f),
                                                                                      acc[la][e][f], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads of this window done before the next overwrite
    }
    // tile (la,e,f), reg: ci = 64*la + VA*(4*q+reg) + e ; co = co_base + VB*i16 + f
    float *dst = partial + ((int64_t)split * kvol + k) * CIN * COUT;
#pragma unroll
    for (int la = 0; la < C::LA; ++la)
#pragma unroll
        for (int e = 0; e < C::VA; ++e)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int ci = 64 * la + C::VA * (4 * q + reg) + e;
                float tmp[C::VB];
#pragma unroll
                for (int f = 0; f < C::VB; ++f) tmp[f] = acc[la][e][f][reg];
                *reinterpret_cast<vecb *>(dst + (int64_t)ci * COUT + co_base + C::VB * i16) = *reinterpret_cast<vecb *>(tmp);
            }
}

// bf16-input variant: same decomposition, but one v_mfma_f32_16x16x32_bf16 contracts 32 pairs.
// Lane (i16, q) owns pairs 8q..8q+7 of a 32-pair group and loads, for each of them, its float4 /
// float2 channel slice of in[j] and dout[o]; the k-contiguous bf16x8 fragments are assembled in
// registers (the "transpose" costs nothing: every lane simply loads the elements it multiplies).
template <int CIN, int COUT, typename T>
__global__ __launch_bounds__(256) void spconv_wgrad_bf16(const T *__restrict__ in, const T *__restrict__ dout,
                                                         const int32_t *__restrict__ nbr, int n_out, int kvol,
                                                         int rows_per_split, float *__restrict__ partial) {
    typedef WgradCfg<CIN, COUT> C;
    typedef typename VecF<C::VB>::type vecb_out;
    __shared__ int2 pair_lds[4][64];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    int k, by;
    wgrad_block(kvol, k, by);
    const int wco = wid % C::WCO, wrow = wid / C::WCO;
    const int split = by * C::WROW + wrow;
    const int co_base = wco * 16 * C::VB;
    const int r_begin = split * rows_per_split;
    const int r_end = min(n_out, r_begin + rows_per_split);
    const int32_t *nk = nbr + (int64_t)k * n_out;
    int2 *mypairs = pair_lds[wid];

    f32x4 acc[C::LA][C::VA][C::VB];
#pragma unroll
    for (int la = 0; la < C::LA; ++la)
#pragma unroll
        for (int e = 0; e < C::VA; ++e)
#pragma unroll
            for (int f = 0; f < C::VB; ++f) acc[la][e][f] = f32x4{0.f, 0.f, 0.f, 0.f};

    // the neighbour indices of the next 64-row window are fetched while this one is processed (one of the serial
    // memory round trips of a window: index -> pair list -> gathers)
    int j_next = r_begin + lane < r_end ? nk[r_begin + lane] : -1;
    for (int base = r_begin; base < r_end; base += 64) {
        const int o_l = base + lane;
        const int j_l = j_next;
        j_next = o_l + 64 < r_end ? nk[o_l + 64] : -1;
        const unsigned long long mask = __ballot(j_l >= 0);
        const int cnt = __popcll(mask);
        if (cnt == 0) continue;
        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
        if (j_l >= 0) mypairs[rank] = make_int2(o_l, j_l);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int g = 0; 32 * g < cnt; ++g) {
            // elements are carried as bf16 bit patterns in the low half of a 32-bit register; the k-contiguous MFMA
            // fragments are assembled with 32-bit selects/shifts/ors
            unsigned abits[8][C::LA][C::VA], bbits[8][C::VB];
            // branch-free: out-of-range slots re-load the last valid pair and are zeroed by a select
            // (a branch around each load makes hipcc wait per element and serialises the gathers)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int idx = 32 * g + 8 * q + e;
                const int2 pr = mypairs[min(idx, cnt - 1)];
#pragma unroll
                for (int la = 0; la < C::LA; ++la) load_bf16_bits<T, C::VA>(in + (int64_t)pr.y * CIN + 64 * la + C::VA * i16, abits[e][la]);
                load_bf16_bits<T, C::VB>(dout + (int64_t)pr.x * COUT + co_base + C::VB * i16, bbits[e]);
            }
            const int nvalid = cnt - 32 * g - 8 * q;  // this lane's valid slots: e < nvalid
            bf16x8 bfr[C::VB];
#pragma unroll
            for (int f = 0; f < C::VB; ++f) {
                u32v4 w;
#pragma unroll
                for (int h = 0; h < 4; ++h)
                    w[h] = (2 * h < nvalid ? bbits[2 * h][f] : 0u) | ((2 * h + 1 < nvalid ? bbits[2 * h + 1][f] : 0u) << 16);
                bfr[f] = __builtin_bit_cast(bf16x8, w);
            }
#pragma unroll
            for (int la = 0; la < C::LA; ++la)
#pragma unroll
                for (int t = 0; t < C::VA; ++t) {
                    u32v4 w;
#pragma unroll
                    for (int h = 0; h < 4; ++h)
                        w[h] = (2 * h < nvalid ? abits[2 * h][la][t] : 0u) | ((2 * h + 1 < nvalid ? abits[2 * h + 1][la][t] : 0u) << 16);
                    const bf16x8 afr = __builtin_bit_cast(bf16x8, w);
#pragma unroll
                    for (int f = 0; f < C::VB; ++f)
                        acc[la][t][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr[f], acc[la][t][f], 0, 0, 0);
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float *dst = partial + ((int64_t)split * kvol + k) * CIN * COUT;
#pragma unroll
    for (int la = 0; la < C::LA; ++la)
#pragma unroll
        for (int e = 0; e < C::VA; ++e)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int ci = 64 * la + C::VA * (4 * q + reg) + e;
                float tmp[C::VB];
#pragma unroll
                for (int f = 0; f < C::VB; ++f) tmp[f] = acc[la][e][f][reg];
                *reinterpret_cast<vecb_out *>(dst + (int64_t)ci * COUT + co_base + C::VB * i16) = *reinterpret_cast<vecb_out *>(tmp);
            }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ partial, int n_split, int64_t size,
                                                           float *__restrict__ dw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= size) return;
    float s = 0.f;
    for (int sp = 0; sp < n_split; ++sp) s += partial[(int64_t)sp * size + i];
    dw[i] = s;
}

// size % 4 == 0: a workgroup owns 16 float4 columns; 16 thread rows stride over the splits (the small layers have ~7k
// elements and ~76 splits: one thread per element walked the splits as a serial chain of loads and took longer than
// the gather kernel's tail), then a fixed-order fold through LDS (deterministic).
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float4 *__restrict__ partial, int n_split, int64_t size4,
                                                            float4 *__restrict__ dw) {
    __shared__ float4 red[16][16];
    const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int64_t i = (int64_t)blockIdx.x * 16 + col;
    float4 s = float4{0.f, 0.f, 0.f, 0.f};
    if (i < size4) {
        for (int sp = sl; sp < n_split; sp += 16) {
            const float4 v = partial[(int64_t)sp * size4 + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[sl][col] = s;
    __syncthreads();
    if (sl == 0 && i < size4) {
        float4 t = red[0][col];
#pragma unroll
        for (int r = 1; r < 16; ++r) {
            const float4 v = red[r][col];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        dw[i] = t;
    }
}

// ---- VALU fallbacks (channel counts not multiple of 16, e.g. the 5-channel input layer) ----------
__global__ __launch_bounds__(256) void spconv_fwd_valu(const float *__restrict__ in, const float *__restrict__ w,
                                                       const float *__restrict__ bias, const int32_t *__restrict__ nbr,
                                                       int n_out, int kvol, int cin, int cout, float *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (row, co)
    const int64_t o = t / cout;
    const int co = (int)(t - o * cout);
    if (o >= n_out) return;
    float acc = bias ? bias[co] : 0.f;
    for (int k = 0; k < kvol; ++k) {
        const int j = nbr[(int64_t)k * n_out + o];
        if (j < 0) continue;
        const float *x = in + (int64_t)j * cin;
        const float *wk = w + (int64_t)k * cin * cout + co;
        for (int ci = 0; ci < cin; ++ci) acc = fmaf(x[ci], wk[(int64_t)ci * cout], acc);
    }
    out[t] = acc;
}

__global__ __launch_bounds__(256) void spconv_wgrad_valu(const float *__restrict__ in, const float *__restrict__ dout,
                                                         const int32_t *__restrict__ nbr, int n_out, int kvol, int cin,
                                                         int cout, int rows_per_split, float *__restrict__ partial) {
    const int k = blockIdx.x, split = blockIdx.y;
    const int r_begin = split * rows_per_split;
    const int r_end = min(n_out, r_begin + rows_per_split);
    const int32_t *nk = nbr + (int64_t)k * n_out;
    for (int e = threadIdx.x; e < cin * cout; e += blockDim.x) {
        const int ci = e / cout, co = e - ci * cout;
        float s = 0.f;
        for (int o = r_begin; o < r_end; ++o) {
            const int j = nk[o];
            if (j >= 0) s = fmaf(in[(int64_t)j * cin + ci], dout[(int64_t)o * cout + co], s);
        }
        partial[(((int64_t)split * kvol + k) * cin + ci) * cout + co] = s;
    }
}

// ---- launch helpers --------------------------------------------------------------------------------
template <int CIN, int NT>
static int launch_fwd(const float *in, const float *w, const float *bias, const int32_t *nbr, int n_out, int kvol,
                      int cout, float *out, hipStream_t st) {
    constexpr int CT = 16 * NT;
    const size_t lds = 2 * (size_t)CIN * CT * sizeof(float);
    const int ytiles = cout / CT;
    const bool small = ceil_div(n_out, 128) * ytiles < 512;  // keep >= 2 workgroups per CU in flight
    if (small) {
        auto kern = spconv_fwd_mfma<CIN, NT, 1>;
        S2D_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(n_out, 64), ytiles), dim3(256), lds, st, in, w, bias, nbr, n_out,
                           kvol, cout, out);
    } else {
        auto kern = spconv_fwd_mfma<CIN, NT, 2>;
        S2D_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(n_out, 128), ytiles), dim3(256), lds, st, in, w, bias, nbr,
                           n_out, kvol, cout, out);
    }
    S2D_LAUNCH_CHECK();
    return 0;
}

template <int CIN>
static int dispatch_fwd_cout(const float *in, const float *w, const float *bias, const int32_t *nbr, int n_out, int kvol,
                             int cout, float *out, hipStream_t st) {
    switch (cout) {
        case 16: return launch_fwd<CIN, 1>(in, w, bias, nbr, n_out, kvol, cout, out, st);
        case 32: return launch_fwd<CIN, 2>(in, w, bias, nbr, n_out, kvol, cout, out, st);
        case 64: return launch_fwd<CIN, 4>(in, w, bias, nbr, n_out, kvol, cout, out, st);
        case 128: return launch_fwd<CIN, 4>(in, w, bias, nbr, n_out, kvol, cout, out, st);
        default: return -1;
    }
}

// bf16-storage weight gradient through LDS transpose reads: with 2-byte elements the kernel above issues one 2- to
// 8-byte load per lane and pair, which leaves the launches with many pairs and few FLOPs bound by load instructions.  Here every lane fetches 16 bytes of a gathered row, the wave parks the 32-pair tile in its
// private LDS slice as [pair][channels] and the MFMA fragments come back through the gfx950 transpose read
// (ds_read_b64_tr_b16: a 16-lane group reads a [4 pair][16 ch] block, each lane gets the 4 pairs of its channel; lane
// group g therefore holds pairs {4g..4g+3, 16+4g..16+4g+3} of the 32-pair K-step for A and B alike).  One wave = one
// row split; no workgroup barrier (LDS operations of a wave execute in order).
typedef int wg_i32x2 __attribute__((ext_vector_type(2)));
typedef int wg_i32x4 __attribute__((ext_vector_type(4)));
template <int HI_OFF>
__device__ __forceinline__ bf16x8 wg_tr_read(unsigned lds_addr) {
    wg_i32x2 lo, hi;   // early-clobber: the address register is read again by the second instruction
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(lo), "=&v"(hi) : "v"(lds_addr), "n"(HI_OFF) : "memory");
    wg_i32x4 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
    return __builtin_bit_cast(bf16x8, v);
}

// issue only (no wait): the caller batches all fragment reads of a chunk and waits once
template <int HI_OFF>
__device__ __forceinline__ void wg_tr_issue(wg_i32x2 &lo, wg_i32x2 &hi, unsigned lds_addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3" : "=&v"(lo), "=&v"(hi) : "v"(lds_addr), "n"(HI_OFF) : "memory");
}
__device__ __forceinline__ bf16x8 wg_tr_pack(const wg_i32x2 &lo, const wg_i32x2 &hi) {
    wg_i32x4 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
    return __builtin_bit_cast(bf16x8, v);
}

template <int CIN, int COUT>
struct WgS16Cfg {
    static constexpr int NBMAX = CIN == 128 ? 2 : 4;
    static constexpr int NB = COUT / 16 < NBMAX ? COUT / 16 : NBMAX;              // 16-column tiles per wave
    static constexpr int WCO = COUT / (16 * NB);                                  // waves across the output channels
    static constexpr int WROW = 4 / WCO;                                          // row splits per workgroup
    static constexpr int MA = CIN / 16;
};

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void spconv_wgrad_s16_tr(const __bf16 *__restrict__ in, const __bf16 *__restrict__ dout,
                                                           const int32_t *__restrict__ nbr, int n_out, int kvol,
                                                           int rows_per_split, float *__restrict__ partial) {
    typedef WgS16Cfg<CIN, COUT> C;
    constexpr int MA = C::MA, NB = C::NB;
    constexpr int CB = 16 * NB;                           // output channels of this wave
    constexpr int RS_A = CIN * 2, RS_B = CB * 2;          // row strides of the LDS tiles in bytes
    constexpr int LPR_A = CIN / 8, LPR_B = CB / 8;        // lanes (16-byte pieces) per gathered row
    constexpr int TILE_A = 32 * RS_A, TILE_B = 32 * RS_B;
    constexpr int LA = (32 * LPR_A + 63) / 64, LB = (32 * LPR_B + 63) / 64;
    constexpr int RING = 256;   // pair ring per wave (at most 64 pending + 64 being appended)
    __shared__ int2 pair_lds[4][RING];
    __shared__ __attribute__((aligned(16))) char tile_lds[4][TILE_A + TILE_B];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wco = wid % C::WCO, wrow = wid / C::WCO;
    const int g = lane >> 4, c16 = lane & 15;
    int k, by;
    wgrad_block(kvol, k, by);
    const int split = by * C::WROW + wrow;
    const int co_base = wco * CB;
    const int r_begin = split * rows_per_split;
    const int r_end = min(n_out, r_begin + rows_per_split);
    const int32_t *nk = nbr + (int64_t)k * n_out;
    char *ta = tile_lds[wid], *tb = tile_lds[wid] + TILE_A;
    const unsigned ta_addr = (unsigned)(size_t)((__attribute__((address_space(3))) char *)ta);
    const unsigned tb_addr = (unsigned)(size_t)((__attribute__((address_space(3))) char *)tb);
    const int tr_off_a = (4 * g + (c16 >> 2)) * RS_A + (c16 & 3) * 8;
    const int tr_off_b = (4 * g + (c16 >> 2)) * RS_B + (c16 & 3) * 8;

    f32x4 acc[MA][NB];
#pragma unroll
    for (int m = 0; m < MA; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // The rows of this wave are a chain of dependent round trips: neighbour index -> compacted pair list (LDS) -> row
    // gathers -> LDS tile -> transpose reads -> MFMA.  It is software pipelined over CHUNKS of 32 pairs: pairs of as many
    // 64-row windows as needed are appended to a ring (indices prefetched two windows ahead), chunks are cut from the
    // ring independent of window boundaries (only the last chunk of the wave is partial - cutting per window left the
    // typical 36-pair window with a 4-pair second chunk), and the gathers of chunk c+1 are issued before chunk c is
    // consumed.  The gather is unconditional straight-line code (clamped indices) so that hipcc's vmcnt bookkeeping stays
    // exact: with a branch around it the merge of the two paths made every consume wait for the next chunk's loads too.
    // The loop is unrolled by two: copying registers that have loads in flight would wait for them.
    struct Gather {
        uint4 va[LA], vb[LB];
    };
    int2 *ring = pair_lds[wid];
    int tail = 0;                    // pairs appended so far (wave-uniform)
    int nbase = r_begin;             // first row of the next window to append
    auto load_idx = [&](int base) -> int { return base + lane < r_end ? nk[min(base + lane, n_out - 1)] : -1; };
    int jp0 = load_idx(nbase), jp1 = load_idx(nbase + 64);
    if (lane == 0) ring[0] = make_int2(0, 0);   // clamped gathers of an empty ring read row 0
    auto top_up = [&](int need) {    // append windows until `need` pairs exist or the rows are exhausted
        while (tail < need && nbase < r_end) {
            const int j_l = jp0;
            jp0 = jp1;
            jp1 = load_idx(nbase + 128);
            const unsigned long long mask = __ballot(j_l >= 0);
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
            if (j_l >= 0) ring[(tail + rank) & (RING - 1)] = make_int2(nbase + lane, j_l);
            tail += __popcll(mask);
            nbase += 64;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto gather = [&](int start, int cnt, Gather &gt) {   // 16 bytes per lane; rows past cnt re-read the chunk's last pair
        const int last = cnt > 0 ? cnt - 1 : 0;
#pragma unroll
        for (int u = 0; u < LA; ++u) {
            const int c = lane + 64 * u, pr = c / LPR_A, part = c % LPR_A;
            const int2 pp = ring[(start + min(pr, last)) & (RING - 1)];
            gt.va[u] = *reinterpret_cast<const uint4 *>(in + (int64_t)pp.y * CIN + part * 8);
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int c = lane + 64 * u, pr = c / LPR_B, part = c % LPR_B;
            const int2 pp = ring[(start + min(pr, last)) & (RING - 1)];
            gt.vb[u] = *reinterpret_cast<const uint4 *>(dout + (int64_t)pp.x * COUT + co_base + part * 8);
        }
    };
    auto consume = [&](Gather &gt, int cnt) {   // rows past the pair count are zeroed; LDS tile; transpose reads; MFMAs
#pragma unroll
        for (int u = 0; u < LA; ++u) {
            const int c = lane + 64 * u;
            const uint4 v = c / LPR_A < cnt ? gt.va[u] : uint4{0u, 0u, 0u, 0u};
            *reinterpret_cast<uint4 *>(ta + c * 16) = v;
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int c = lane + 64 * u;
            const uint4 v = c / LPR_B < cnt ? gt.vb[u] : uint4{0u, 0u, 0u, 0u};
            *reinterpret_cast<uint4 *>(tb + c * 16) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wg_i32x2 blo[NB], bhi[NB], alo[MA], ahi[MA];   // all fragment reads of the chunk, one wait
#pragma unroll
        for (int n = 0; n < NB; ++n) wg_tr_issue<16 * RS_B>(blo[n], bhi[n], tb_addr + tr_off_b + n * 32);
#pragma unroll
        for (int m = 0; m < MA; ++m) wg_tr_issue<16 * RS_A>(alo[m], ahi[m], ta_addr + tr_off_a + m * 32);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MA; ++m) {
            const bf16x8 a = wg_tr_pack(alo[m], ahi[m]);
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wg_tr_pack(blo[n], bhi[n]), acc[m][n], 0, 0, 0);
        }
    };
    // chunk (start, cnt) is in flight in `cur`; stage the next chunk into `nxt`, then consume `cur`
    int start = 0, cnt = 0;
    auto step = [&](Gather &cur, Gather &nxt) {
        const int nstart = start + cnt;
        top_up(nstart + 32);
        const int ncnt = min(32, tail - nstart);
        gather(ncnt > 0 ? nstart : max(tail - 1, 0), ncnt, nxt);   // nothing left: re-read the last valid pair (never consumed)
        __builtin_amdgcn_sched_barrier(0);   // the scheduler otherwise sinks the gather below the MFMAs of `cur`
        consume(cur, cnt);
        __builtin_amdgcn_sched_barrier(0);
        start = nstart;
        cnt = ncnt;
    };
    Gather g0, g1;
    top_up(32);
    cnt = min(32, tail);
    gather(0, cnt, g0);   // tail == 0: ring[0] = (0, 0)
    while (cnt > 0) {   // no exit between the two steps: an empty chunk (cnt = 0) is a zero tile, consumed once at most
        step(g0, g1);
        step(g1, g0);
    }
    // C/D layout: row (ci within tile) = 4*(lane>>4)+reg, col (co within tile) = lane&15
    float *dst = partial + ((int64_t)split * kvol + k) * CIN * COUT;
#pragma unroll
    for (int m = 0; m < MA; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) dst[(int64_t)(16 * m + 4 * g + reg) * COUT + co_base + 16 * n + c16] = acc[m][n][reg];
}

// ---- 128 input channels, bf16 storage: workgroup-cooperative variant ------------------------------------------------
// At 128 input channels a wave of the kernel above owns all 128 ci x 32 co, so the four waves of a workgroup each
// gathered the same 256-byte input rows.  Here the workgroup gathers a chunk of 32 pairs ONCE (16 bytes per thread and
// piece: A [32][128] and B [32][128] into double-buffered LDS tiles, one barrier per chunk) and every wave takes its
// 32-column slice of B with transpose reads.  The pair ring / chunk pipeline is the one of spconv_wgrad_s16_tr; all four
// waves run the (identical) index bookkeeping redundantly, so the ring needs no synchronisation of its own.
template <int COUT>
__global__ __launch_bounds__(256) void spconv_wgrad_s16_coop128(const __bf16 *__restrict__ in, const __bf16 *__restrict__ dout,
                                                                const int32_t *__restrict__ nbr, int n_out, int kvol,
                                                                int rows_per_split, float *__restrict__ partial) {
    constexpr int CIN = 128, MA = CIN / 16, NB = COUT / 64;   // a wave: 128 ci x COUT/4 co
    constexpr int RS_A = CIN * 2 + 32, RS_B = COUT * 2 + 32;   // padded pixel-major rows (conflict-free transpose reads)
    constexpr int TILE_A = 32 * RS_A, TILE_B = 32 * RS_B;
    constexpr int PA = 32 * (CIN / 8) / 256, PB = 32 * (COUT / 8) / 256;   // 16-byte pieces per thread
    static_assert(PA >= 1 && PB >= 1 && NB >= 1, "tile shape");
    constexpr int RING = 256;
    __shared__ int2 ring[RING];
    __shared__ __attribute__((aligned(16))) char tiles[2][TILE_A + TILE_B];
    const int t = threadIdx.x, lane = t & 63;
    const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
    const int g = lane >> 4, c16 = lane & 15;
    int k, by;
    wgrad_block(kvol, k, by);
    const int r_begin = by * rows_per_split;
    const int r_end = min(n_out, r_begin + rows_per_split);
    const int32_t *nk = nbr + (int64_t)k * n_out;
    const unsigned tiles_addr = (unsigned)(size_t)((__attribute__((address_space(3))) char *)&tiles[0][0]);
    const int tr_off_a = (4 * g + (c16 >> 2)) * RS_A + (c16 & 3) * 8;
    const int tr_off_b = TILE_A + (4 * g + (c16 >> 2)) * RS_B + (c16 & 3) * 8 + wid * (COUT / 4) * 2;

    f32x4 acc[MA][NB];
#pragma unroll
    for (int m = 0; m < MA; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    struct Gather {
        uint4 va[PA], vb[PB];
    };
    int tail = 0, nbase = r_begin;
    auto load_idx = [&](int base) -> int { return base + lane < r_end ? nk[min(base + lane, n_out - 1)] : -1; };
    int jp0 = load_idx(nbase), jp1 = load_idx(nbase + 64);
    // the clamp target of an empty ring.  The ring is shared by the four waves, so this store must be ordered before ANY wave's first
    // top_up: a wave that starts a microsecond late (a second stream's kernel competing for the CU's wave slots - side.py) used to
    // put (0, 0) over the first real pair between an early wave's top_up and its first gather: one pair of one (offset, split)
    // replaced by rows (0, 0), found as an intermittent last-digit difference of conv4's weight gradients (r04 stress run)
    if (t == 0) ring[0] = make_int2(0, 0);
    __syncthreads();
    auto top_up = [&](int need) {   // every wave appends the same pairs to the same slots
        while (tail < need && nbase < r_end) {
            const int j_l = jp0;
            jp0 = jp1;
            jp1 = load_idx(nbase + 128);
            const unsigned long long mask = __ballot(j_l >= 0);
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
            if (j_l >= 0) ring[(tail + rank) & (RING - 1)] = make_int2(nbase + lane, j_l);
            tail += __popcll(mask);
            nbase += 64;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto gather = [&](int start, int cnt, Gather &gt) {
        const int last = cnt > 0 ? cnt - 1 : 0;
#pragma unroll
        for (int u = 0; u < PA; ++u) {
            const int c = t + 256 * u, pr = c / (CIN / 8), part = c % (CIN / 8);
            const int2 pp = ring[(start + min(pr, last)) & (RING - 1)];
            gt.va[u] = *reinterpret_cast<const uint4 *>(in + (int64_t)pp.y * CIN + part * 8);
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int c = t + 256 * u, pr = c / (COUT / 8), part = c % (COUT / 8);
            const int2 pp = ring[(start + min(pr, last)) & (RING - 1)];
            gt.vb[u] = *reinterpret_cast<const uint4 *>(dout + (int64_t)pp.x * COUT + part * 8);
        }
    };
    int parity = 0;
    auto consume = [&](Gather &gt, int cnt) {
        char *tl = tiles[parity];
#pragma unroll
        for (int u = 0; u < PA; ++u) {
            const int c = t + 256 * u, pr = c / (CIN / 8), part = c % (CIN / 8);
            const uint4 v = pr < cnt ? gt.va[u] : uint4{0u, 0u, 0u, 0u};
            *reinterpret_cast<uint4 *>(tl + pr * RS_A + part * 16) = v;
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int c = t + 256 * u, pr = c / (COUT / 8), part = c % (COUT / 8);
            const uint4 v = pr < cnt ? gt.vb[u] : uint4{0u, 0u, 0u, 0u};
            *reinterpret_cast<uint4 *>(tl + TILE_A + pr * RS_B + part * 16) = v;
        }
        __syncthreads();   // tile complete; also: every wave is done reading the other tile (it is overwritten next)
        const unsigned base = tiles_addr + parity * (TILE_A + TILE_B);
        wg_i32x2 blo[NB], bhi[NB], alo[MA], ahi[MA];
#pragma unroll
        for (int n = 0; n < NB; ++n) wg_tr_issue<16 * RS_B>(blo[n], bhi[n], base + tr_off_b + n * 32);
#pragma unroll
        for (int m = 0; m < MA; ++m) wg_tr_issue<16 * RS_A>(alo[m], ahi[m], base + tr_off_a + m * 32);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MA; ++m) {
            const bf16x8 a = wg_tr_pack(alo[m], ahi[m]);
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wg_tr_pack(blo[n], bhi[n]), acc[m][n], 0, 0, 0);
        }
        parity ^= 1;
    };
    int start = 0, cnt = 0;
    auto step = [&](Gather &cur, Gather &nxt) {
        const int nstart = start + cnt;
        top_up(nstart + 32);
        const int ncnt = min(32, tail - nstart);
        gather(ncnt > 0 ? nstart : max(tail - 1, 0), ncnt, nxt);
        __builtin_amdgcn_sched_barrier(0);
        consume(cur, cnt);
        __builtin_amdgcn_sched_barrier(0);
        start = nstart;
        cnt = ncnt;
    };
    Gather g0, g1;
    top_up(32);
    cnt = min(32, tail);
    gather(0, cnt, g0);
    while (cnt > 0) {   // wave-uniform AND identical in the four waves (same indices): the barriers inside match
        step(g0, g1);
        step(g1, g0);
    }
    float *dst = partial + ((int64_t)by * kvol + k) * CIN * COUT;
    const int co_base = wid * (COUT / 4);
#pragma unroll
    for (int m = 0; m < MA; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) dst[(int64_t)(16 * m + 4 * g + reg) * COUT + co_base + 16 * n + c16] = acc[m][n][reg];
}

struct WgradPlan {
    int mode;           // storage / MFMA variant, see wgrad_plan
    bool tr;            // bf16-storage transpose-read kernel (its wave layout differs)
    bool coop;          // bf16 storage, 128 input channels: workgroup-cooperative gather (one row split per workgroup)
    int n_split;        // total row splits
    int rows_per_split; // multiple of 4
    int grid_y;
    bool mfma;
    int wrow;
};

// mode: 0 fp32 MFMA, 1 bf16 MFMA on fp32 storage, 2 bf16 MFMA on bf16 storage (one per C-ABI entry point)
static int wgrad_tr_override() {   // S2D_WGRAD_TR=0/1 forces the register-assembled / transpose-read bf16-storage kernel (A/B runs)
    static const int v = [] {
        const char *e = getenv("S2D_WGRAD_TR");
        return e ? (e[0] == '1' ? 1 : 0) : 2;
    }();
    return v;
}

static WgradPlan wgrad_plan(int mode, int64_t n_out, int kvol, int cin, int cout) {
    WgradPlan p;
    p.mode = mode;
    const int tr_pref = wgrad_tr_override();
    p.mfma = (cin == 16 || cin == 32 || cin == 64 || cin == 128) && (cout == 16 || cout == 32 || cout == 64 || cout == 128);
    // measured (r01, bench scene): the transpose-read kernel wins for cin <= 64 (16->16 144 -> 97 us ... 64->128 81 -> 58 us,
    // launch + reduce included) and loses at cin = 128 (120 -> 131 us: four waves each re-gather the 256-byte rows)
    p.tr = mode == 2 && p.mfma && (tr_pref == 1 || (tr_pref == 2 && cin <= 64));
    static const int coop_env = [] { const char *e = getenv("S2D_WGRAD_COOP"); return e ? atoi(e) : 1; }();
    p.coop = mode == 2 && !p.tr && coop_env && cin == 128 && (cout == 128 || cout == 64);
    int vb = cout >= 32 ? 2 : 1;
    int wco = p.mfma ? cout / (16 * vb) : 1;
    if (p.tr) {
        const int nbmax = cin == 128 ? 2 : 4;
        const int nb = cout / 16 < nbmax ? cout / 16 : nbmax;
        wco = cout / (16 * nb);
    }
    p.wrow = p.mfma ? 4 / wco : 1;
    if (p.coop) p.wrow = 1;
    // aim at ~2048 waves in flight, at least 256 rows per split
    static const int waves_env = [] { const char *e = getenv("S2D_WGRAD_WAVES"); return e ? atoi(e) : 0; }();
    const int want_waves = waves_env > 0 ? waves_env : 2048;
    // rounded DOWN: the 128-channel kernel (174 VGPRs) fits two workgroups per CU = 512 on the chip, and 27 x 19 = 513
    // workgroups ran as two rounds
    int64_t want_blocks_y = want_waves / ((int64_t)kvol * 4);
    if (want_blocks_y < 1) want_blocks_y = 1;
    int64_t max_split = ceil_div(n_out > 0 ? n_out : 1, 256);
    if (p.tr && cin * cout <= 32 * 32) want_blocks_y *= 2;   // small register footprint: more resident workgroups
    if (want_blocks_y * p.wrow > max_split) want_blocks_y = max_split / p.wrow > 0 ? max_split / p.wrow : 1;
    // multiple of 8: XCD-local row blocks (wgrad_block); measured r01: 16->16 61 -> 40 us, 32->32 73 -> 50 us, neutral at
    // 64 channels, and the 128-channel register-assembled kernel prefers its 18 blocks (125 vs 132 us)
    if (p.tr && want_blocks_y >= 8) want_blocks_y &= ~(int64_t)7;
    int64_t splits = want_blocks_y * p.wrow;
    if (splits > max_split) splits = max_split;
    splits = ceil_div(splits, p.wrow) * p.wrow;
    p.n_split = (int)splits;
    p.grid_y = (int)(splits / p.wrow);
    p.rows_per_split = (int)(ceil_div(ceil_div(n_out > 0 ? n_out : 1, splits), 4) * 4);
    return p;
}

template <int CIN, int COUT>
static void launch_wgrad(const float *in, const float *dout, const int32_t *nbr, int n_out, int kvol, const WgradPlan &p,
                         float *partial, hipStream_t st) {
    if (p.mode == 2 && p.coop) {
        if constexpr (CIN == 128 && (COUT == 128 || COUT == 64)) {
            hipLaunchKernelGGL((spconv_wgrad_s16_coop128<COUT>), dim3(kvol, p.grid_y), dim3(256), 0, st, (const __bf16 *)in,
                               (const __bf16 *)dout, nbr, n_out, kvol, p.rows_per_split, partial);
            return;
        }
    }
    if (p.mode == 2 && p.tr) {   // bf16 storage: 16-byte gathers + per-wave LDS tile + transpose reads
        hipLaunchKernelGGL((spconv_wgrad_s16_tr<CIN, COUT>), dim3(kvol, p.grid_y), dim3(256), 0, st, (const __bf16 *)in,
                           (const __bf16 *)dout, nbr, n_out, kvol, p.rows_per_split, partial);
        return;
    }
    if (p.mode == 2)
        hipLaunchKernelGGL((spconv_wgrad_bf16<CIN, COUT, __bf16>), dim3(kvol, p.grid_y), dim3(256), 0, st, (const __bf16 *)in,
                           (const __bf16 *)dout, nbr, n_out, kvol, p.rows_per_split, partial);
    else if (p.mode == 1)
        hipLaunchKernelGGL((spconv_wgrad_bf16<CIN, COUT, float>), dim3(kvol, p.grid_y), dim3(256), 0, st, in, dout, nbr, n_out,
                           kvol, p.rows_per_split, partial);
    else
        hipLaunchKernelGGL((spconv_wgrad_mfma<CIN, COUT>), dim3(kvol, p.grid_y), dim3(256), 0, st, in, dout, nbr, n_out,
                           kvol, p.rows_per_split, partial);
}

template <int CIN>
static bool dispatch_wgrad_cout(const float *in, const float *dout, const int32_t *nbr, int n_out, int kvol, int cout,
                                const WgradPlan &p, float *partial, hipStream_t st) {
    switch (cout) {
        case 16: launch_wgrad<CIN, 16>(in, dout, nbr, n_out, kvol, p, partial, st); return true;
        case 32: launch_wgrad<CIN, 32>(in, dout, nbr, n_out, kvol, p, partial, st); return true;
        case 64: launch_wgrad<CIN, 64>(in, dout, nbr, n_out, kvol, p, partial, st); return true;
        case 128: launch_wgrad<CIN, 128>(in, dout, nbr, n_out, kvol, p, partial, st); return true;
        default: return false;
    }
}

}  // namespace s2d

using namespace s2d;

extern "C" int s2d_spconv_fwd_f32(const float *in_feat, int64_t n_in, const float *weight, const float *bias,
                                  const int32_t *nbr, int64_t n_out, int kvol, int cin, int cout, float *out_feat,
                                  s2d_stream_t stream) {
    S2D_CHECK_ARG(n_in >= 0 && n_out >= 0 && n_out < 0x7fffffff && kvol > 0 && cin > 0 && cout > 0, "spconv_fwd: bad sizes");
    S2D_CHECK_ARG(weight && (n_out == 0 || (nbr && out_feat)) && (n_in == 0 || in_feat), "spconv_fwd: null argument");
    if (n_out == 0) return S2D_OK;
    hipStream_t st = (hipStream_t)stream;
    int rc = -1;
    const bool cout_ok = cout == 16 || cout == 32 || cout == 64 || cout == 128;
    if (cout_ok && n_in > 0) {
        switch (cin) {
            case 16: rc = dispatch_fwd_cout<16>(in_feat, weight, bias, nbr, (int)n_out, kvol, cout, out_feat, st); break;
            case 32: rc = dispatch_fwd_cout<32>(in_feat, weight, bias, nbr, (int)n_out, kvol, cout, out_feat, st); break;
            case 64: rc = dispatch_fwd_cout<64>(in_feat, weight, bias, nbr, (int)n_out, kvol, cout, out_feat, st); break;
            case 128: rc = dispatch_fwd_cout<128>(in_feat, weight, bias, nbr, (int)n_out, kvol, cout, out_feat, st); break;
            default: break;
        }
        if (rc > 0) return rc;
    }
    if (rc != 0) {
        const int64_t threads = n_out * cout;
        hipLaunchKernelGGL(spconv_fwd_valu, dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0, st, in_feat, weight, bias,
                           nbr, (int)n_out, kvol, cin, cout, out_feat);
        S2D_LAUNCH_CHECK();
    }
    return S2D_OK;
}

extern "C" size_t s2d_spconv_wgrad_workspace_bytes(int64_t n_out, int kvol, int cin, int cout) {
    if (n_out < 0 || kvol <= 0 || cin <= 0 || cout <= 0) return 0;
    // one size for every entry point: the bf16-storage kernels split the rows differently from the fp32-storage ones
    size_t need = 0;
    for (int mode = 0; mode <= 2; mode += 2) {
        const WgradPlan p = wgrad_plan(mode, n_out, kvol, cin, cout);
        const size_t b = (size_t)p.n_split * kvol * cin * cout * sizeof(float);
        need = b > need ? b : need;
    }
    return align_up(need, 256);
}

static int wgrad_impl(int mode, const float *in_feat, int64_t n_in, const float *dout, const int32_t *nbr, int64_t n_out, int kvol,
                      int cin, int cout, float *dweight, void *ws, size_t ws_bytes, s2d_stream_t stream);

extern "C" int s2d_spconv_wgrad_f32(const float *in_feat, int64_t n_in, const float *dout, const int32_t *nbr,
                                    int64_t n_out, int kvol, int cin, int cout, float *dweight, void *ws, size_t ws_bytes,
                                    s2d_stream_t stream) {
    return wgrad_impl(0, in_feat, n_in, dout, nbr, n_out, kvol, cin, cout, dweight, ws, ws_bytes, stream);
}

extern "C" int s2d_spconv_wgrad_bf16(const float *in_feat, int64_t n_in, const float *dout, const int32_t *nbr,
                                     int64_t n_out, int kvol, int cin, int cout, float *dweight, void *ws, size_t ws_bytes,
                                     s2d_stream_t stream) {
    return wgrad_impl(1, in_feat, n_in, dout, nbr, n_out, kvol, cin, cout, dweight, ws, ws_bytes, stream);
}

// bf16-storage ("s16") weight gradient: in_feat / dout are bf16 [n][c]; cin, cout in {16,32,64,128}
extern "C" int s2d_spconv_s16_wgrad(const void *in_feat, int64_t n_in, const void *dout, const int32_t *nbr, int64_t n_out,
                                    int kvol, int cin, int cout, float *dweight, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    const bool ok = (cin == 16 || cin == 32 || cin == 64 || cin == 128) && (cout == 16 || cout == 32 || cout == 64 || cout == 128);
    if (!ok) {
        set_error("spconv_s16_wgrad: unsupported channels %d -> %d", cin, cout);
        return S2D_ERR_UNSUPPORTED;
    }
    return wgrad_impl(2, (const float *)in_feat, n_in, (const float *)dout, nbr, n_out, kvol, cin, cout, dweight, ws, ws_bytes,
                      stream);
}

static int wgrad_impl(int mode, const float *in_feat, int64_t n_in, const float *dout, const int32_t *nbr, int64_t n_out, int kvol,
                      int cin, int cout, float *dweight, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(n_in >= 0 && n_out >= 0 && n_out < 0x7fffffff && kvol > 0 && cin > 0 && cout > 0, "spconv_wgrad: bad sizes");
    S2D_CHECK_ARG(dweight, "spconv_wgrad: null dweight");
    hipStream_t st = (hipStream_t)stream;
    const int64_t size = (int64_t)kvol * cin * cout;
    if (n_out == 0 || n_in == 0) {
        S2D_HIP(hipMemsetAsync(dweight, 0, size * sizeof(float), st));
        return S2D_OK;
    }
    S2D_CHECK_ARG(in_feat && dout && nbr, "spconv_wgrad: null argument");
    WgradPlan p = wgrad_plan(mode, n_out, kvol, cin, cout);
    const size_t need = (size_t)p.n_split * size * sizeof(float);
    if (!ws || ws_bytes < need) {
        set_error("spconv_wgrad: workspace too small (%zu < %zu)", ws_bytes, need);
        return S2D_ERR_WORKSPACE;
    }
    float *partial = (float *)ws;
    bool done = false;
    if (p.mfma) {
        switch (cin) {
            case 16: done = dispatch_wgrad_cout<16>(in_feat, dout, nbr, (int)n_out, kvol, cout, p, partial, st); break;
            case 32: done = dispatch_wgrad_cout<32>(in_feat, dout, nbr, (int)n_out, kvol, cout, p, partial, st); break;
            case 64: done = dispatch_wgrad_cout<64>(in_feat, dout, nbr, (int)n_out, kvol, cout, p, partial, st); break;
            case 128: done = dispatch_wgrad_cout<128>(in_feat, dout, nbr, (int)n_out, kvol, cout, p, partial, st); break;
            default: break;
        }
    }
    if (!done) {
        hipLaunchKernelGGL(spconv_wgrad_valu, dim3(kvol, p.n_split), dim3(128), 0, st, in_feat, dout, nbr, (int)n_out, kvol,
                           cin, cout, p.rows_per_split, partial);
    }
    if (size % 4 == 0)
        hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3((unsigned)ceil_div(size / 4, 16)), dim3(256), 0, st, (const float4 *)partial,
                           p.n_split, size / 4, (float4 *)dweight);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)ceil_div(size, 256)), dim3(256), 0, st, partial, p.n_split, size,
                           dweight);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
