// Loss kernels of the S2D student step that would otherwise stream dense ground-truth volumes through HBM.
//
// PCR (point-cloud reconstruction) losses, /root/reference/det3d/models/detectors/voxelnet.py:171-185,203-249:
//   gt      = SparseConvTensor(recon voxel means [M,5], coors, [D,H,W]).dense()        -> [B,5,D,H,W] (226 MB / frame at 2x voxels)
//   gt_mask = gt.sum(1) != 0 ; beta = #neg / #pos
//   mask_loss   = BCEWithLogits(gen_mask[:,0], gt_mask, pos_weight=beta)                (mean over all B*D*H*W cells)
//   tgt         = gt[:, :3] - grid * gt_mask ; sel = tgt != 0
//   offset_loss = L1(gen_offset[sel], tgt[sel])                                          (mean over selected entries)
// The dense GT is zero outside the M occupied cells, so both losses split into ONE dense reduction over the predicted
// occupancy logits plus sparse gathers at the M recon voxels; the GT volume, its mask, the metric grid and `tgt` are never
// materialised:
//   mask_loss   = [ sum_all softplus(x) + sum_pos (beta*softplus(-x) - softplus(x)) ] / N
//   offset_loss = sum_{m, c<3, t != 0} |gen_offset[b,c,cell_m] - t| / n_sel ,   t = mean_c - grid_c(cell)  (pos cells)
// grid = cell-centre metric coordinates exactly as voxelnet.py:232-236 writes them (the x step reuses 150.4/H).
// All reductions are two-stage with a fixed fold order (deterministic).
#include "s2d_common.h"
#include <algorithm>

namespace s2d {

__device__ __forceinline__ float softplusf(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// block-wide sum of K floats per thread -> out[blockIdx][K] (fixed order: lanes via DPP-free shuffles, then waves in order)
template <int K>
__device__ __forceinline__ void block_sums(float (&v)[K], float *out) {
    __shared__ float red[4][K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float s = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < K) out[(int64_t)blockIdx.x * K + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) +
                                                                    (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void pcr_softplus_sum_kernel(const float *__restrict__ x, int64_t n4, int64_t n, float *__restrict__ partial) {
    float acc[1] = {0.f};
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        acc[0] += (softplusf(v.x) + softplusf(v.y)) + (softplusf(v.z) + softplusf(v.w));
    }
    if (blockIdx.x == 0)
        for (int64_t i = 4 * n4 + threadIdx.x; i < n; i += 256) acc[0] += softplusf(x[i]);
    block_sums<1>(acc, partial);
}

struct PcrGeo {
    int batch, d, h, w;
};

__device__ __forceinline__ void pcr_grid(const PcrGeo &g, int z, int y, int x, float (&c)[3]) {
    // voxelnet.py:232-236 in fp32, operation by operation: xs*(150.4/w) - 75.2 + (150.4/h)/2 etc.
    const float sx = (float)(150.4 / g.w), sy = (float)(150.4 / g.h), sz = (float)(6.0 / g.d);
    const float hx = (float)((150.4 / g.h) / 2), hz = (float)((6.0 / g.d) / 2);
    c[0] = ((float)x * sx - 75.2f) + hx;
    c[1] = ((float)y * sy - 75.2f) + hx;
    c[2] = ((float)z * sz - 2.f) + hz;
}

// per recon voxel: [n_pos, sum_pos softplus(-x), sum_pos softplus(x), L1 sum, n_sel]
__global__ __launch_bounds__(256) void pcr_sparse_terms_kernel(const int32_t *__restrict__ coors, const float *__restrict__ feats, int64_t m,
                                                               PcrGeo g, const float *__restrict__ gen_mask,
                                                               const float *__restrict__ gen_offset, float *__restrict__ partial) {
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const int64_t cells = (int64_t)g.d * g.h * g.w;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += stride) {
        const int4 c = reinterpret_cast<const int4 *>(coors)[i];   // b,z,y,x
        if ((unsigned)c.x >= (unsigned)g.batch || (unsigned)c.y >= (unsigned)g.d || (unsigned)c.z >= (unsigned)g.h ||
            (unsigned)c.w >= (unsigned)g.w)
            continue;
        const float *f = feats + i * 5;
        const float s = (((f[0] + f[1]) + f[2]) + f[3]) + f[4];
        const bool pos = s != 0.f;
        const int64_t cell = ((int64_t)c.y * g.h + c.z) * g.w + c.w;
        if (pos) {
            const float x = gen_mask[(int64_t)c.x * cells + cell];
            acc[0] += 1.f;
            acc[1] += softplusf(-x);
            acc[2] += softplusf(x);
        }
        float gc[3];
        pcr_grid(g, c.y, c.z, c.w, gc);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float t = pos ? f[k] - gc[k] : f[k];
            if (t != 0.f) {
                acc[3] += fabsf(gen_offset[((int64_t)c.x * 3 + k) * cells + cell] - t);
                acc[4] += 1.f;
            }
        }
    }
    block_sums<5>(acc, partial);
}

// out[0] = mask_loss, out[1] = offset_loss, out[2] = beta, out[3] = n_sel, out[4] = N, out[5] = n_pos
__global__ __launch_bounds__(64) void pcr_finalize_kernel(const float *__restrict__ dense_partial, int nd, const float *__restrict__ sparse_partial, int ns,
                                                          double n_cells, float *__restrict__ out) {
    // one wave: lanes stride over the partial rows in double, fixed-order shuffle fold
    double v[6] = {0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < nd; i += 64) v[5] += dense_partial[i];
    for (int i = threadIdx.x; i < ns; i += 64)
        for (int k = 0; k < 5; ++k) v[k] += sparse_partial[i * 5 + k];
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double s_all = v[5];
    const double *t = v;
    const double npos = t[0], nneg = n_cells - npos;
    const float beta = (float)(nneg / npos);          // count_neg / count_pos (inf / nan for an empty target, as in torch)
    out[0] = (float)((s_all - t[2] + (double)beta * t[1]) / n_cells);
    out[1] = (float)(t[3] / t[4]);
    out[2] = beta;
    out[3] = (float)t[4];
    out[4] = (float)n_cells;
    out[5] = (float)npos;
}

// d mask_loss / d x = sigmoid(x)/N for every cell (occupied cells are overwritten by the sparse pass)
__global__ __launch_bounds__(256) void pcr_mask_grad_dense_kernel(const float *__restrict__ x, const float *__restrict__ go, const float *__restrict__ fin,
                                                                  int64_t n4, int64_t n, float *__restrict__ gx) {
    const float scale = go[0] / fin[4];
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        reinterpret_cast<float4 *>(gx)[i] = float4{scale * sigmoidf(v.x), scale * sigmoidf(v.y), scale * sigmoidf(v.z), scale * sigmoidf(v.w)};
    }
    if (blockIdx.x == 0)
        for (int64_t i = 4 * n4 + threadIdx.x; i < n; i += 256) gx[i] = scale * sigmoidf(x[i]);
}

__global__ __launch_bounds__(256) void pcr_sparse_grad_kernel(const int32_t *__restrict__ coors, const float *__restrict__ feats, int64_t m, PcrGeo g,
                                                              const float *__restrict__ gen_mask, const float *__restrict__ gen_offset,
                                                              const float *__restrict__ go_mask, const float *__restrict__ go_off,
                                                              const float *__restrict__ fin, float *__restrict__ g_mask,
                                                              float *__restrict__ g_off) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const int4 c = reinterpret_cast<const int4 *>(coors)[i];
    if ((unsigned)c.x >= (unsigned)g.batch || (unsigned)c.y >= (unsigned)g.d || (unsigned)c.z >= (unsigned)g.h || (unsigned)c.w >= (unsigned)g.w)
        return;
    const int64_t cells = (int64_t)g.d * g.h * g.w;
    const float *f = feats + i * 5;
    const float s = (((f[0] + f[1]) + f[2]) + f[3]) + f[4];
    const bool pos = s != 0.f;
    const int64_t cell = ((int64_t)c.y * g.h + c.z) * g.w + c.w;
    if (pos && g_mask) {
        const float x = gen_mask[(int64_t)c.x * cells + cell];
        g_mask[(int64_t)c.x * cells + cell] = -(go_mask[0] / fin[4]) * fin[2] * (1.f - sigmoidf(x));
    }
    if (g_off) {
        float gc[3];
        pcr_grid(g, c.y, c.z, c.w, gc);
        const float sc = go_off[0] / fin[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float t = pos ? f[k] - gc[k] : f[k];
            if (t != 0.f) {
                const int64_t at = ((int64_t)c.x * 3 + k) * cells + cell;
                const float dlt = gen_offset[at] - t;
                g_off[at] = dlt > 0.f ? sc : (dlt < 0.f ? -sc : 0.f);
            }
        }
    }
}

constexpr int PCR_DENSE_BLOCKS = 1024, PCR_SPARSE_BLOCKS = 256;

}  // namespace s2d

using namespace s2d;

extern "C" size_t s2d_pcr_loss_workspace_bytes(void) { return (size_t)(PCR_DENSE_BLOCKS + 5 * PCR_SPARSE_BLOCKS) * sizeof(float) + 512; }

extern "C" int s2d_pcr_loss_fwd_f32(const float *gen_offset, const float *gen_mask, const int32_t *coors, const float *feats,
                                    int64_t m, int batch, int d, int h, int w, float *out8, void *ws, size_t ws_bytes,
                                    s2d_stream_t stream) {
    S2D_CHECK_ARG(gen_offset && gen_mask && out8 && batch > 0 && d > 0 && h > 0 && w > 0 && m >= 0 && (m == 0 || (coors && feats)),
                  "pcr_loss_fwd: bad argument");
    if (!ws || ws_bytes < s2d_pcr_loss_workspace_bytes()) {
        set_error("pcr_loss_fwd: workspace too small");
        return S2D_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float *dense_partial = (float *)ws, *sparse_partial = dense_partial + PCR_DENSE_BLOCKS;
    const int64_t n = (int64_t)batch * d * h * w;
    const int nd = (int)std::min<int64_t>(PCR_DENSE_BLOCKS, std::max<int64_t>(1, ceil_div(n / 4, 256)));
    const int ns = (int)std::min<int64_t>(PCR_SPARSE_BLOCKS, std::max<int64_t>(1, ceil_div(m, 256)));
    hipLaunchKernelGGL(pcr_softplus_sum_kernel, dim3(nd), dim3(256), 0, st, gen_mask, n / 4, n, dense_partial);
    PcrGeo g{batch, d, h, w};
    hipLaunchKernelGGL(pcr_sparse_terms_kernel, dim3(ns), dim3(256), 0, st, coors, feats, m, g, gen_mask, gen_offset, sparse_partial);
    hipLaunchKernelGGL(pcr_finalize_kernel, dim3(1), dim3(64), 0, st, dense_partial, nd, sparse_partial, ns, (double)n, out8);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_pcr_loss_bwd_f32(const float *gen_offset, const float *gen_mask, const int32_t *coors, const float *feats,
                                    int64_t m, int batch, int d, int h, int w, const float *fwd_out8, const float *go_mask,
                                    const float *go_offset, float *g_gen_mask, float *g_gen_offset_zeroed, s2d_stream_t stream) {
    S2D_CHECK_ARG(gen_offset && gen_mask && fwd_out8 && go_mask && go_offset && batch > 0 && d > 0 && h > 0 && w > 0 && m >= 0,
                  "pcr_loss_bwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)batch * d * h * w;
    if (g_gen_mask) {
        const int nd = (int)std::min<int64_t>(4096, std::max<int64_t>(1, ceil_div(n / 4, 256)));
        hipLaunchKernelGGL(pcr_mask_grad_dense_kernel, dim3(nd), dim3(256), 0, st, gen_mask, go_mask, fwd_out8, n / 4, n, g_gen_mask);
    }
    if (m > 0 && (g_gen_mask || g_gen_offset_zeroed)) {
        PcrGeo g{batch, d, h, w};
        hipLaunchKernelGGL(pcr_sparse_grad_kernel, dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, st, coors, feats, m, g, gen_mask, gen_offset,
                           go_mask, go_offset, fwd_out8, g_gen_mask, g_gen_offset_zeroed);
    }
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

// =====================================================================================================================
// Fused PCR level heads: gen_mask_k / gen_out_k (1x1x1 Conv3d C->1 and C->3, rpn.py:273-275,292-294) + their two losses
// evaluated straight from the level's feature volume g[B][C][S] (S = D*H*W cells):
//   * the occupancy logits are never written: one dense pass computes x = w_mask.g + b and accumulates softplus(x);
//   * the offsets are only ever looked at in the occupied target cells (sel = tgt != 0 is empty elsewhere), so the C->3 conv
//     runs at the M recon voxels only;
//   * backward: ONE dense pass writes dg = w_mask * dL/dx (+ w2^T . dz, the data gradient of the level's next 1x1x1 conv when the
//     caller hands its output gradient in) and the dense part of dw_mask/db_mask; a sparse pass adds the occupied-cell
//     corrections and the offset-head terms in place (recon voxels are unique cells: no atomics).
// Compared with conv -> loss kernels this removes the logits / offsets / zero-filled offset gradient volumes and every separate
// data-gradient accumulation over g (at the 2x level: 7 passes over 181-543 MB tensors -> 3).
// =====================================================================================================================
namespace s2d {

template <int C>
struct PcrHeadW {
    float wm[C];
    float wo[3][C];
    float bm;
    float bo[3];
};

template <int V> struct VecOf;
template <> struct VecOf<2> { using T = float2; };
template <> struct VecOf<4> { using T = float4; };

template <int V> __device__ __forceinline__ void vec_load(const float *p, float (&v)[V]) {
    const typename VecOf<V>::T t = *reinterpret_cast<const typename VecOf<V>::T *>(p);
    const float *f = reinterpret_cast<const float *>(&t);
#pragma unroll
    for (int k = 0; k < V; ++k) v[k] = f[k];
}
template <int V> __device__ __forceinline__ void vec_store(float *p, const float (&v)[V]) {
    typename VecOf<V>::T t;
    float *f = reinterpret_cast<float *>(&t);
#pragma unroll
    for (int k = 0; k < V; ++k) f[k] = v[k];
    *reinterpret_cast<typename VecOf<V>::T *>(p) = t;
}

// head parameters (device): w_mask[C] | w_off[3][C] | b_mask | b_off[3] = the memory image of PcrHeadW<C>; every block keeps a copy in LDS
template <int C>
__device__ __forceinline__ void pcr_load_head(const float *__restrict__ hp, PcrHeadW<C> &hw) {
    float *dst = reinterpret_cast<float *>(&hw);
    for (int i = threadIdx.x; i < 4 * C + 4; i += 256) dst[i] = hp[i];
    __syncthreads();
}

// planar tensors through buffer instructions (s2d_common.h): V consecutive floats of one plane
__device__ __forceinline__ __amdgpu_buffer_rsrc_t planes_rsrc(const void *base, unsigned bytes) { return buf_rsrc(base, bytes); }
template <int V> __device__ __forceinline__ void buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float (&v)[V]) {
    if constexpr (V == 2) {
        const buf_f32x2 t = buf_load2(r, voff, soff);
        v[0] = t[0]; v[1] = t[1];
    } else {
        const buf_f32x4 t = buf_load4(r, voff, soff);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = t[k];
    }
}
template <int V> __device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, const float (&v)[V]) {
    if constexpr (V == 2) {
        buf_store2(r, voff, soff, buf_f32x2{v[0], v[1]});
    } else {
        buf_store4(r, voff, soff, buf_f32x4{v[0], v[1], v[2], v[3]});
    }
}

// r04: the RAW up-sampler outputs y (724 MB / 543 MB in fp32 at batch 4, read by four passes per level) may be stored in bf16: the level
// kernels take the element type of y as a template parameter; everything they write (z, dy, sums) stays fp32
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t hi16) { return __builtin_bit_cast(float, hi16 << 16); }
template <int V, typename T> __device__ __forceinline__ void buf_load_t(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float (&v)[V]) {
    if constexpr (sizeof(T) == 4) {
        buf_load<V>(r, voff, soff, v);
    } else if constexpr (V == 2) {
        const uint32_t t = __builtin_bit_cast(uint32_t, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
        v[0] = bf16_bits_to_f32(t & 0xFFFFu); v[1] = bf16_bits_to_f32(t >> 16);
    } else {
        const uint64_t t2 = __builtin_bit_cast(uint64_t, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
        const uint32_t a = (uint32_t)t2, b = (uint32_t)(t2 >> 32);
        v[0] = bf16_bits_to_f32(a & 0xFFFFu); v[1] = bf16_bits_to_f32(a >> 16);
        v[2] = bf16_bits_to_f32(b & 0xFFFFu); v[3] = bf16_bits_to_f32(b >> 16);
    }
}
// V consecutive elements stored as T (fp32 or round-to-nearest-even bf16, two per dword)
template <int V, typename T> __device__ __forceinline__ void buf_store_t(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, const float (&v)[V]) {
    if constexpr (sizeof(T) == 4) {
        buf_store<V>(r, voff, soff, v);
    } else {
        uint32_t w[V / 2];
#pragma unroll
        for (int k = 0; k < V / 2; ++k) {
            const __bf16 lo = (__bf16)v[2 * k], hi = (__bf16)v[2 * k + 1];
            w[k] = (uint32_t)__builtin_bit_cast(uint16_t, lo) | ((uint32_t)__builtin_bit_cast(uint16_t, hi) << 16);
        }
        if constexpr (V == 2) buf_store1(r, voff, soff, __builtin_bit_cast(float, w[0]));
        else buf_store2(r, voff, soff, __builtin_bit_cast(buf_f32x2, ((uint64_t)w[1] << 32) | w[0]));
    }
}
// the V consecutive elements of C planes held the way they were loaded: fp32 as floats, bf16 PACKED (two per register) and unpacked at
// every use - unpacking at the load kept both forms live and cost the 32-channel backward kernels their occupancy (166 -> 257 registers)
template <int C, int V, typename T> struct YPack;
template <int C, int V> struct YPack<C, V, float> {
    float v[C][V];
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, int c) { buf_load<V>(r, voff, soff, v[c]); }
    __device__ __forceinline__ float get(int c, int k) const { return v[c][k]; }
};
template <int C, int V> struct YPack<C, V, __bf16> {
    uint32_t w[C][V / 2];
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, int c) {
        if constexpr (V == 2) {
            w[c][0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
        } else {
            const uint64_t t2 = __builtin_bit_cast(uint64_t, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
            w[c][0] = (uint32_t)t2;
            w[c][1] = (uint32_t)(t2 >> 32);
        }
    }
    __device__ __forceinline__ float get(int c, int k) const {
        uint32_t x = w[c][k >> 1];
        asm volatile("" : "+v"(x));   // opaque: the unpack is redone at each use instead of being hoisted next to the load
        return __builtin_bit_cast(float, (k & 1) ? (x & 0xFFFF0000u) : (x << 16));
    }
};

template <typename T> __device__ __forceinline__ float plane_elem(const T *p) {
    if constexpr (sizeof(T) == 4) return *p;
    else return (float)*p;
}

template <int K>
__device__ __forceinline__ void block_sums_n(float (&v)[K], float *out) {
    __shared__ float red[4][K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float s = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = s;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 256) out[(int64_t)blockIdx.x * K + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
}

template <int C, int V>
__global__ __launch_bounds__(256) void pcr_heads_fwd_dense_kernel(const float *__restrict__ g, const float *__restrict__ hp, int64_t cells, int batch,
                                                                  float *__restrict__ partial) {
    __shared__ PcrHeadW<C> hw;
    pcr_load_head<C>(hp, hw);
    float acc[1] = {0.f};
    const uint32_t sv = (uint32_t)(cells / V), stride = gridDim.x * 256u;
    const unsigned plane = (unsigned)cells * 4u;   // bytes per channel plane (C * plane < 4 GB: checked on the host)
    for (int b = 0; b < batch; ++b) {
    const __amdgpu_buffer_rsrc_t gr = planes_rsrc(g + (int64_t)b * C * cells, C * plane);
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < sv; j += stride) {
        float x[V];
#pragma unroll
        for (int k = 0; k < V; ++k) x[k] = hw.bm;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float v[V];
            buf_load<V>(gr, j * (V * 4u), c * plane, v);
#pragma unroll
            for (int k = 0; k < V; ++k) x[k] = fmaf(hw.wm[c], v[k], x[k]);
        }
#pragma unroll
        for (int k = 0; k < V; ++k) acc[0] += softplusf(x[k]);
    }
    }
    block_sums<1>(acc, partial);
}

// the site's logit and offsets from its C feature values
template <int C>
__device__ __forceinline__ void pcr_site_eval(const float *__restrict__ g, const PcrHeadW<C> &hw, int64_t cells, int b, int64_t cell, float (&gv)[C],
                                              float &x, float (&off)[3]) {
    x = hw.bm;
    off[0] = hw.bo[0]; off[1] = hw.bo[1]; off[2] = hw.bo[2];
    const float *base = g + (int64_t)b * C * cells + cell;
#pragma unroll
    for (int c = 0; c < C; ++c) gv[c] = base[(int64_t)c * cells];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        x = fmaf(hw.wm[c], gv[c], x);
#pragma unroll
        for (int k = 0; k < 3; ++k) off[k] = fmaf(hw.wo[k][c], gv[c], off[k]);
    }
}

template <int C>
__global__ __launch_bounds__(256) void pcr_heads_fwd_sparse_kernel(const int32_t *__restrict__ coors, const float *__restrict__ feats, int64_t m,
                                                                   PcrGeo geo, const float *__restrict__ g, const float *__restrict__ hp,
                                                                   float *__restrict__ partial) {
    __shared__ PcrHeadW<C> hw;
    pcr_load_head<C>(hp, hw);
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const int64_t cells = (int64_t)geo.d * geo.h * geo.w;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += stride) {
        const int4 c = reinterpret_cast<const int4 *>(coors)[i];   // b,z,y,x
        if ((unsigned)c.x >= (unsigned)geo.batch || (unsigned)c.y >= (unsigned)geo.d || (unsigned)c.z >= (unsigned)geo.h ||
            (unsigned)c.w >= (unsigned)geo.w)
            continue;
        const float *f = feats + i * 5;
        const float s = (((f[0] + f[1]) + f[2]) + f[3]) + f[4];
        const bool pos = s != 0.f;
        const int64_t cell = ((int64_t)c.y * geo.h + c.z) * geo.w + c.w;
        float gv[C], x, off[3];
        pcr_site_eval<C>(g, hw, cells, c.x, cell, gv, x, off);
        if (pos) {
            acc[0] += 1.f;
            acc[1] += softplusf(-x);
            acc[2] += softplusf(x);
        }
        float gc[3];
        pcr_grid(geo, c.y, c.z, c.w, gc);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float t = pos ? f[k] - gc[k] : f[k];
            if (t != 0.f) {
                acc[3] += fabsf(off[k] - t);
                acc[4] += 1.f;
            }
        }
    }
    block_sums<5>(acc, partial);
}

template <int C, int CO, int V>
__global__ __launch_bounds__(256) void pcr_heads_bwd_dense_kernel(const float *__restrict__ g, const float *__restrict__ dz,
                                                                  const float *__restrict__ w2, const float *__restrict__ hp, const float *__restrict__ go_mask,
                                                                  const float *__restrict__ fin, int64_t cells, int batch, float *__restrict__ dg,
                                                                  float *__restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float w2s[CO > 0 ? CO * C : 4];   // (read as float4 rows)
    __shared__ PcrHeadW<C> hw;
    pcr_load_head<C>(hp, hw);
    if (CO > 0) {
        for (int i = threadIdx.x; i < CO * C; i += 256) w2s[(i % C) * CO + i / C] = w2[i];   // [C][CO]
        __syncthreads();
    }
    const float scale = go_mask[0] / fin[4];
    // an opaque per-lane zero keeps the weight reads ordinary vector LDS loads with immediate offsets: known-uniform addresses are
    // scalarised (v_readfirstlane into ~550 SGPRs, which then spill into VGPR lanes: 256 VGPRs, one wave per SIMD, 1.4 ms)
    int lane_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    const float *wmv = reinterpret_cast<const float *>(&hw) + lane_zero;   // w_mask[C] leads the struct image
    const float *w2v = w2s + lane_zero;
    float pw[C + 1];
#pragma unroll
    for (int c = 0; c <= C; ++c) pw[c] = 0.f;
    // batch index in the outer (uniform) loop: plane bases stay scalar, lanes carry one 32-bit offset (per-lane 64-bit plane
    // addresses cost 2 VGPRs per plane: 160 for the 32 + 16 + 32 planes of this kernel)
    const uint32_t sv = (uint32_t)(cells / V), stride = gridDim.x * 256u;
    const unsigned plane = (unsigned)cells * 4u;
    for (int b = 0; b < batch; ++b) {
    const __amdgpu_buffer_rsrc_t gr = planes_rsrc(g + (int64_t)b * C * cells, C * plane), dgr = planes_rsrc(dg + (int64_t)b * C * cells, C * plane),
                                 zr = planes_rsrc(CO > 0 ? dz + (int64_t)b * CO * cells : g, (CO > 0 ? CO : C) * plane);
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < sv; j += stride) {
        asm volatile("" ::: "memory");   // and keep them inside the loop (hoisted, 512 + C weights would live in registers)
        const unsigned voff = j * (V * 4u);
        float x[V], dm[V];
#pragma unroll
        for (int k = 0; k < V; ++k) x[k] = hw.bm;
        float gv[C][V];
#pragma unroll
        for (int c = 0; c < C; ++c) buf_load<V>(gr, voff, c * plane, gv[c]);
        float zv[CO > 0 ? CO : 1][V];
        if (CO > 0) {
#pragma unroll
            for (int q = 0; q < CO; ++q) buf_load<V>(zr, voff, q * plane, zv[q]);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float wc = wmv[c];
#pragma unroll
            for (int k = 0; k < V; ++k) x[k] = fmaf(wc, gv[c][k], x[k]);
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            dm[k] = scale * sigmoidf(x[k]);
            pw[C] += dm[k];
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float o[V];
            const float wc = wmv[c];
#pragma unroll
            for (int k = 0; k < V; ++k) {
                pw[c] = fmaf(dm[k], gv[c][k], pw[c]);
                o[k] = wc * dm[k];
            }
            if constexpr (CO > 0 && CO % 4 == 0) {   // four weights per 16-byte LDS broadcast (see pcr_level_bwd_dense_kernel)
#pragma unroll
                for (int q4 = 0; q4 < CO / 4; ++q4) {
                    const float4 wq = *reinterpret_cast<const float4 *>(w2v + c * CO + 4 * q4);
#pragma unroll
                    for (int k = 0; k < V; ++k) {
                        o[k] = fmaf(wq.x, zv[4 * q4][k], o[k]);
                        o[k] = fmaf(wq.y, zv[4 * q4 + 1][k], o[k]);
                        o[k] = fmaf(wq.z, zv[4 * q4 + 2][k], o[k]);
                        o[k] = fmaf(wq.w, zv[4 * q4 + 3][k], o[k]);
                    }
                }
            } else if (CO > 0) {
#pragma unroll
                for (int q = 0; q < CO; ++q) {
                    const float wq = w2v[c * CO + q];
#pragma unroll
                    for (int k = 0; k < V; ++k) o[k] = fmaf(wq, zv[q][k], o[k]);
                }
            }
            buf_store<V>(dgr, voff, c * plane, o);
        }
    }
    }
    block_sums_n<C + 1>(pw, partial);
}

// per recon voxel: corrections of dg in place + partial sums [dw_mask(C) | dw_off(3C) | db_mask | db_off(3)]
template <int C>
__global__ __launch_bounds__(256) void pcr_heads_bwd_sparse_kernel(const int32_t *__restrict__ coors, const float *__restrict__ feats, int64_t m,
                                                                   PcrGeo geo, const float *__restrict__ g, const float *__restrict__ hp,
                                                                   const float *__restrict__ go_mask, const float *__restrict__ go_off,
                                                                   const float *__restrict__ fin, float *__restrict__ dg,
                                                                   float *__restrict__ partial) {
    __shared__ PcrHeadW<C> hw;
    pcr_load_head<C>(hp, hw);
    float acc[4 * C + 4];
#pragma unroll
    for (int k = 0; k < 4 * C + 4; ++k) acc[k] = 0.f;
    const int64_t cells = (int64_t)geo.d * geo.h * geo.w;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const float sm = go_mask[0] / fin[4], beta = fin[2], so = go_off[0] / fin[3];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += stride) {
        const int4 c = reinterpret_cast<const int4 *>(coors)[i];
        if ((unsigned)c.x >= (unsigned)geo.batch || (unsigned)c.y >= (unsigned)geo.d || (unsigned)c.z >= (unsigned)geo.h ||
            (unsigned)c.w >= (unsigned)geo.w)
            continue;
        const float *f = feats + i * 5;
        const float s = (((f[0] + f[1]) + f[2]) + f[3]) + f[4];
        const bool pos = s != 0.f;
        const int64_t cell = ((int64_t)c.y * geo.h + c.z) * geo.w + c.w;
        float gv[C], x, off[3];
        pcr_site_eval<C>(g, hw, cells, c.x, cell, gv, x, off);
        // occupied cell: d/dx beta*softplus(-x) = -beta*(1 - sigmoid(x)) replaces the dense pass's sigmoid(x)
        float dmk = 0.f;
        if (pos) {
            const float sg = sigmoidf(x);
            dmk = -sm * (beta * (1.f - sg) + sg);
        }
        float gc[3], dk[3];
        pcr_grid(geo, c.y, c.z, c.w, gc);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float t = pos ? f[k] - gc[k] : f[k];
            const float dlt = off[k] - t;
            dk[k] = (t != 0.f) ? (dlt > 0.f ? so : (dlt < 0.f ? -so : 0.f)) : 0.f;
        }
        float *ob = dg + (int64_t)c.x * C * cells + cell;
#pragma unroll
        for (int q = 0; q < C; ++q) {
            const float add = (hw.wm[q] * dmk + hw.wo[0][q] * dk[0]) + (hw.wo[1][q] * dk[1] + hw.wo[2][q] * dk[2]);
            ob[(int64_t)q * cells] += add;
            acc[q] = fmaf(dmk, gv[q], acc[q]);
#pragma unroll
            for (int k = 0; k < 3; ++k) acc[C + k * C + q] = fmaf(dk[k], gv[q], acc[C + k * C + q]);
        }
        acc[4 * C] += dmk;
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[4 * C + 1 + k] += dk[k];
    }
    block_sums_n<4 * C + 4>(acc, partial);
}

// one block per parameter-gradient element t (4C+4 of them): 256 threads stride over the partial rows, fixed-order fold
template <int C>
__global__ __launch_bounds__(256) void pcr_heads_param_grads_kernel(const float *__restrict__ dense_partial, int nd,
                                                                    const float *__restrict__ sparse_partial, int ns, float *__restrict__ dw_mask,
                                                                    float *__restrict__ db_mask, float *__restrict__ dw_off,
                                                                    float *__restrict__ db_off) {
    const int t = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < ns; i += 256) s += sparse_partial[(int64_t)i * (4 * C + 4) + t];
    if (t < C || t == 4 * C) {
        const int col = t < C ? t : C;
        for (int i = threadIdx.x; i < nd; i += 256) s += dense_partial[(int64_t)i * (C + 1) + col];
    }
    float v[1] = {s};
    __shared__ float tot[1];
    // block_sums writes out[blockIdx.x * K + k]: fold into a one-element LDS slot instead
    {
        __shared__ float red[4];
        float x = v[0];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
        __syncthreads();
        if (threadIdx.x == 0) tot[0] = (red[0] + red[1]) + (red[2] + red[3]);
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const float r = tot[0];
    if (t < C) dw_mask[t] = r;
    else if (t < 4 * C) dw_off[t - C] = r;
    else if (t == 4 * C) db_mask[0] = r;
    else db_off[t - 4 * C - 1] = r;
}

constexpr int PCRH_DENSE_BLOCKS = 2048, PCRH_SPARSE_BLOCKS = 256;

template <int C, int V>
static int pcr_heads_fwd_t(const float *g, const float *hw, const int32_t *coors, const float *feats, int64_t m, PcrGeo geo, float *out8,
                           float *ws, hipStream_t st) {
    const int64_t cells = (int64_t)geo.d * geo.h * geo.w, n = cells * geo.batch;
    float *dense_partial = ws, *sparse_partial = ws + PCRH_DENSE_BLOCKS;
    const int nd = (int)std::min<int64_t>(PCRH_DENSE_BLOCKS, std::max<int64_t>(1, ceil_div(n / V, 256)));
    const int ns = (int)std::min<int64_t>(PCRH_SPARSE_BLOCKS, std::max<int64_t>(1, ceil_div(m, 256)));
    hipLaunchKernelGGL((pcr_heads_fwd_dense_kernel<C, V>), dim3(nd), dim3(256), 0, st, g, hw, cells, geo.batch, dense_partial);
    hipLaunchKernelGGL((pcr_heads_fwd_sparse_kernel<C>), dim3(ns), dim3(256), 0, st, coors, feats, m, geo, g, hw, sparse_partial);
    hipLaunchKernelGGL(pcr_finalize_kernel, dim3(1), dim3(64), 0, st, dense_partial, nd, sparse_partial, ns, (double)n, out8);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

template <int C, int CO, int V>
static int pcr_heads_bwd_t(const float *g, const float *hw, const int32_t *coors, const float *feats, int64_t m, PcrGeo geo, const float *fin,
                           const float *go_mask, const float *go_off, const float *dz, const float *w2, float *dg, float *dw_mask, float *db_mask,
                           float *dw_off, float *db_off, float *ws, hipStream_t st) {
    const int64_t cells = (int64_t)geo.d * geo.h * geo.w, n = cells * geo.batch;
    float *dense_partial = ws, *sparse_partial = ws + (size_t)PCRH_DENSE_BLOCKS * (C + 1);
    const int nd = (int)std::min<int64_t>(PCRH_DENSE_BLOCKS, std::max<int64_t>(1, ceil_div(n / V, 256)));
    const int ns = (int)std::min<int64_t>(PCRH_SPARSE_BLOCKS, std::max<int64_t>(1, ceil_div(m, 256)));
    hipLaunchKernelGGL((pcr_heads_bwd_dense_kernel<C, CO, V>), dim3(nd), dim3(256), 0, st, g, dz, w2, hw, go_mask, fin, cells, geo.batch, dg,
                       dense_partial);
    hipLaunchKernelGGL((pcr_heads_bwd_sparse_kernel<C>), dim3(ns), dim3(256), 0, st, coors, feats, m, geo, g, hw, go_mask, go_off, fin, dg,
                       sparse_partial);
    hipLaunchKernelGGL((pcr_heads_param_grads_kernel<C>), dim3(4 * C + 4), dim3(256), 0, st, dense_partial, nd, sparse_partial, ns, dw_mask, db_mask, dw_off,
                       db_off);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

}  // namespace s2d

extern "C" int s2d_pcr_heads_supported(int c, int co, int64_t cells) {
    // one batch item's planes are addressed through a 32-bit buffer offset
    return ((c == 32 && (co == 0 || co == 16)) || (c == 3 && co == 0)) && cells > 0 && cells % 4 == 0 && (int64_t)c * cells * 4 < ((int64_t)1 << 31);
}

extern "C" size_t s2d_pcr_heads_workspace_bytes(int c) {
    return ((size_t)PCRH_DENSE_BLOCKS * (c + 1) + (size_t)PCRH_SPARSE_BLOCKS * (4 * c + 4)) * sizeof(float) + 512;
}

// head_params (device, 4C+4 floats): w_mask[C] | w_off[3][C] | b_mask | b_off[3]  (the two Conv3d's weights and biases)
extern "C" int s2d_pcr_heads_fwd_f32(const float *g, const float *head_params, const int32_t *coors, const float *feats, int64_t m, int batch,
                                     int c, int d, int h, int w, float *out8, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(g && head_params && out8 && batch > 0 && d > 0 && h > 0 && w > 0 && m >= 0 && (m == 0 || (coors && feats)),
                  "pcr_heads_fwd: bad argument");
    if (!s2d_pcr_heads_supported(c, 0, (int64_t)d * h * w)) {
        set_error("pcr_heads_fwd: unsupported channels %d / cells %lld", c, (long long)d * h * w);
        return S2D_ERR_UNSUPPORTED;
    }
    if (!ws || ws_bytes < s2d_pcr_heads_workspace_bytes(c)) {
        set_error("pcr_heads_fwd: workspace too small");
        return S2D_ERR_WORKSPACE;
    }
    PcrGeo geo{batch, d, h, w};
    if (c == 32) return pcr_heads_fwd_t<32, 2>(g, head_params, coors, feats, m, geo, out8, (float *)ws, (hipStream_t)stream);
    return pcr_heads_fwd_t<3, 4>(g, head_params, coors, feats, m, geo, out8, (float *)ws, (hipStream_t)stream);
}

extern "C" int s2d_pcr_heads_bwd_f32(const float *g, const float *head_params, const int32_t *coors, const float *feats, int64_t m, int batch,
                                     int c, int d, int h, int w, const float *fwd_out8, const float *go_mask, const float *go_offset,
                                     const float *dz, const float *w2, int co, float *dg, float *dw_mask, float *db_mask, float *dw_off,
                                     float *db_off, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(g && head_params && fwd_out8 && go_mask && go_offset && dg && dw_mask && db_mask && dw_off && db_off && batch > 0 && d > 0 &&
                      h > 0 && w > 0 && m >= 0 && (m == 0 || (coors && feats)) && (co == 0 || (dz && w2)),
                  "pcr_heads_bwd: bad argument");
    if (!s2d_pcr_heads_supported(c, co, (int64_t)d * h * w)) {
        set_error("pcr_heads_bwd: unsupported channels %d -> %d / cells %lld", c, co, (long long)d * h * w);
        return S2D_ERR_UNSUPPORTED;
    }
    if (!ws || ws_bytes < s2d_pcr_heads_workspace_bytes(c)) {
        set_error("pcr_heads_bwd: workspace too small");
        return S2D_ERR_WORKSPACE;
    }
    PcrGeo geo{batch, d, h, w};
    hipStream_t st = (hipStream_t)stream;
    float *wsf = (float *)ws;
    if (c == 32 && co == 16)
        return pcr_heads_bwd_t<32, 16, 2>(g, head_params, coors, feats, m, geo, fwd_out8, go_mask, go_offset, dz, w2, dg, dw_mask, db_mask, dw_off,
                                          db_off, wsf, st);
    if (c == 32)
        return pcr_heads_bwd_t<32, 0, 2>(g, head_params, coors, feats, m, geo, fwd_out8, go_mask, go_offset, dz, w2, dg, dw_mask, db_mask, dw_off,
                                         db_off, wsf, st);
    return pcr_heads_bwd_t<3, 0, 4>(g, head_params, coors, feats, m, geo, fwd_out8, go_mask, go_offset, dz, w2, dg, dw_mask, db_mask, dw_off, db_off,
                                    wsf, st);
}

// =====================================================================================================================
// PCR level with the preceding BatchNorm3d + ReLU folded in (rpn.py:265-272,287-291: ConvTranspose3d -> BatchNorm3d -> ReLU ->
// {gen_mask_k, gen_out_k, next 1x1x1 conv}).  The kernels read the RAW ConvTranspose3d output y and apply g = relu(y*scale+shift)
// on the fly; the post-norm volume g, its gradient and the masked gradient of the batch norm are never written:
//   fwd   one dense pass: logits -> softplus sums, z = w2.g + b2 written (the level's next conv);  sparse terms at the recon voxels
//   bwd A one dense pass, no writes: dG = w_mask*dL/dlogit + w2^T.dz; per-channel batch-norm sums (sum dG*m, sum dG*m*y) and the
//         dense part of dw_mask / db_mask;  sparse pass: corrections of both + the offset-head gradients
//   (batch-norm finalisation on the host side: (a, b, d) per channel)
//   bwd B one dense pass: dy = a*dG*m + b*y + d;  sparse pass: dy[cell] += a * correction * m
// against conv -> BN apply -> heads -> BN reduce -> BN apply this removes 5 of 9 passes over the level's 0.5-0.7 GB volumes.
// =====================================================================================================================
namespace s2d {

template <int C>
struct PcrNorm {
    float sc[C];
    float sh[C];
};
template <int C>
__device__ __forceinline__ void pcr_load_norm(const float *__restrict__ bnp, PcrNorm<C> &nm) {
    float *dst = reinterpret_cast<float *>(&nm);
    for (int i = threadIdx.x; i < 2 * C; i += 256) dst[i] = bnp[i];
    __syncthreads();
}

template <int C, int CO, int V, typename TY = float, typename TZ = float>
__global__ __launch_bounds__(256) void pcr_level_fwd_dense_kernel(const TY *__restrict__ y, const float *__restrict__ bnp, const float *__restrict__ hp,
                                                                  const float *__restrict__ w2, const float *__restrict__ b2, int64_t cells, int batch,
                                                                  TZ *__restrict__ z, float *__restrict__ partial,
                                                                  float *__restrict__ zstat_partial) {
    // TZ (r06): z stored in fp32 or bf16 (round to nearest even; its statistics are then those of the STORED values - what the batch norm behind it reads)
    __shared__ __attribute__((aligned(16))) float w2s[CO > 0 ? CO * C : 4];   // [C][CO]; read as float4 rows (see pcr_level_bwd_dense_kernel)
    __shared__ float b2s[CO > 0 ? CO : 1];
    __shared__ PcrHeadW<C> hw;
    __shared__ PcrNorm<C> nm;
    pcr_load_head<C>(hp, hw);
    pcr_load_norm<C>(bnp, nm);
    if (CO > 0) {
        for (int i = threadIdx.x; i < CO * C; i += 256) w2s[(i % C) * CO + i / C] = w2[i];
        for (int i = threadIdx.x; i < CO; i += 256) b2s[i] = b2 ? b2[i] : 0.f;
        __syncthreads();
    }
    int lane_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    const float *w2v = w2s + lane_zero;
    // per-channel constants (BN scale, BN shift, mask-head weight) as ONE 16-byte LDS broadcast per channel instead of three dword reads
    __shared__ __attribute__((aligned(16))) float4 cst[C];
    for (int i = threadIdx.x; i < C; i += 256) cst[i] = float4{nm.sc[i], nm.sh[i], hw.wm[i], 0.f};
    __syncthreads();
    const float4 *cstv = cst + lane_zero;
    float acc[1] = {0.f};
    // zstat_partial (block-uniform, CO > 0): per-channel (sum, sum of squares) of the z this kernel writes = the statistics pass of the
    // BatchNorm3d that follows the 1x1x1 conv (generator_2[1], rpn.py:263-296) - one 362 MB read less per step
    float zs[CO > 0 ? 2 * CO : 1];
#pragma unroll
    for (int q = 0; q < (CO > 0 ? 2 * CO : 1); ++q) zs[q] = 0.f;
    const uint32_t sv = (uint32_t)(cells / V), stride = gridDim.x * 256u;
    const unsigned plane = (unsigned)cells * (unsigned)sizeof(TZ), yplane = (unsigned)cells * (unsigned)sizeof(TY);
    for (int b = 0; b < batch; ++b) {
        const __amdgpu_buffer_rsrc_t yr = planes_rsrc(y + (int64_t)b * C * cells, C * yplane);
        const __amdgpu_buffer_rsrc_t zr = CO > 0 ? planes_rsrc(z + (int64_t)b * CO * cells, CO * plane) : yr;
        for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < sv; j += stride) {
            asm volatile("" ::: "memory");
            const unsigned voff = j * (V * (unsigned)sizeof(TZ)), yoff = j * (V * (unsigned)sizeof(TY));
            YPack<C, V, TY> yv;
#pragma unroll
            for (int c = 0; c < C; ++c) yv.load(yr, yoff, c * yplane, c);
            float x[V], za[CO > 0 ? CO : 1][V];
#pragma unroll
            for (int k = 0; k < V; ++k) x[k] = hw.bm;
            if (CO > 0) {
#pragma unroll
                for (int q = 0; q < CO; ++q)
#pragma unroll
                    for (int k = 0; k < V; ++k) za[q][k] = b2s[q];
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float4 cc = cstv[c];
                const float sc = cc.x, sh = cc.y, wc = cc.z;
                float g[V];
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    g[k] = fmaxf(fmaf(yv.get(c, k), sc, sh), 0.f);
                    x[k] = fmaf(wc, g[k], x[k]);
                }
                if constexpr (CO > 0 && CO % 4 == 0) {
#pragma unroll
                    for (int q4 = 0; q4 < CO / 4; ++q4) {
                        const float4 wq = *reinterpret_cast<const float4 *>(w2v + c * CO + 4 * q4);   // one 16-byte LDS broadcast per four weights
#pragma unroll
                        for (int k = 0; k < V; ++k) {
                            za[4 * q4][k] = fmaf(wq.x, g[k], za[4 * q4][k]);
                            za[4 * q4 + 1][k] = fmaf(wq.y, g[k], za[4 * q4 + 1][k]);
                            za[4 * q4 + 2][k] = fmaf(wq.z, g[k], za[4 * q4 + 2][k]);
                            za[4 * q4 + 3][k] = fmaf(wq.w, g[k], za[4 * q4 + 3][k]);
                        }
                    }
                } else if (CO > 0) {
#pragma unroll
                    for (int q = 0; q < CO; ++q) {
                        const float wq = w2v[c * CO + q];
#pragma unroll
                        for (int k = 0; k < V; ++k) za[q][k] = fmaf(wq, g[k], za[q][k]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < V; ++k) acc[0] += softplusf(x[k]);
            if (CO > 0) {
#pragma unroll
                for (int q = 0; q < CO; ++q) {
                    if constexpr (sizeof(TZ) == 2) {
#pragma unroll
                        for (int k = 0; k < V; ++k) za[q][k] = (float)(__bf16)za[q][k];
                    }
                    buf_store_t<V, TZ>(zr, voff, q * plane, za[q]);
                }
                if (zstat_partial) {
#pragma unroll
                    for (int q = 0; q < CO; ++q)
#pragma unroll
                        for (int k = 0; k < V; ++k) {
                            zs[q] += za[q][k];
                            zs[CO + q] = fmaf(za[q][k], za[q][k], zs[CO + q]);
                        }
                }
            }
        }
    }
    block_sums<1>(acc, partial);
    if (CO > 0 && zstat_partial) {
        __syncthreads();   // block_sums' LDS rows are reused
        block_sums<(CO > 0 ? 2 * CO : 1)>(zs, zstat_partial);
    }
}

// zstats[k] = sum over the nd blocks of zstat_partial[block][k] (fixed order)
__global__ __launch_bounds__(64) void pcr_zstats_fold_kernel(const float *__restrict__ partial, int nd, int k, float *__restrict__ out) {
    const int col = blockIdx.x;
    double v = 0;
    for (int i = threadIdx.x; i < nd; i += 64) v += partial[(int64_t)i * k + col];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (threadIdx.x == 0) out[col] = (float)v;
}

// r06 site cache: the forward's per-voxel pass keeps the C raw values of every recon voxel's cell as one row [site][ROW] (ROW = C, or 4 for C = 3:
// 64-byte / 8-byte rows in bf16); the two backward passes read the row - one coalesced access - instead of C scattered 2-byte loads from C planes
// (C lines per site, and the sums pass evaluates every site once per channel group: 139 us for 1.2e5 voxels at C = 32 were line fetches).
template <int C> struct PcrRow { static constexpr int N = C == 3 ? 4 : C; };
template <int C, typename TY> __device__ __forceinline__ void pcr_row_load(const TY *__restrict__ row, float (&yv)[C]) {
    constexpr int N = PcrRow<C>::N;
    if constexpr (sizeof(TY) == 2) {
        uint32_t w[N / 2];
        if constexpr (N % 8 == 0) {
#pragma unroll
            for (int q = 0; q < N / 8; ++q) {
                const uint4 v = reinterpret_cast<const uint4 *>(row)[q];
                w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
            }
        } else {
            const uint2 v = *reinterpret_cast<const uint2 *>(row);
            w[0] = v.x; w[1] = v.y;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) yv[c] = __builtin_bit_cast(float, (c & 1) ? (w[c >> 1] & 0xFFFF0000u) : (w[c >> 1] << 16));
    } else {
        float t[N];
#pragma unroll
        for (int q = 0; q < N / 4; ++q) {
            const float4 v = reinterpret_cast<const float4 *>(row)[q];
            t[4 * q] = v.x; t[4 * q + 1] = v.y; t[4 * q + 2] = v.z; t[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) yv[c] = t[c];
    }
}
template <int C, typename TY> __device__ __forceinline__ void pcr_row_store(TY *__restrict__ row, const float (&yv)[C]) {   // (yv holds exact TY values)
    constexpr int N = PcrRow<C>::N;
    if constexpr (sizeof(TY) == 2) {
        uint32_t w[N / 2];
#pragma unroll
        for (int k = 0; k < N / 2; ++k) {
            const uint32_t lo = 2 * k < C ? (__builtin_bit_cast(uint32_t, yv[2 * k < C ? 2 * k : 0]) >> 16) : 0u;
            const uint32_t hi = 2 * k + 1 < C ? (__builtin_bit_cast(uint32_t, yv[2 * k + 1 < C ? 2 * k + 1 : 0]) & 0xFFFF0000u) : 0u;
            w[k] = lo | hi;
        }
        if constexpr (N % 8 == 0) {
#pragma unroll
            for (int q = 0; q < N / 8; ++q) reinterpret_cast<uint4 *>(row)[q] = uint4{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]};
        } else {
            *reinterpret_cast<uint2 *>(row) = uint2{w[0], w[1]};
        }
    } else {
#pragma unroll
        for (int q = 0; q < N / 4; ++q)
            reinterpret_cast<float4 *>(row)[q] = float4{yv[4 * q < C ? 4 * q : 0], 4 * q + 1 < C ? yv[4 * q + 1 < C ? 4 * q + 1 : 0] : 0.f,
                                                        4 * q + 2 < C ? yv[4 * q + 2 < C ? 4 * q + 2 : 0] : 0.f,
                                                        4 * q + 3 < C ? yv[4 * q + 3 < C ? 4 * q + 3 : 0] : 0.f};
    }
}

// the site's raw values, post-norm values, logit and offsets; yrow (optional): the site's row of the site cache instead of the C planes
template <int C, typename TY = float>
__device__ __forceinline__ void pcr_site_eval_norm(const TY *__restrict__ y, const PcrHeadW<C> &hw, const PcrNorm<C> &nm, int64_t cells, int b,
                                                   int64_t cell, float (&yv)[C], float (&gv)[C], float &x, float (&off)[3],
                                                   const TY *__restrict__ yrow = nullptr) {
    x = hw.bm;
    off[0] = hw.bo[0]; off[1] = hw.bo[1]; off[2] = hw.bo[2];
    if (yrow) {   // block-uniform
        pcr_row_load<C, TY>(yrow, yv);
    } else {
        const TY *base = y + (int64_t)b * C * cells + cell;
#pragma unroll
        for (int c = 0; c < C; ++c) yv[c] = plane_elem<TY>(base + (int64_t)c * cells);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        gv[c] = fmaxf(fmaf(yv[c], nm.sc[c], nm.sh[c]), 0.f);
        x = fmaf(hw.wm[c], gv[c], x);
#pragma unroll
        for (int k = 0; k < 3; ++k) off[k] = fmaf(hw.wo[k][c], gv[c], off[k]);
    }
}

template <int C, typename TY = float>
__global__ __launch_bounds__(256) void pcr_level_fwd_sparse_kernel(const int32_t *__restrict__ coors, const float *__restrict__ feats, int64_t m,
                                                                   PcrGeo geo, const TY *__restrict__ y, const float *__restrict__ bnp,
                                                                   const float *__restrict__ hp, float *__restrict__ partial,
                                                                   TY *__restrict__ ysite = nullptr) {
    __shared__ PcrHeadW<C> hw;
    __shared__ PcrNorm<C> nm;
    pcr_load_head<C>(hp, hw);
    pcr_load_norm<C>(bnp, nm);
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const int64_t cells = (int64_t)geo.d * geo.h * geo.w;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += stride) {
        const int4 c = reinterpret_cast<const int4 *>(coors)[i];   // b,z,y,x
        if ((unsigned)c.x >= (unsigned)geo.batch || (unsigned)c.y >= (unsigned)geo.d || (unsigned)c.z >= (unsigned)geo.h ||
            (unsigned)c.w >= (unsigned)geo.w)
            continue;
        const float *f = feats + i * 5;
        const float s = (((f[0] + f[1]) + f[2]) + f[3]) + f[4];
        const bool pos = s != 0.f;
        const int64_t cell = ((int64_t)c.y * geo.h + c.z) * geo.w + c.w;
        float yv[C], gv[C], x, off[3];
        pcr_site_eval_norm<C, TY>(y, hw, nm, cells, c.x, cell, yv, gv, x, off);
        if (ysite) pcr_row_store<C, TY>(ysite + i * PcrRow<C>::N, yv);   // the site cache row the backward passes read (invalid sites: skipped there too)
        if (pos) {
            acc[0] += 1.f;
            acc[1] += softplusf(-x);
            acc[2] += softplusf(x);
        }
        float gc[3];
        pcr_grid(geo, c.y, c.z, c.w, gc);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float t = pos ? f[k] - gc[k] : f[k];
            if (t != 0.f) {
                acc[3] += fabsf(off[k] - t);
                acc[4] += 1.f;
            }
        }
    }
    block_sums<5>(acc, partial);
}

// pass A (APPLY = false): partial[block][3C+1] = dw_mask(C) | db_mask | sum dG*m (C) | sum dG*m*y (C), nothing written;
// pass B (APPLY = true):  dy = a*dG*m + b*y + d
template <int C, int CO, int V, bool APPLY, typename TY = float, typename TD = float, typename TZ = float>
__global__ __launch_bounds__(256) void pcr_level_bwd_dense_kernel(const TY *__restrict__ y, const TZ *__restrict__ dz, const float *__restrict__ bnp,
                                                                  const float *__restrict__ w2, const float *__restrict__ hp,
                                                                  const float *__restrict__ go_mask, const float *__restrict__ fin,
                                                                  const float *__restrict__ abd, int64_t cells, int batch, TD *__restrict__ dy,
                                                                  float *__restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float w2s[CO > 0 ? CO * C : 4];   // [C][CO]; read as float4 rows (r06: a quarter of the LDS instructions)
    __shared__ float abds[APPLY ? 3 * C : 1];
    __shared__ PcrHeadW<C> hw;
    __shared__ PcrNorm<C> nm;
    pcr_load_head<C>(hp, hw);
    pcr_load_norm<C>(bnp, nm);
    if (CO > 0)
        for (int i = threadIdx.x; i < CO * C; i += 256) w2s[(i % C) * CO + i / C] = w2[i];
    if (APPLY)
        for (int i = threadIdx.x; i < 3 * C; i += 256) abds[i] = abd[i];
    __syncthreads();
    const float scale = go_mask[0] / fin[4];
    int lane_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    const float *w2v = w2s + lane_zero, *abv = abds + lane_zero;
    __shared__ __attribute__((aligned(16))) float4 cst[C];   // (BN scale, BN shift, mask-head weight) per channel: one 16-byte LDS broadcast
    for (int i = threadIdx.x; i < C; i += 256) cst[i] = float4{nm.sc[i], nm.sh[i], hw.wm[i], 0.f};
    __syncthreads();
    const float4 *cstv = cst + lane_zero;
    float pw[APPLY ? 1 : 3 * C + 1];
#pragma unroll
    for (int c = 0; c < (APPLY ? 1 : 3 * C + 1); ++c) pw[c] = 0.f;
    const uint32_t sv = (uint32_t)(cells / V), stride = gridDim.x * 256u;
    const unsigned plane = (unsigned)cells * (unsigned)sizeof(TZ), yplane = (unsigned)cells * (unsigned)sizeof(TY), dplane = (unsigned)cells * (unsigned)sizeof(TD);
    for (int b = 0; b < batch; ++b) {
        const __amdgpu_buffer_rsrc_t yr = planes_rsrc(y + (int64_t)b * C * cells, C * yplane);
        const __amdgpu_buffer_rsrc_t dyr = APPLY ? planes_rsrc(dy + (int64_t)b * C * cells, C * dplane) : yr;
        const __amdgpu_buffer_rsrc_t zr = CO > 0 ? planes_rsrc(dz + (int64_t)b * CO * cells, CO * plane) : yr;
        for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < sv; j += stride) {
            asm volatile("" ::: "memory");
            const unsigned voff = j * (V * (unsigned)sizeof(TZ)), yoff = j * (V * (unsigned)sizeof(TY));   // (voff: dz, r06 fp32 or bf16)
            YPack<C, V, TY> yv;
#pragma unroll
            for (int c = 0; c < C; ++c) yv.load(yr, yoff, c * yplane, c);
            // dz as loaded (a bf16 dz stays PACKED until the second channel loop: unpacked next to the load, hipcc funnelled the 16 loads through one
            // register with an s_waitcnt vmcnt(0) behind each - 16 memory latencies in a row, 284 -> 416 us for the sums pass)
            YPack<(CO > 0 ? CO : 1), V, TZ> zp;
            if (CO > 0) {
#pragma unroll
                for (int q = 0; q < CO; ++q) zp.load(zr, voff, q * plane, q);
            }
            float x[V], dm[V];
#pragma unroll
            for (int k = 0; k < V; ++k) x[k] = hw.bm;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float4 cc = cstv[c];
                const float sc = cc.x, sh = cc.y, wc = cc.z;
#pragma unroll
                for (int k = 0; k < V; ++k) x[k] = fmaf(wc, fmaxf(fmaf(yv.get(c, k), sc, sh), 0.f), x[k]);
            }
#pragma unroll
            for (int k = 0; k < V; ++k) {
                dm[k] = scale * sigmoidf(x[k]);
                if constexpr (!APPLY) pw[C] += dm[k];
            }
            asm volatile("" ::: "memory");   // re-read the per-channel constants below instead of keeping 3C of them live across both loops
            float zv[CO > 0 ? CO : 1][V];
            if (CO > 0) {
#pragma unroll
                for (int q = 0; q < CO; ++q)
#pragma unroll
                    for (int k = 0; k < V; ++k) zv[q][k] = zp.get(q, k);
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float4 cc = cstv[c];
                const float sc = cc.x, sh = cc.y, wc = cc.z;
                float o[V];
#pragma unroll
                for (int k = 0; k < V; ++k) o[k] = wc * dm[k];
                if constexpr (CO > 0 && CO % 4 == 0) {
                    // the CO weights of channel c as CO / 4 16-byte LDS reads (wave-uniform address: a broadcast) - with one ds_read_b32 per
                    // weight the loop issued one LDS instruction per V fused multiply-adds and was bound by them (r06: built without packed
                    // FP32, rule 36, the kernel went from 299 to 410 us)
#pragma unroll
                    for (int q4 = 0; q4 < CO / 4; ++q4) {
                        const float4 wq = *reinterpret_cast<const float4 *>(w2v + c * CO + 4 * q4);
#pragma unroll
                        for (int k = 0; k < V; ++k) {
                            o[k] = fmaf(wq.x, zv[4 * q4][k], o[k]);
                            o[k] = fmaf(wq.y, zv[4 * q4 + 1][k], o[k]);
                            o[k] = fmaf(wq.z, zv[4 * q4 + 2][k], o[k]);
                            o[k] = fmaf(wq.w, zv[4 * q4 + 3][k], o[k]);
                        }
                    }
                } else if (CO > 0) {
#pragma unroll
                    for (int q = 0; q < CO; ++q) {
                        const float wq = w2v[c * CO + q];
#pragma unroll
                        for (int k = 0; k < V; ++k) o[k] = fmaf(wq, zv[q][k], o[k]);
                    }
                }
                if constexpr (APPLY) {
                    const float a = abv[c], bb = abv[C + c], dd = abv[2 * C + c];
                    float r[V];
#pragma unroll
                    for (int k = 0; k < V; ++k) {
                        const float g = fmaf(yv.get(c, k), sc, sh);
                        r[k] = fmaf(a, g > 0.f ? o[k] : 0.f, fmaf(bb, yv.get(c, k), dd));
                    }
                    buf_store_t<V, TD>(dyr, j * (V * (unsigned)sizeof(TD)), c * dplane, r);
                } else {
#pragma unroll
                    for (int k = 0; k < V; ++k) {
                        const float g = fmaxf(fmaf(yv.get(c, k), sc, sh), 0.f);
                        const float om = g > 0.f ? o[k] : 0.f;
                        pw[c] = fmaf(dm[k], g, pw[c]);
                        pw[C + 1 + c] += om;
                        pw[2 * C + 1 + c] = fmaf(om, yv.get(c, k), pw[2 * C + 1 + c]);
                    }
                }
            }
        }
    }
    if (!APPLY) block_sums_n<APPLY ? 1 : 3 * C + 1>(pw, partial);
}

// sparse pass A (APPLY = false): partial[block][6C+4] = dw_mask(C) | dw_off(3C) | db_mask | db_off(3) | sum corr*m (C) | sum corr*m*y (C);
//   blockIdx.y selects a group of CG channels whose accumulators the thread keeps (6*CG+4 registers instead of 6*C+4)
// sparse pass B (APPLY = true):  dy[c][cell] += a[c] * corr[c] * m
template <int C, int CG, bool APPLY, typename TY = float, typename TD = float>
__global__ __launch_bounds__(256) void pcr_level_bwd_sparse_kernel(const int32_t *__restrict__ coors, const float *__restrict__ feats, int64_t m,
                                                                   PcrGeo geo, const TY *__restrict__ y, const float *__restrict__ bnp,
                                                                   const float *__restrict__ hp, const float *__restrict__ go_mask,
                                                                   const float *__restrict__ go_off, const float *__restrict__ fin,
                                                                   const float *__restrict__ abd, TD *__restrict__ dy, float *__restrict__ partial,
                                                                   const TY *__restrict__ ysite = nullptr) {
    __shared__ PcrHeadW<C> hw;
    __shared__ PcrNorm<C> nm;
    pcr_load_head<C>(hp, hw);
    pcr_load_norm<C>(bnp, nm);
    constexpr int NACC = APPLY ? 1 : 6 * CG + 4;
    const int q0 = APPLY ? 0 : blockIdx.y * CG;   // first channel of this block's group
    float acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.f;
    const int64_t cells = (int64_t)geo.d * geo.h * geo.w;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const float sm = go_mask[0] / fin[4], beta = fin[2], so = go_off[0] / fin[3];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += stride) {
        const int4 c = reinterpret_cast<const int4 *>(coors)[i];
        if ((unsigned)c.x >= (unsigned)geo.batch || (unsigned)c.y >= (unsigned)geo.d || (unsigned)c.z >= (unsigned)geo.h ||
            (unsigned)c.w >= (unsigned)geo.w)
            continue;
        const float *f = feats + i * 5;
        const float s = (((f[0] + f[1]) + f[2]) + f[3]) + f[4];
        const bool pos = s != 0.f;
        const int64_t cell = ((int64_t)c.y * geo.h + c.z) * geo.w + c.w;
        float yv[C], gv[C], x, off[3];
        pcr_site_eval_norm<C, TY>(y, hw, nm, cells, c.x, cell, yv, gv, x, off, ysite ? ysite + i * PcrRow<C>::N : nullptr);
        float dmk = 0.f;
        if (pos) {
            const float sg = sigmoidf(x);
            dmk = -sm * (beta * (1.f - sg) + sg);
        }
        float gc[3], dk[3];
        pcr_grid(geo, c.y, c.z, c.w, gc);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float t = pos ? f[k] - gc[k] : f[k];
            const float dlt = off[k] - t;
            dk[k] = (t != 0.f) ? (dlt > 0.f ? so : (dlt < 0.f ? -so : 0.f)) : 0.f;
        }
        if constexpr (APPLY) {
            TD *ob = dy + (int64_t)c.x * C * cells + cell;   // (a cell belongs to one recon voxel: a 2-byte read-modify-write of a bf16 dy is private)
#pragma unroll
            for (int q = 0; q < C; ++q) {
                const float corr = (hw.wm[q] * dmk + hw.wo[0][q] * dk[0]) + (hw.wo[1][q] * dk[1] + hw.wo[2][q] * dk[2]);
                if (gv[q] > 0.f) ob[(int64_t)q * cells] = (TD)((float)ob[(int64_t)q * cells] + abd[q] * corr);
            }
        } else {
#pragma unroll
            for (int e = 0; e < CG; ++e) {
                // channel q0 + e: q0 is block-uniform; the unrolled selects below pick the group's values out of the C registers
                float g = 0.f, yy = 0.f, wmq = 0.f, wo0 = 0.f, wo1 = 0.f, wo2 = 0.f;
#pragma unroll
                for (int grp = 0; grp < C / CG; ++grp)
                    if (q0 == grp * CG) {
                        g = gv[grp * CG + e]; yy = yv[grp * CG + e];
                        wmq = hw.wm[grp * CG + e]; wo0 = hw.wo[0][grp * CG + e]; wo1 = hw.wo[1][grp * CG + e]; wo2 = hw.wo[2][grp * CG + e];
                    }
                const float corr = (wmq * dmk + wo0 * dk[0]) + (wo1 * dk[1] + wo2 * dk[2]);
                const float cm = g > 0.f ? corr : 0.f;
                acc[e] = fmaf(dmk, g, acc[e]);
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[CG + k * CG + e] = fmaf(dk[k], g, acc[CG + k * CG + e]);
                acc[4 * CG + 4 + e] += cm;
                acc[5 * CG + 4 + e] = fmaf(cm, yy, acc[5 * CG + 4 + e]);
            }
            acc[4 * CG] += dmk;
#pragma unroll
            for (int k = 0; k < 3; ++k) acc[4 * CG + 1 + k] += dk[k];
        }
    }
    if (!APPLY) {
        // scatter the group's sums into the [6C+4] row of this block: fold per value, then lane 0 of the block writes
        __shared__ float red[4][NACC];
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            float v = acc[k];
#pragma unroll
            for (int off2 = 32; off2 > 0; off2 >>= 1) v += __shfl_down(v, off2, 64);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
        }
        __syncthreads();
        float *row = partial + (int64_t)blockIdx.x * (6 * C + 4);
        for (int k = threadIdx.x; k < NACC; k += 256) {
            const float v = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
            int col;   // group layout [dwm CG | dwo 3CG | dbm | dbo 3 | S1 CG | S2 CG] -> row layout [dwm C | dwo 3C | dbm | dbo 3 | S1 C | S2 C]
            if (k < CG) col = q0 + k;
            else if (k < 4 * CG) col = C + ((k - CG) / CG) * C + q0 + (k - CG) % CG;
            else if (k < 4 * CG + 4) col = (blockIdx.y == 0) ? 4 * C + (k - 4 * CG) : -1;   // the scalars once
            else if (k < 5 * CG + 4) col = 4 * C + 4 + q0 + (k - 4 * CG - 4);
            else col = 5 * C + 4 + q0 + (k - 5 * CG - 4);
            if (col >= 0) row[col] = v;
        }
    }
}

// one block per output t: grads[4C+4] = dw_mask | dw_off | db_mask | db_off, then bn_sums[2C]
template <int C>
__global__ __launch_bounds__(256) void pcr_level_fold_kernel(const float *__restrict__ dense_partial, int nd, const float *__restrict__ sparse_partial,
                                                             int ns, float *__restrict__ grads, float *__restrict__ bn_sums) {
    const int t = blockIdx.x;   // 0 .. 6C+3 in the sparse layout
    float s = 0.f;
    for (int i = threadIdx.x; i < ns; i += 256) s += sparse_partial[(int64_t)i * (6 * C + 4) + t];
    int col = -1;   // matching column of the dense layout dw_mask(C) | db_mask | S1(C) | S2(C)
    if (t < C) col = t;
    else if (t == 4 * C) col = C;
    else if (t >= 4 * C + 4) col = C + 1 + (t - 4 * C - 4);
    if (col >= 0)
        for (int i = threadIdx.x; i < nd; i += 256) s += dense_partial[(int64_t)i * (3 * C + 1) + col];
    __shared__ float red[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const float r = (red[0] + red[1]) + (red[2] + red[3]);
    if (t < 4 * C + 4) grads[t] = r;
    else bn_sums[t - 4 * C - 4] = r;
}

// the site cache of the NEXT s2d_pcr_level_* call on this thread (s2d_pcr_level_site_cache): consumed (and cleared) by that call
static thread_local void *g_site_buf = nullptr;
static thread_local size_t g_site_bytes = 0;
struct SiteCacheScope {   // one per entry call: whatever was set is gone when the call returns
    ~SiteCacheScope() { g_site_buf = nullptr; g_site_bytes = 0; }
};
template <int C, typename TY> static TY *site_cache(int64_t m) {
    return (g_site_buf && m > 0 && g_site_bytes >= (size_t)m * PcrRow<C>::N * sizeof(TY)) ? (TY *)g_site_buf : nullptr;
}

template <int C, int CO, int V, typename TY = float, typename TZ = float>
static int pcr_level_fwd_t(const TY *y, const float *bnp, const float *hp, const float *w2, const float *b2, const int32_t *coors, const float *feats,
                           int64_t m, PcrGeo geo, TZ *z, float *out8, float *ws, hipStream_t st, float *z_stats = nullptr) {
    const int64_t cells = (int64_t)geo.d * geo.h * geo.w, n = cells * geo.batch;
    float *dense_partial = ws, *sparse_partial = ws + PCRH_DENSE_BLOCKS;
    float *zstat_partial = (CO > 0 && z_stats) ? ws + PCRH_DENSE_BLOCKS + (size_t)PCRH_SPARSE_BLOCKS * 5 : nullptr;   // [nd][2 CO] behind the sparse rows
    const int nd = (int)std::min<int64_t>(PCRH_DENSE_BLOCKS, std::max<int64_t>(1, ceil_div(cells / V, 256)));
    const int ns = (int)std::min<int64_t>(PCRH_SPARSE_BLOCKS, std::max<int64_t>(1, ceil_div(m, 256)));
    hipLaunchKernelGGL((pcr_level_fwd_dense_kernel<C, CO, V, TY, TZ>), dim3(nd), dim3(256), 0, st, y, bnp, hp, w2, b2, cells, geo.batch, z, dense_partial,
                       zstat_partial);
    if (zstat_partial) hipLaunchKernelGGL(pcr_zstats_fold_kernel, dim3(2 * CO), dim3(64), 0, st, (const float *)zstat_partial, nd, 2 * CO, z_stats);
    hipLaunchKernelGGL((pcr_level_fwd_sparse_kernel<C, TY>), dim3(ns), dim3(256), 0, st, coors, feats, m, geo, y, bnp, hp, sparse_partial,
                       site_cache<C, TY>(m));
    hipLaunchKernelGGL(pcr_finalize_kernel, dim3(1), dim3(64), 0, st, dense_partial, nd, sparse_partial, ns, (double)n, out8);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

template <int C, int CO, int V, typename TY = float, typename TZ = float>
static int pcr_level_bwd_sums_t(const TY *y, const float *bnp, const float *hp, const int32_t *coors, const float *feats, int64_t m, PcrGeo geo,
                                const float *fin, const float *go_mask, const float *go_off, const TZ *dz, const float *w2, float *grads,
                                float *bn_sums, float *ws, hipStream_t st) {
    const int64_t cells = (int64_t)geo.d * geo.h * geo.w;
    float *dense_partial = ws, *sparse_partial = ws + (size_t)PCRH_DENSE_BLOCKS * (3 * C + 1);
    const int nd = (int)std::min<int64_t>(PCRH_DENSE_BLOCKS, std::max<int64_t>(1, ceil_div(cells / V, 256)));
    const int ns = (int)std::min<int64_t>(PCRH_SPARSE_BLOCKS, std::max<int64_t>(1, ceil_div(m, 256)));
    hipLaunchKernelGGL((pcr_level_bwd_dense_kernel<C, CO, V, false, TY, float, TZ>), dim3(nd), dim3(256), 0, st, y, dz, bnp, w2, hp, go_mask, fin, nullptr, cells,
                       geo.batch, (float *)nullptr, dense_partial);
    constexpr int CG = C == 32 ? 8 : C;   // channel groups of the sparse pass (accumulator registers)
    hipLaunchKernelGGL((pcr_level_bwd_sparse_kernel<C, CG, false, TY, float>), dim3(ns, C / CG), dim3(256), 0, st, coors, feats, m, geo, y, bnp, hp, go_mask,
                       go_off, fin, nullptr, (float *)nullptr, sparse_partial, (const TY *)site_cache<C, TY>(m));
    hipLaunchKernelGGL((pcr_level_fold_kernel<C>), dim3(6 * C + 4), dim3(256), 0, st, dense_partial, nd, sparse_partial, ns, grads, bn_sums);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

template <int C, int CO, int V, typename TY = float, typename TD = float, typename TZ = float>
static int pcr_level_bwd_apply_t(const TY *y, const float *bnp, const float *hp, const int32_t *coors, const float *feats, int64_t m, PcrGeo geo,
                                 const float *fin, const float *go_mask, const float *go_off, const TZ *dz, const float *w2, const float *abd,
                                 TD *dy, hipStream_t st) {
    const int64_t cells = (int64_t)geo.d * geo.h * geo.w;
    const int nd = (int)std::min<int64_t>(2 * PCRH_DENSE_BLOCKS, std::max<int64_t>(1, ceil_div(cells / V, 256)));
    hipLaunchKernelGGL((pcr_level_bwd_dense_kernel<C, CO, V, true, TY, TD, TZ>), dim3(nd), dim3(256), 0, st, y, dz, bnp, w2, hp, go_mask, fin, abd, cells, geo.batch,
                       dy, nullptr);
    if (m > 0)
        hipLaunchKernelGGL((pcr_level_bwd_sparse_kernel<C, C, true, TY, TD>), dim3((unsigned)std::min<int64_t>(4096, ceil_div(m, 256))), dim3(256), 0, st,
                           coors, feats, m, geo, y, bnp, hp, go_mask, go_off, fin, abd, dy, nullptr, (const TY *)site_cache<C, TY>(m));
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

}  // namespace s2d

extern "C" size_t s2d_pcr_level_workspace_bytes(int c) {
    // (the forward pass uses [dense blocks][1] + [sparse blocks][5] + [dense blocks][2 co <= 32]: inside the backward passes' size for c >= 3... made explicit)
    const size_t bwd = (size_t)PCRH_DENSE_BLOCKS * (3 * c + 1) + (size_t)PCRH_SPARSE_BLOCKS * (6 * c + 4);
    const size_t fwd = (size_t)PCRH_DENSE_BLOCKS * (1 + 32) + (size_t)PCRH_SPARSE_BLOCKS * 5;
    return std::max(bwd, fwd) * sizeof(float) + 512;
}

// y: RAW ConvTranspose3d output [B][C][cells]; bn_scale_shift (device, 2C) = scale[C] | shift[C] of the batch norm that follows it
// (g = relu(y*scale + shift)); head_params as s2d_pcr_heads_fwd_f32; z[B][co][cells] = w2.g + b2 (co > 0); out8 as s2d_pcr_loss_fwd_f32.
static int pcr_level_fwd_any(const void *yv, int y16, const float *bn_scale_shift, const float *head_params, const float *w2, const float *b2,
                                     const int32_t *coors, const float *feats, int64_t m, int batch, int c, int co, int d, int h, int w, float *z,
                                     float *z_stats /* [2 co] = sum | sum of squares of z per channel, or NULL */, float *out8, void *ws,
                                     size_t ws_bytes, s2d_stream_t stream, int z16 = 0) {
    SiteCacheScope site_scope;   // (s2d_pcr_level_site_cache applies to this call only)
    S2D_CHECK_ARG(yv && bn_scale_shift && head_params && out8 && batch > 0 && d > 0 && h > 0 && w > 0 && m >= 0 && (m == 0 || (coors && feats)) &&
                      (co == 0 || (w2 && z)),
                  "pcr_level_fwd: bad argument");
    if (!s2d_pcr_heads_supported(c, co, (int64_t)d * h * w)) {
        set_error("pcr_level_fwd: unsupported channels %d -> %d / cells %lld", c, co, (long long)d * h * w);
        return S2D_ERR_UNSUPPORTED;
    }
    if (!ws || ws_bytes < s2d_pcr_level_workspace_bytes(c)) {
        set_error("pcr_level_fwd: workspace too small");
        return S2D_ERR_WORKSPACE;
    }
    PcrGeo geo{batch, d, h, w};
    hipStream_t st = (hipStream_t)stream;
    float *wsf = (float *)ws;
    if (z16) {   // r06: z written in bf16 (the 32 -> 16 level behind a bf16-stored y)
        if (!(y16 && c == 32 && co == 16)) {
            set_error("pcr_level_fwd: bf16-stored z is built for the bf16-y 32 -> 16 level only");
            return S2D_ERR_UNSUPPORTED;
        }
        return pcr_level_fwd_t<32, 16, 2, __bf16, __bf16>((const __bf16 *)yv, bn_scale_shift, head_params, w2, b2, coors, feats, m, geo, (__bf16 *)z, out8,
                                                          wsf, st, z_stats);
    }
    if (y16) {
        const __bf16 *y = (const __bf16 *)yv;
        if (c == 32 && co == 16) return pcr_level_fwd_t<32, 16, 2, __bf16>(y, bn_scale_shift, head_params, w2, b2, coors, feats, m, geo, z, out8, wsf, st, z_stats);
        if (c == 32) return pcr_level_fwd_t<32, 0, 2, __bf16>(y, bn_scale_shift, head_params, w2, b2, coors, feats, m, geo, z, out8, wsf, st);
        return pcr_level_fwd_t<3, 0, 4, __bf16>(y, bn_scale_shift, head_params, w2, b2, coors, feats, m, geo, z, out8, wsf, st);
    }
    const float *y = (const float *)yv;
    if (c == 32 && co == 16) return pcr_level_fwd_t<32, 16, 2>(y, bn_scale_shift, head_params, w2, b2, coors, feats, m, geo, z, out8, wsf, st, z_stats);
    if (c == 32) return pcr_level_fwd_t<32, 0, 2>(y, bn_scale_shift, head_params, w2, b2, coors, feats, m, geo, z, out8, wsf, st);
    return pcr_level_fwd_t<3, 0, 4>(y, bn_scale_shift, head_params, w2, b2, coors, feats, m, geo, z, out8, wsf, st);
}
extern "C" int s2d_pcr_level_fwd_f32(const float *y, const float *bn_scale_shift, const float *head_params, const float *w2, const float *b2,
                                     const int32_t *coors, const float *feats, int64_t m, int batch, int c, int co, int d, int h, int w, float *z,
                                     float *z_stats, float *out8, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    return pcr_level_fwd_any(y, 0, bn_scale_shift, head_params, w2, b2, coors, feats, m, batch, c, co, d, h, w, z, z_stats, out8, ws, ws_bytes, stream);
}
/* r04: the same level with y stored as bf16 [B][C][cells] (the up-sampler's bf16 output, s2d_convt3d_mfma_fwd_stats_y16) */
extern "C" int s2d_pcr_level_fwd_y16(const void *y, const float *bn_scale_shift, const float *head_params, const float *w2, const float *b2,
                                     const int32_t *coors, const float *feats, int64_t m, int batch, int c, int co, int d, int h, int w, float *z,
                                     float *z_stats, float *out8, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    return pcr_level_fwd_any(y, 1, bn_scale_shift, head_params, w2, b2, coors, feats, m, batch, c, co, d, h, w, z, z_stats, out8, ws, ws_bytes, stream);
}

/* r06: ... and z written as bf16 [B][co][cells] too (c = 32, co = 16): read by s2d_convt3d_mfma_fwd_stats_y16_norm_x16 / _wgrad_d16_norm_x16 and the
 * typed s2d_bncm_bwd_* passes; z_stats are the sums of the stored (rounded) values */
extern "C" int s2d_pcr_level_fwd_y16_z16(const void *y, const float *bn_scale_shift, const float *head_params, const float *w2, const float *b2,
                                         const int32_t *coors, const float *feats, int64_t m, int batch, int c, int co, int d, int h, int w, void *z_bf16,
                                         float *z_stats, float *out8, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    return pcr_level_fwd_any(y, 1, bn_scale_shift, head_params, w2, b2, coors, feats, m, batch, c, co, d, h, w, (float *)z_bf16, z_stats, out8, ws, ws_bytes,
                             stream, 1);
}

/* r06: a per-voxel cache for the NEXT s2d_pcr_level_* call on this thread (any storage variant).  The forward call fills it - row i = the c raw
 * values of y at recon voxel i's cell, [m][c] elements of y's type ([m][4] for c = 3) - and the two backward calls of the same level read their
 * voxels' values from it instead of c scattered loads per voxel.  buf: device memory of at least m * (c == 3 ? 4 : c) * sizeof(y element) bytes
 * (smaller: ignored); the setting is consumed by the next call and must be repeated per call.  NULL / 0 clears. */
extern "C" int s2d_pcr_level_site_cache(void *buf, size_t bytes) {
    g_site_buf = buf;
    g_site_bytes = buf ? bytes : 0;
    return S2D_OK;
}

// pass A of the backward: grads[4C+4] = dw_mask[C] | dw_off[3][C] | db_mask | db_off[3] and bn_sums[2C] = (sum dG*m, sum dG*m*y) per
// channel, the batch norm's backward reduction (s2d_bncm_bwd_reduce_f32's output)
static int pcr_level_bwd_sums_any(const void *yv, int y16, const float *bn_scale_shift, const float *head_params, const int32_t *coors,
                                          const float *feats, int64_t m, int batch, int c, int d, int h, int w, const float *fwd_out8,
                                          const float *go_mask, const float *go_offset, const float *dz, const float *w2, int co, float *grads,
                                          float *bn_sums, void *ws, size_t ws_bytes, s2d_stream_t stream, int z16 = 0) {
    SiteCacheScope site_scope;   // (s2d_pcr_level_site_cache applies to this call only)
    S2D_CHECK_ARG(yv && bn_scale_shift && head_params && fwd_out8 && go_mask && go_offset && grads && bn_sums && batch > 0 && d > 0 && h > 0 && w > 0 &&
                      m >= 0 && (m == 0 || (coors && feats)) && (co == 0 || (dz && w2)),
                  "pcr_level_bwd_sums: bad argument");
    if (!s2d_pcr_heads_supported(c, co, (int64_t)d * h * w)) {
        set_error("pcr_level_bwd_sums: unsupported channels %d -> %d / cells %lld", c, co, (long long)d * h * w);
        return S2D_ERR_UNSUPPORTED;
    }
    if (!ws || ws_bytes < s2d_pcr_level_workspace_bytes(c)) {
        set_error("pcr_level_bwd_sums: workspace too small");
        return S2D_ERR_WORKSPACE;
    }
    PcrGeo geo{batch, d, h, w};
    hipStream_t st = (hipStream_t)stream;
    float *wsf = (float *)ws;
#define S2D_LVL_SUMS(TY_)                                                                                                                       \
    do {                                                                                                                                         \
        const TY_ *y = (const TY_ *)yv;                                                                                                          \
        if (c == 32 && co == 16)                                                                                                                 \
            return pcr_level_bwd_sums_t<32, 16, 2, TY_>(y, bn_scale_shift, head_params, coors, feats, m, geo, fwd_out8, go_mask, go_offset, dz, w2, \
                                                        grads, bn_sums, wsf, st);                                                                \
        if (c == 32)                                                                                                                             \
            return pcr_level_bwd_sums_t<32, 0, 2, TY_>(y, bn_scale_shift, head_params, coors, feats, m, geo, fwd_out8, go_mask, go_offset, dz, w2,  \
                                                       grads, bn_sums, wsf, st);                                                                 \
        return pcr_level_bwd_sums_t<3, 0, 4, TY_>(y, bn_scale_shift, head_params, coors, feats, m, geo, fwd_out8, go_mask, go_offset, dz, w2, grads, \
                                                  bn_sums, wsf, st);                                                                             \
    } while (0)
    if (z16) {   // r06: dz stored in bf16 (see s2d_pcr_level_fwd_y16_z16)
        if (!(y16 && c == 32 && co == 16)) {
            set_error("pcr_level_bwd_sums: bf16-stored dz is built for the bf16-y 32 -> 16 level only");
            return S2D_ERR_UNSUPPORTED;
        }
        return pcr_level_bwd_sums_t<32, 16, 2, __bf16, __bf16>((const __bf16 *)yv, bn_scale_shift, head_params, coors, feats, m, geo, fwd_out8, go_mask,
                                                               go_offset, (const __bf16 *)dz, w2, grads, bn_sums, wsf, st);
    }
    if (y16) S2D_LVL_SUMS(__bf16);
    S2D_LVL_SUMS(float);
#undef S2D_LVL_SUMS
}
extern "C" int s2d_pcr_level_bwd_sums_f32(const float *y, const float *bn_scale_shift, const float *head_params, const int32_t *coors,
                                          const float *feats, int64_t m, int batch, int c, int d, int h, int w, const float *fwd_out8,
                                          const float *go_mask, const float *go_offset, const float *dz, const float *w2, int co, float *grads,
                                          float *bn_sums, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    return pcr_level_bwd_sums_any(y, 0, bn_scale_shift, head_params, coors, feats, m, batch, c, d, h, w, fwd_out8, go_mask, go_offset, dz, w2, co, grads,
                                  bn_sums, ws, ws_bytes, stream);
}
extern "C" int s2d_pcr_level_bwd_sums_y16(const void *y, const float *bn_scale_shift, const float *head_params, const int32_t *coors,
                                          const float *feats, int64_t m, int batch, int c, int d, int h, int w, const float *fwd_out8,
                                          const float *go_mask, const float *go_offset, const float *dz, const float *w2, int co, float *grads,
                                          float *bn_sums, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    return pcr_level_bwd_sums_any(y, 1, bn_scale_shift, head_params, coors, feats, m, batch, c, d, h, w, fwd_out8, go_mask, go_offset, dz, w2, co, grads,
                                  bn_sums, ws, ws_bytes, stream);
}
extern "C" int s2d_pcr_level_bwd_sums_y16_z16(const void *y, const float *bn_scale_shift, const float *head_params, const int32_t *coors,
                                              const float *feats, int64_t m, int batch, int c, int d, int h, int w, const float *fwd_out8,
                                              const float *go_mask, const float *go_offset, const void *dz_bf16, const float *w2, int co, float *grads,
                                              float *bn_sums, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    return pcr_level_bwd_sums_any(y, 1, bn_scale_shift, head_params, coors, feats, m, batch, c, d, h, w, fwd_out8, go_mask, go_offset,
                                  (const float *)dz_bf16, w2, co, grads, bn_sums, ws, ws_bytes, stream, 1);
}

// pass B: dy[B][C][cells] = a*dG*m + b*y + d with abd (device, 3C) = a[C] | b[C] | d[C] from the batch-norm backward finalisation
static int pcr_level_bwd_apply_any(const void *yv, int y16 /* 0: fp32 y, fp32 dy; 1: bf16 y, fp32 dy; 2: bf16 y, bf16 dy */, const float *bn_scale_shift, const float *head_params, const int32_t *coors,
                                           const float *feats, int64_t m, int batch, int c, int d, int h, int w, const float *fwd_out8,
                                           const float *go_mask, const float *go_offset, const float *dz, const float *w2, int co, const float *abd,
                                           void *dyv, s2d_stream_t stream, int z16 = 0) {
    SiteCacheScope site_scope;   // (s2d_pcr_level_site_cache applies to this call only)
    S2D_CHECK_ARG(yv && bn_scale_shift && head_params && fwd_out8 && go_mask && go_offset && abd && dyv && batch > 0 && d > 0 && h > 0 && w > 0 &&
                      m >= 0 && (m == 0 || (coors && feats)) && (co == 0 || (dz && w2)),
                  "pcr_level_bwd_apply: bad argument");
    if (!s2d_pcr_heads_supported(c, co, (int64_t)d * h * w)) {
        set_error("pcr_level_bwd_apply: unsupported channels %d -> %d / cells %lld", c, co, (long long)d * h * w);
        return S2D_ERR_UNSUPPORTED;
    }
    PcrGeo geo{batch, d, h, w};
    hipStream_t st = (hipStream_t)stream;
#define S2D_LVL_APPLY(TY_, TD_)                                                                                                                  \
    do {                                                                                                                                          \
        const TY_ *y = (const TY_ *)yv;                                                                                                           \
        TD_ *dy = (TD_ *)dyv;                                                                                                                     \
        if (c == 32 && co == 16)                                                                                                                  \
            return pcr_level_bwd_apply_t<32, 16, 2, TY_, TD_>(y, bn_scale_shift, head_params, coors, feats, m, geo, fwd_out8, go_mask, go_offset, dz, w2, \
                                                         abd, dy, st);                                                                            \
        if (c == 32)                                                                                                                              \
            return pcr_level_bwd_apply_t<32, 0, 2, TY_, TD_>(y, bn_scale_shift, head_params, coors, feats, m, geo, fwd_out8, go_mask, go_offset, dz, w2,  \
                                                        abd, dy, st);                                                                             \
        return pcr_level_bwd_apply_t<3, 0, 4, TY_, TD_>(y, bn_scale_shift, head_params, coors, feats, m, geo, fwd_out8, go_mask, go_offset, dz, w2, abd, dy, \
                                                   st);                                                                                           \
    } while (0)
    if (z16) {   // r06: dz stored in bf16
        if (!(y16 == 2 && c == 32 && co == 16)) {
            set_error("pcr_level_bwd_apply: bf16-stored dz is built for the bf16-y / bf16-dy 32 -> 16 level only");
            return S2D_ERR_UNSUPPORTED;
        }
        return pcr_level_bwd_apply_t<32, 16, 2, __bf16, __bf16, __bf16>((const __bf16 *)yv, bn_scale_shift, head_params, coors, feats, m, geo, fwd_out8, go_mask,
                                                                        go_offset, (const __bf16 *)dz, w2, abd, (__bf16 *)dyv, st);
    }
    if (y16 == 2) S2D_LVL_APPLY(__bf16, __bf16);
    if (y16) S2D_LVL_APPLY(__bf16, float);
    S2D_LVL_APPLY(float, float);
#undef S2D_LVL_APPLY
}
extern "C" int s2d_pcr_level_bwd_apply_f32(const float *y, const float *bn_scale_shift, const float *head_params, const int32_t *coors,
                                           const float *feats, int64_t m, int batch, int c, int d, int h, int w, const float *fwd_out8,
                                           const float *go_mask, const float *go_offset, const float *dz, const float *w2, int co, const float *abd,
                                           float *dy, s2d_stream_t stream) {
    return pcr_level_bwd_apply_any(y, 0, bn_scale_shift, head_params, coors, feats, m, batch, c, d, h, w, fwd_out8, go_mask, go_offset, dz, w2, co, abd, dy,
                                   stream);
}
extern "C" int s2d_pcr_level_bwd_apply_y16(const void *y, const float *bn_scale_shift, const float *head_params, const int32_t *coors,
                                           const float *feats, int64_t m, int batch, int c, int d, int h, int w, const float *fwd_out8,
                                           const float *go_mask, const float *go_offset, const float *dz, const float *w2, int co, const float *abd,
                                           float *dy, s2d_stream_t stream) {
    return pcr_level_bwd_apply_any(y, 1, bn_scale_shift, head_params, coors, feats, m, batch, c, d, h, w, fwd_out8, go_mask, go_offset, dz, w2, co, abd, dy,
                                   stream);
}
/* bf16 y AND bf16 dy: the gradient's only readers (the up-sampler's data- and weight-gradient kernels) round it to bf16 anyway */
extern "C" int s2d_pcr_level_bwd_apply_y16_d16(const void *y, const float *bn_scale_shift, const float *head_params, const int32_t *coors,
                                               const float *feats, int64_t m, int batch, int c, int d, int h, int w, const float *fwd_out8,
                                               const float *go_mask, const float *go_offset, const float *dz, const float *w2, int co,
                                               const float *abd, void *dy_bf16, s2d_stream_t stream) {
    return pcr_level_bwd_apply_any(y, 2, bn_scale_shift, head_params, coors, feats, m, batch, c, d, h, w, fwd_out8, go_mask, go_offset, dz, w2, co, abd,
                                   dy_bf16, stream);
}
extern "C" int s2d_pcr_level_bwd_apply_y16_d16_z16(const void *y, const float *bn_scale_shift, const float *head_params, const int32_t *coors,
                                                   const float *feats, int64_t m, int batch, int c, int d, int h, int w, const float *fwd_out8,
                                                   const float *go_mask, const float *go_offset, const void *dz_bf16, const float *w2, int co,
                                                   const float *abd, void *dy_bf16, s2d_stream_t stream) {
    return pcr_level_bwd_apply_any(y, 2, bn_scale_shift, head_params, coors, feats, m, batch, c, d, h, w, fwd_out8, go_mask, go_offset,
                                   (const float *)dz_bf16, w2, co, abd, dy_bf16, stream, 1);
}

// =====================================================================================================================
// Feature-distillation loss of the S2D step (/root/reference/det3d/torchie/trainer/trainer.py:783-789):
//   w_pos * MSE(student[teacher > 0], teacher[teacher > 0]) + w_neg * MSE(student[teacher <= 0], teacher[teacher <= 0])
// over [B,256,188,188] feature maps (36 MB in bf16): ONE pass for the three sums (pos squared error, all squared error, pos count),
// one for the gradient.  The torch formulation (mask, difference, square, two masked sums, their backward) made ~10 passes.
// Elementwise over the flat memory of two dense tensors with IDENTICAL strides (the caller checks); bf16 or fp32 elements.
// =====================================================================================================================
namespace s2d {

template <typename T> struct Vec8;
template <> struct Vec8<float> {
    static __device__ __forceinline__ void load(const float *p, float (&v)[8]) {
        const float4 a = reinterpret_cast<const float4 *>(p)[0], b = reinterpret_cast<const float4 *>(p)[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store(float *p, const float (&v)[8]) {
        reinterpret_cast<float4 *>(p)[0] = float4{v[0], v[1], v[2], v[3]};
        reinterpret_cast<float4 *>(p)[1] = float4{v[4], v[5], v[6], v[7]};
    }
};
typedef __bf16 bf16x8m __attribute__((ext_vector_type(8)));
template <> struct Vec8<__bf16> {
    static __device__ __forceinline__ void load(const __bf16 *p, float (&v)[8]) {
        const bf16x8m a = *reinterpret_cast<const bf16x8m *>(p);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
    }
    static __device__ __forceinline__ void store(__bf16 *p, const float (&v)[8]) {
        bf16x8m a;
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = (__bf16)v[e];
        *reinterpret_cast<bf16x8m *>(p) = a;
    }
};

constexpr int MSE_BLOCKS = 1024;

// partial[block][3] = (sum_{t>0} (s-t)^2, sum (s-t)^2, count t>0)
template <typename TS, typename TT>
__global__ __launch_bounds__(256) void masked_mse_fwd_kernel(const TS *__restrict__ s, const TT *__restrict__ t, int64_t n8, float *__restrict__ partial) {
    float acc[3] = {0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float sv[8], tv[8];
        Vec8<TS>::load(s + i * 8, sv);
        Vec8<TT>::load(t + i * 8, tv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = sv[e] - tv[e], d2 = d * d;
            const bool pos = tv[e] > 0.f;
            acc[0] += pos ? d2 : 0.f;
            acc[1] += d2;
            acc[2] += pos ? 1.f : 0.f;
        }
    }
    block_sums<3>(acc, partial);
}

// out[0] = loss, out[1] = 2*w_pos/n_pos, out[2] = 2*w_neg/n_neg (the gradient scales), out[3] = n_pos
__global__ __launch_bounds__(64) void masked_mse_finalize_kernel(const float *__restrict__ partial, int nb, double n, float w_pos, float w_neg,
                                                                 float *__restrict__ out) {
    double v[3] = {0, 0, 0};
    for (int i = threadIdx.x; i < nb; i += 64)
        for (int k = 0; k < 3; ++k) v[k] += partial[i * 3 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
    if (threadIdx.x != 0) return;
    const double n_pos = v[2], n_neg = n - n_pos;
    out[0] = (float)((double)w_pos * v[0] / n_pos + (double)w_neg * (v[1] - v[0]) / n_neg);   // an empty class gives nan, as torch's mean of nothing
    out[1] = (float)(2.0 * w_pos / n_pos);
    out[2] = (float)(2.0 * w_neg / n_neg);
    out[3] = (float)n_pos;
}

template <typename TS, typename TT>
__global__ __launch_bounds__(256) void masked_mse_bwd_kernel(const TS *__restrict__ s, const TT *__restrict__ t, const float *__restrict__ fin,
                                                             const float *__restrict__ go, int64_t n8, TS *__restrict__ ds) {
    const float gp = go[0] * fin[1], gn = go[0] * fin[2];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float sv[8], tv[8], o[8];
        Vec8<TS>::load(s + i * 8, sv);
        Vec8<TT>::load(t + i * 8, tv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (sv[e] - tv[e]) * (tv[e] > 0.f ? gp : gn);
        Vec8<TS>::store(ds + i * 8, o);
    }
}

}  // namespace s2d

extern "C" size_t s2d_masked_mse_workspace_bytes(void) { return (size_t)MSE_BLOCKS * 3 * sizeof(float) + 256; }

// student / teacher: dense tensors of n elements (n % 8 == 0) in the SAME memory order; *_bf16 = 1: bf16 elements, 0: fp32.
// out4 (device): loss, gradient scales, positive count.
extern "C" int s2d_masked_mse_fwd(const void *student, int student_bf16, const void *teacher, int teacher_bf16, int64_t n, float w_pos,
                                  float w_neg, float *out4, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(student && teacher && out4 && n > 0, "masked_mse_fwd: bad argument");
    if (n % 8) {
        set_error("masked_mse_fwd: the element count must be a multiple of 8 (%lld)", (long long)n);
        return S2D_ERR_UNSUPPORTED;
    }
    if (!ws || ws_bytes < s2d_masked_mse_workspace_bytes()) {
        set_error("masked_mse_fwd: workspace too small");
        return S2D_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int64_t n8 = n / 8;
    const int nb = (int)std::min<int64_t>(MSE_BLOCKS, ceil_div(n8, 256));
    float *partial = (float *)ws;
#define S2D_MSE_FWD(TS, TT) \
    hipLaunchKernelGGL((masked_mse_fwd_kernel<TS, TT>), dim3(nb), dim3(256), 0, st, (const TS *)student, (const TT *)teacher, n8, partial)
    if (student_bf16 && teacher_bf16) S2D_MSE_FWD(__bf16, __bf16);
    else if (student_bf16) S2D_MSE_FWD(__bf16, float);
    else if (teacher_bf16) S2D_MSE_FWD(float, __bf16);
    else S2D_MSE_FWD(float, float);
#undef S2D_MSE_FWD
    hipLaunchKernelGGL(masked_mse_finalize_kernel, dim3(1), dim3(64), 0, st, partial, nb, (double)n, w_pos, w_neg, out4);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

// dstudent (same element type and order as student) = go * d loss / d student
extern "C" int s2d_masked_mse_bwd(const void *student, int student_bf16, const void *teacher, int teacher_bf16, int64_t n, const float *fwd_out4,
                                  const float *go, void *dstudent, s2d_stream_t stream) {
    S2D_CHECK_ARG(student && teacher && fwd_out4 && go && dstudent && n > 0 && n % 8 == 0, "masked_mse_bwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n8 = n / 8;
    const int nb = (int)std::min<int64_t>(4096, ceil_div(n8, 256));
#define S2D_MSE_BWD(TS, TT) \
    hipLaunchKernelGGL((masked_mse_bwd_kernel<TS, TT>), dim3(nb), dim3(256), 0, st, (const TS *)student, (const TT *)teacher, fwd_out4, go, n8, \
                       (TS *)dstudent)
    if (student_bf16 && teacher_bf16) S2D_MSE_BWD(__bf16, __bf16);
    else if (student_bf16) S2D_MSE_BWD(__bf16, float);
    else if (teacher_bf16) S2D_MSE_BWD(float, __bf16);
    else S2D_MSE_BWD(float, float);
#undef S2D_MSE_BWD
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
