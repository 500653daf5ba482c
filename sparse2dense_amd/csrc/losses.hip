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
__global__ void pcr_finalize_kernel(const float *__restrict__ dense_partial, int nd, const float *__restrict__ sparse_partial, int ns,
                                    double n_cells, float *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s_all = 0;
    for (int i = 0; i < nd; ++i) s_all += dense_partial[i];
    double t[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < ns; ++i)
        for (int k = 0; k < 5; ++k) t[k] += sparse_partial[i * 5 + k];
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
