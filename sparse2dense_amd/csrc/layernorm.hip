// LayerNorm over a whole [C,H,W] map per sample (the ConvNeXt blocks of the S2D module, /root/reference/det3d/models/necks/rpn.py:
// 210-247: nn.LayerNorm([256, 47, 47])): B = 2..4 rows of ~5.7e5 elements.  torch's LayerNorm gives each row ONE workgroup
// (B workgroups on a 256-CU part: 0.44 ms forward + 1.13 ms backward per layer); composed from torch reductions and elementwise
// ops on the NHWC bf16 maps it still cost ~0.55 ms per layer (strided Welford / sum kernels).  Here a row is split over
// LN_BLOCKS workgroups with a two-level, fixed-order reduction; everything is elementwise in the MEMORY order of x (the caller
// hands weight / bias permuted to that order, and gets their gradients in it):
//   fwd:  partial (sum, sumsq) -> y = (x - mean) * rstd * w + b, stats[b] = (mean, rstd)
//   bwd:  g = dy * w; partial (sum g, sum g*xhat) -> dx = rstd * (g - mean(g) - xhat * mean(g*xhat));
//         dw = sum_b dy * xhat, db = sum_b dy   (one thread owns an element of all B samples)
// x, y, dy, dx bf16; statistics, parameters and their gradients fp32.  HBM-bound: 4 passes over B * row * 2 bytes.
#include "s2d_common.h"

namespace s2d {

constexpr int LN_BLOCKS = 64;
typedef __bf16 bf16x8l __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void ln_block_sum2(float a, float b, float *out2) {
    __shared__ float red[2][4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off, 64);
        b += __shfl_down(b, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = a;
        red[1][threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out2[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        out2[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// grid (LN_BLOCKS, batch): partial[b][blk] = (sum x, sum x^2) over the block's slice of the row (8-element groups)
__global__ __launch_bounds__(256) void ln_stats_kernel(const __bf16 *__restrict__ x, int64_t row8, float *__restrict__ partial) {
    const bf16x8l *xr = reinterpret_cast<const bf16x8l *>(x) + (int64_t)blockIdx.y * row8;
    float s0 = 0.f, s1 = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < row8; i += (int64_t)LN_BLOCKS * 256) {
        const bf16x8l v = xr[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = (float)v[e];
            s0 += f;
            s1 += f * f;
        }
    }
    ln_block_sum2(s0, s1, partial + ((int64_t)blockIdx.y * LN_BLOCKS + blockIdx.x) * 2);
}

// every block folds the row's LN_BLOCKS partials itself (fixed order): (mean, rstd) forward, (mean g, mean g*xhat) backward
__device__ __forceinline__ void ln_fold(const float *__restrict__ partial, int b, double n, double &a, double &c) {
    a = 0;
    c = 0;
    for (int i = 0; i < LN_BLOCKS; ++i) {
        a += partial[((int64_t)b * LN_BLOCKS + i) * 2];
        c += partial[((int64_t)b * LN_BLOCKS + i) * 2 + 1];
    }
    a /= n;
    c /= n;
}

__global__ __launch_bounds__(256) void ln_apply_kernel(const __bf16 *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                       const float *__restrict__ partial, int64_t row8, float eps, __bf16 *__restrict__ y,
                                                       float *__restrict__ stats) {
    const int b = blockIdx.y;
    double m, q;
    ln_fold(partial, b, (double)row8 * 8.0, m, q);
    const float mean = (float)m, rstd = (float)(1.0 / sqrt(fmax(q - m * m, 0.0) + (double)eps));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        stats[2 * b] = mean;
        stats[2 * b + 1] = rstd;
    }
    const bf16x8l *xr = reinterpret_cast<const bf16x8l *>(x) + (int64_t)b * row8;
    bf16x8l *yr = reinterpret_cast<bf16x8l *>(y) + (int64_t)b * row8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < row8; i += (int64_t)LN_BLOCKS * 256) {
        const bf16x8l v = xr[i];
        const float4 w0 = reinterpret_cast<const float4 *>(w)[2 * i], w1 = reinterpret_cast<const float4 *>(w)[2 * i + 1];
        const float4 b0 = reinterpret_cast<const float4 *>(bias)[2 * i], b1 = reinterpret_cast<const float4 *>(bias)[2 * i + 1];
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        bf16x8l o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)fmaf(((float)v[e] - mean) * rstd, wv[e], bv[e]);
        yr[i] = o;
    }
}

__global__ __launch_bounds__(256) void ln_bwd_stats_kernel(const __bf16 *__restrict__ dy, const __bf16 *__restrict__ x, const float *__restrict__ w,
                                                           const float *__restrict__ stats, int64_t row8, float *__restrict__ partial) {
    const int b = blockIdx.y;
    const float mean = stats[2 * b], rstd = stats[2 * b + 1];
    const bf16x8l *xr = reinterpret_cast<const bf16x8l *>(x) + (int64_t)b * row8;
    const bf16x8l *gr = reinterpret_cast<const bf16x8l *>(dy) + (int64_t)b * row8;
    float s0 = 0.f, s1 = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < row8; i += (int64_t)LN_BLOCKS * 256) {
        const bf16x8l v = xr[i], d = gr[i];
        const float4 w0 = reinterpret_cast<const float4 *>(w)[2 * i], w1 = reinterpret_cast<const float4 *>(w)[2 * i + 1];
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float g = (float)d[e] * wv[e];
            s0 += g;
            s1 += g * (((float)v[e] - mean) * rstd);
        }
    }
    ln_block_sum2(s0, s1, partial + ((int64_t)b * LN_BLOCKS + blockIdx.x) * 2);
}

// grid (blocks over the row): a thread owns 8 elements of ALL samples: dx per sample, dw / db summed over the samples in order
__global__ __launch_bounds__(256) void ln_bwd_apply_kernel(const __bf16 *__restrict__ dy, const __bf16 *__restrict__ x, const float *__restrict__ w,
                                                           const float *__restrict__ stats, const float *__restrict__ partial, int batch,
                                                           int64_t row8, __bf16 *__restrict__ dx, float *__restrict__ dw, float *__restrict__ db) {
    extern __shared__ float fold[];   // [batch][4]: mean, rstd, mean g, mean g*xhat
    for (int b = threadIdx.x; b < batch; b += 256) {
        double a, c;
        ln_fold(partial, b, (double)row8 * 8.0, a, c);
        fold[4 * b] = stats[2 * b];
        fold[4 * b + 1] = stats[2 * b + 1];
        fold[4 * b + 2] = (float)a;
        fold[4 * b + 3] = (float)c;
    }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= row8) return;
    const float4 w0 = reinterpret_cast<const float4 *>(w)[2 * i], w1 = reinterpret_cast<const float4 *>(w)[2 * i + 1];
    const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    float aw[8], ab[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        aw[e] = 0.f;
        ab[e] = 0.f;
    }
    for (int b = 0; b < batch; ++b) {
        const float mean = fold[4 * b], rstd = fold[4 * b + 1], mg = fold[4 * b + 2], mgx = fold[4 * b + 3];
        const bf16x8l v = reinterpret_cast<const bf16x8l *>(x)[(int64_t)b * row8 + i];
        const bf16x8l d = reinterpret_cast<const bf16x8l *>(dy)[(int64_t)b * row8 + i];
        bf16x8l o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xh = ((float)v[e] - mean) * rstd, df = (float)d[e];
            aw[e] = fmaf(df, xh, aw[e]);
            ab[e] += df;
            o[e] = (__bf16)(rstd * (df * wv[e] - mg - xh * mgx));
        }
        if (dx) reinterpret_cast<bf16x8l *>(dx)[(int64_t)b * row8 + i] = o;
    }
    if (dw) {
        reinterpret_cast<float4 *>(dw)[2 * i] = float4{aw[0], aw[1], aw[2], aw[3]};
        reinterpret_cast<float4 *>(dw)[2 * i + 1] = float4{aw[4], aw[5], aw[6], aw[7]};
    }
    if (db) {
        reinterpret_cast<float4 *>(db)[2 * i] = float4{ab[0], ab[1], ab[2], ab[3]};
        reinterpret_cast<float4 *>(db)[2 * i + 1] = float4{ab[4], ab[5], ab[6], ab[7]};
    }
}

}  // namespace s2d

using namespace s2d;

extern "C" size_t s2d_lnwide_workspace_bytes(int batch) { return batch > 0 ? align_up((size_t)batch * LN_BLOCKS * 2 * sizeof(float), 256) : 0; }

/* x, y: bf16 [batch][row] in memory order; weight / bias fp32 [row] in THAT order (row % 8 == 0); stats [batch][2] = (mean, rstd) */
extern "C" int s2d_lnwide_fwd_bf16(const void *x, const float *weight, const float *bias, int batch, int64_t row, float eps, void *y,
                                   float *stats, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(x && weight && bias && y && stats && batch > 0 && batch <= 65535 && row > 0, "lnwide_fwd: bad argument");
    if (row % 8) {
        set_error("lnwide_fwd: the row length must be a multiple of 8 (%lld)", (long long)row);
        return S2D_ERR_UNSUPPORTED;
    }
    if (!ws || ws_bytes < s2d_lnwide_workspace_bytes(batch)) {
        set_error("lnwide_fwd: workspace too small");
        return S2D_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(LN_BLOCKS, batch), blk(256);
    hipLaunchKernelGGL(ln_stats_kernel, grid, blk, 0, st, (const __bf16 *)x, row / 8, (float *)ws);
    hipLaunchKernelGGL(ln_apply_kernel, grid, blk, 0, st, (const __bf16 *)x, weight, bias, (const float *)ws, row / 8, eps, (__bf16 *)y, stats);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

/* dx bf16 [batch][row] (optional), dweight / dbias fp32 [row] in memory order (optional) */
extern "C" int s2d_lnwide_bwd_bf16(const void *dy, const void *x, const float *weight, const float *stats, int batch, int64_t row, void *dx,
                                   float *dweight, float *dbias, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(dy && x && weight && stats && batch > 0 && batch <= 4096 && row > 0, "lnwide_bwd: bad argument");
    if (row % 8) {
        set_error("lnwide_bwd: the row length must be a multiple of 8 (%lld)", (long long)row);
        return S2D_ERR_UNSUPPORTED;
    }
    if (!ws || ws_bytes < s2d_lnwide_workspace_bytes(batch)) {
        set_error("lnwide_bwd: workspace too small");
        return S2D_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ln_bwd_stats_kernel, dim3(LN_BLOCKS, batch), dim3(256), 0, st, (const __bf16 *)dy, (const __bf16 *)x, weight, stats, row / 8,
                       (float *)ws);
    hipLaunchKernelGGL(ln_bwd_apply_kernel, dim3((unsigned)ceil_div(row / 8, 256)), dim3(256), (size_t)batch * 4 * sizeof(float), st,
                       (const __bf16 *)dy, (const __bf16 *)x, weight, stats, (const float *)ws, batch, row / 8, (__bf16 *)dx, dweight, dbias);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
