// Depth-wise 7x7 convolution (padding 3, stride 1) on NHWC bf16 maps: the first layer of the three ConvNeXt blocks of
// the S2D module (nn.Conv2d(256, 256, 7, padding=3, groups=256) on [B,256,47,47];
// /root/reference/det3d/models/necks/rpn.py:204-225).  MIOpen ran these through grouped-conv GEMM kernels
// (r02 baseline profile: 413 us forward, 653 us data gradient, 1389 us weight gradient PER LAYER for 0.2 GFLOP) - a
// depth-wise conv has no reduction over channels, so it is a streaming stencil, not a GEMM: HBM/L2 bound.
//
// Layout: x, y bf16 [N][H][W][C]; weight fp32 [C][49] (torch's [C,1,7,7]); bias fp32 [C]; fp32 accumulation.
//   forward / data gradient: a thread owns 8 channels (one 16-byte piece) of DW_XT horizontally adjacent output
//     pixels, so every loaded input piece feeds up to DW_XT outputs; the weights live in LDS as [tap][C] fp32 (the data
//     gradient is the same stencil with the taps mirrored).
//   weight gradient: grid (pixel chunk, kernel row); a thread accumulates 7 taps x 8 channels over its pixels, the
//     block folds its pixel lanes through LDS and writes one fp32 slab; a second kernel folds the slabs in a fixed
//     order (deterministic) into dW [C][49] (+ dbias).
#include "s2d_common.h"

namespace s2d {

typedef __bf16 bf16x8d __attribute__((ext_vector_type(8)));

constexpr int DW_XT = 4;      // output pixels (along x) per thread
constexpr int DW_K = 7, DW_TAPS = 49, DW_R = 3;

template <bool FLIP>
__global__ __launch_bounds__(256) void dwconv7_nhwc_bf16_kernel(const __bf16 *__restrict__ x, const float *__restrict__ w,
                                                                const float *__restrict__ bias, int n, int h, int wd, int c,
                                                                __bf16 *__restrict__ y) {
    // [49][c + 4]: the row stride is padded by four floats - the transposing store below has consecutive lanes on consecutive TAPS of one
    // channel, which with stride c (a multiple of 64 words) put all 49 of them on ONE bank (a 49-way conflict in every block's prologue: most
    // of the r02 kernel's 49 us); with c + 4 they spread over 16 banks (r06)
    extern __shared__ __attribute__((aligned(16))) float wl[];
    const int cp = c + 4;
    for (int e = threadIdx.x; e < DW_TAPS * c; e += 256) {
        const int ch = e / DW_TAPS, tap = e - ch * DW_TAPS;      // coalesced read of the torch layout
        wl[(FLIP ? DW_TAPS - 1 - tap : tap) * cp + ch] = w[e];
    }
    __syncthreads();
    const int groups = c / 8;
    const int xtiles = (wd + DW_XT - 1) / DW_XT;
    const int64_t total = (int64_t)n * h * xtiles * groups;
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int g = (int)(id % groups);
    int64_t r = id / groups;
    const int xt = (int)(r % xtiles); r /= xtiles;
    const int oy = (int)(r % h);
    const int b = (int)(r / h);
    const int x0 = xt * DW_XT, c0 = g * 8;

    float acc[DW_XT][8];
#pragma unroll
    for (int p = 0; p < DW_XT; ++p)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[p][e] = bias ? bias[c0 + e] : 0.f;

    // r06: branch-free.  The r02 form skipped rows / columns outside the map with `continue`: every 16-byte load then sat behind its own
    // branch, hipcc waited for each before issuing the next (70 dependent L2 round trips per thread: 49 us for a 4.5 MB map).  Now the ten
    // pieces of a kernel row are buffer loads whose offset is out of range where the pixel is outside the map (reads zero, no branch), all
    // issued before the first is used.
    const __amdgpu_buffer_rsrc_t xr = buf_rsrc(x, (unsigned)((int64_t)n * h * wd * c * 2));
#pragma unroll 1
    for (int ky = 0; ky < DW_K; ++ky) {
        const int iy = oy + ky - DW_R;
        const bool rowok = (unsigned)iy < (unsigned)h;
        const unsigned rowoff = (unsigned)(((((int64_t)b * h + (rowok ? iy : 0)) * wd) * c + c0) * 2);
        buf_f32x4 raw[DW_XT + DW_K - 1];
#pragma unroll
        for (int j = 0; j < DW_XT + DW_K - 1; ++j) {   // input columns x0-3 .. x0+DW_XT+2
            const int ix = x0 + j - DW_R;
            const bool ok = rowok && (unsigned)ix < (unsigned)wd;
            raw[j] = buf_load4(xr, ok ? rowoff + (unsigned)(ix * c * 2) : BUF_OOB, 0);
        }
        float wk[DW_K][8];
#pragma unroll
        for (int kx = 0; kx < DW_K; ++kx) {
            const float4 a = *reinterpret_cast<const float4 *>(&wl[(ky * DW_K + kx) * cp + c0]);
            const float4 bq = *reinterpret_cast<const float4 *>(&wl[(ky * DW_K + kx) * cp + c0 + 4]);
            wk[kx][0] = a.x; wk[kx][1] = a.y; wk[kx][2] = a.z; wk[kx][3] = a.w;
            wk[kx][4] = bq.x; wk[kx][5] = bq.y; wk[kx][6] = bq.z; wk[kx][7] = bq.w;
        }
#pragma unroll
        for (int j = 0; j < DW_XT + DW_K - 1; ++j) {
            const bf16x8d v = __builtin_bit_cast(bf16x8d, raw[j]);
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
#pragma unroll
            for (int p = 0; p < DW_XT; ++p) {
                const int kx = j - p;   // ix = (x0+p) + kx - 3
                if (kx >= 0 && kx < DW_K) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[p][e] = fmaf(f[e], wk[kx][e], acc[p][e]);
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < DW_XT; ++p) {
        if (x0 + p < wd) {
            bf16x8d o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)acc[p][e];
            *reinterpret_cast<bf16x8d *>(y + ((((int64_t)b * h + oy) * wd) + x0 + p) * c + c0) = o;
        }
    }
}

// ---- weight gradient ---------------------------------------------------------------------------
constexpr int DWG_PIX = 64;    // pixels per block
constexpr int DWG_LANES = 8;   // pixel lanes per channel group inside a block (256 threads = 32 groups x 8 lanes)

// slab layout: partial[chunk][50][c]: taps 0..48 of dW and row 49 = dbias (written by the ky == 3 blocks only)
__global__ __launch_bounds__(256) void dwconv7_wgrad_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ dy, int n, int h,
                                                            int wd, int c, int g0, float *__restrict__ partial) {
    __shared__ float red[DWG_LANES][8][32 * 8];   // [lane][kx or bias slot][group*8 + e]
    const int groups_here = 32;
    const int g = g0 + (threadIdx.x & 31);          // channel group of this thread
    const int lane = threadIdx.x >> 5;              // pixel lane
    const int ky = blockIdx.y;
    const int64_t pixels = (int64_t)n * h * wd;
    const int64_t p0 = (int64_t)blockIdx.x * DWG_PIX;
    const bool live = g * 8 < c;
    const int c0 = g * 8;
    float acc[DW_K + 1][8];
#pragma unroll
    for (int k = 0; k <= DW_K; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
    const __amdgpu_buffer_rsrc_t xr = buf_rsrc(x, (unsigned)(pixels * c * 2));
    if (live) {
        for (int64_t p = p0 + lane; p < min(p0 + DWG_PIX, pixels); p += DWG_LANES) {
            const int ox = (int)(p % wd);
            const int oy = (int)((p / wd) % h);
            const int b = (int)(p / ((int64_t)wd * h));
            const bf16x8d gq = *reinterpret_cast<const bf16x8d *>(dy + p * c + c0);
            float gf[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) gf[e] = (float)gq[e];
            if (ky == DW_R) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[DW_K][e] += gf[e];
            }
            // r06: branch-free (see dwconv7_nhwc_bf16_kernel): the seven pieces of the kernel row as buffer loads, out of range = zero
            const int iy = oy + ky - DW_R;
            const bool rowok = (unsigned)iy < (unsigned)h;
            const unsigned rowoff = (unsigned)(((((int64_t)b * h + (rowok ? iy : 0)) * wd) * c + c0) * 2);
            buf_f32x4 raw[DW_K];
#pragma unroll
            for (int kx = 0; kx < DW_K; ++kx) {
                const int ix = ox + kx - DW_R;
                raw[kx] = buf_load4(xr, (rowok && (unsigned)ix < (unsigned)wd) ? rowoff + (unsigned)(ix * c * 2) : BUF_OOB, 0);
            }
#pragma unroll
            for (int kx = 0; kx < DW_K; ++kx) {
                const bf16x8d v = __builtin_bit_cast(bf16x8d, raw[kx]);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[kx][e] = fmaf((float)v[e], gf[e], acc[kx][e]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k <= DW_K; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[lane][k][(threadIdx.x & 31) * 8 + e] = acc[k][e];
    __syncthreads();
    // fold the 8 pixel lanes in a fixed order; thread t < 8*256 elements / 256 threads = 8 elements each
    float *slab = partial + (int64_t)blockIdx.x * (DW_TAPS + 1) * c;
    for (int e = threadIdx.x; e < (DW_K + 1) * groups_here * 8; e += 256) {
        const int k = e / (groups_here * 8), col = e - k * (groups_here * 8);
        const int ch = g0 * 8 + col;
        if (ch >= c) continue;
        if (k == DW_K && ky != DW_R) continue;
        float s = 0.f;
#pragma unroll
        for (int l = 0; l < DWG_LANES; ++l) s += red[l][k][col];
        const int rowi = k == DW_K ? DW_TAPS : ky * DW_K + k;
        slab[(int64_t)rowi * c + ch] = s;
    }
}

// dW[ch][tap] = sum over chunks of partial[chunk][tap][ch]; db[ch] = sum of partial[chunk][49][ch]
__global__ __launch_bounds__(256) void dwconv7_wgrad_reduce_kernel(const float *__restrict__ partial, int chunks, int c,
                                                                   float *__restrict__ dw, float *__restrict__ db) {
    __shared__ float red[4][64];
    // one wave-quarter layout: block = 64 (tap, channel) outputs x 4 chunk lanes
    const int out = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;
    const int total = (DW_TAPS + 1) * c;
    float s = 0.f;
    if (out < total)
        for (int k = part; k < chunks; k += 4) s += partial[(int64_t)k * total + out];
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && out < total) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        const int tap = out / c, ch = out - tap * c;
        if (tap < DW_TAPS) dw[(int64_t)ch * DW_TAPS + tap] = v;
        else if (db) db[ch] = v;
    }
}

}  // namespace s2d

using namespace s2d;

extern "C" int s2d_dwconv7_supported(int channels) { return channels > 0 && channels % 8 == 0 && channels <= 640; }

extern "C" int s2d_dwconv7_nhwc_bf16(const void *x, const float *weight, const float *bias, int n, int h, int w, int c, int flip,
                                     void *y, s2d_stream_t stream) {
    S2D_CHECK_ARG(x && weight && y && n > 0 && h > 0 && w > 0, "dwconv7: bad argument");
    if (!s2d_dwconv7_supported(c)) {
        set_error("dwconv7: unsupported channel count %d", c);
        return S2D_ERR_UNSUPPORTED;
    }
    const int64_t total = (int64_t)n * h * ((w + DW_XT - 1) / DW_XT) * (c / 8);
    const size_t lds = (size_t)DW_TAPS * (c + 4) * sizeof(float);
    const dim3 grid((unsigned)ceil_div(total, 256)), blk(256);
    if (flip) {
        static size_t attr = 48 * 1024;
        if (lds > attr) {
            S2D_HIP(hipFuncSetAttribute((const void *)dwconv7_nhwc_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr = lds;
        }
        hipLaunchKernelGGL(dwconv7_nhwc_bf16_kernel<true>, grid, blk, lds, (hipStream_t)stream, (const __bf16 *)x, weight, bias, n, h, w,
                           c, (__bf16 *)y);
    } else {
        static size_t attr = 48 * 1024;
        if (lds > attr) {
            S2D_HIP(hipFuncSetAttribute((const void *)dwconv7_nhwc_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr = lds;
        }
        hipLaunchKernelGGL(dwconv7_nhwc_bf16_kernel<false>, grid, blk, lds, (hipStream_t)stream, (const __bf16 *)x, weight, bias, n, h,
                           w, c, (__bf16 *)y);
    }
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" size_t s2d_dwconv7_wgrad_workspace_bytes(int n, int h, int w, int c) {
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0) return 0;
    const int64_t chunks = ceil_div((int64_t)n * h * w, DWG_PIX);
    return align_up((size_t)chunks * (DW_TAPS + 1) * c * sizeof(float), 256);
}

extern "C" int s2d_dwconv7_wgrad_nhwc_bf16(const void *x, const void *dy, int n, int h, int w, int c, float *dweight, float *dbias,
                                           void *ws, size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(x && dy && dweight && n > 0 && h > 0 && w > 0, "dwconv7_wgrad: bad argument");
    if (!s2d_dwconv7_supported(c)) {
        set_error("dwconv7_wgrad: unsupported channel count %d", c);
        return S2D_ERR_UNSUPPORTED;
    }
    const size_t need = s2d_dwconv7_wgrad_workspace_bytes(n, h, w, c);
    if (!ws || ws_bytes < need) {
        set_error("dwconv7_wgrad: workspace too small (%zu < %zu)", ws_bytes, need);
        return S2D_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int chunks = (int)ceil_div((int64_t)n * h * w, DWG_PIX);
    float *partial = (float *)ws;
    for (int g0 = 0; g0 < c / 8; g0 += 32) {   // 32 channel groups (256 channels) per launch
        hipLaunchKernelGGL(dwconv7_wgrad_kernel, dim3((unsigned)chunks, DW_K), dim3(256), 0, st, (const __bf16 *)x, (const __bf16 *)dy, n,
                           h, w, c, g0, partial);
    }
    S2D_LAUNCH_CHECK();
    const int total = (DW_TAPS + 1) * c;
    hipLaunchKernelGGL(dwconv7_wgrad_reduce_kernel, dim3((unsigned)ceil_div(total, 64)), dim3(256), 0, st, partial, chunks, c, dweight,
                       dbias);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
