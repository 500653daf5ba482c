// Layout + precision hand-over between the bf16 NHWC neck and the fp32 planar (NCDHW) PCR head of S2D_RPN
// (/root/reference/det3d/models/necks/rpn.py:283-285: `gen = out_conv(F_S_b).view(n, 128, 5, h, w)`): one tiled transpose pass
// per direction through LDS.  torch's strided copy kernels ran these at 0.8 TB/s (0.66 ms forward + 0.58 ms backward at
// [4,640,188,188]); both sides of the tile move as 16-byte accesses, 256 contiguous bytes per channel row.
#include "s2d_common.h"

namespace s2d {

typedef __bf16 bf16x8t __attribute__((ext_vector_type(8)));
constexpr int LT = 64;   // tile: 64 pixels x 64 channels

// grid (pixel tiles, channel tiles, batch).  c % 8 == 0, hw % 4 == 0.
// ld = row stride of the NHWC side in elements (>= c, a multiple of 8): the first c channels of ld-wide rows (r05: a 1x1 conv whose output
// channel count was padded to the tile kernels' multiple of 64 hands its map over without a compaction copy)
__global__ __launch_bounds__(256) void nhwc_bf16_to_nchw_f32_kernel(const __bf16 *__restrict__ x, int c, int ld, int64_t hw, float *__restrict__ y) {
    __shared__ float tile[LT][LT + 1];   // [channel][pixel]
    const int64_t p0 = (int64_t)blockIdx.x * LT;
    const int c0 = blockIdx.y * LT;
    const int64_t b = blockIdx.z;
    const int t = threadIdx.x;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int p = pass * 32 + (t >> 3), g = t & 7;
        if (p0 + p < hw && c0 + 8 * g < c) {
            const bf16x8t v = *reinterpret_cast<const bf16x8t *>(x + ((b * hw + p0 + p) * ld + c0 + 8 * g));
#pragma unroll
            for (int e = 0; e < 8; ++e) tile[8 * g + e][p] = (float)v[e];
        }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int ch = pass * 16 + (t >> 4), pq = t & 15;
        if (c0 + ch < c && p0 + 4 * pq < hw)
            *reinterpret_cast<float4 *>(y + ((b * c + c0 + ch) * hw + p0 + 4 * pq)) =
                float4{tile[ch][4 * pq], tile[ch][4 * pq + 1], tile[ch][4 * pq + 2], tile[ch][4 * pq + 3]};
    }
}

// ld >= c: rows of ld channels are written, channels c .. ld-1 as zeros (grid.y covers ld)
__global__ __launch_bounds__(256) void nchw_f32_to_nhwc_bf16_kernel(const float *__restrict__ x, int c, int ld, int64_t hw, __bf16 *__restrict__ y) {
    __shared__ float tile[LT][LT + 1];
    const int64_t p0 = (int64_t)blockIdx.x * LT;
    const int c0 = blockIdx.y * LT;
    const int64_t b = blockIdx.z;
    const int t = threadIdx.x;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int ch = pass * 16 + (t >> 4), pq = t & 15;
        if (c0 + ch < c && p0 + 4 * pq < hw) {
            const float4 v = *reinterpret_cast<const float4 *>(x + ((b * c + c0 + ch) * hw + p0 + 4 * pq));
            tile[ch][4 * pq] = v.x; tile[ch][4 * pq + 1] = v.y; tile[ch][4 * pq + 2] = v.z; tile[ch][4 * pq + 3] = v.w;
        } else if (c0 + ch >= c) {
            tile[ch][4 * pq] = 0.f; tile[ch][4 * pq + 1] = 0.f; tile[ch][4 * pq + 2] = 0.f; tile[ch][4 * pq + 3] = 0.f;
        }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int p = pass * 32 + (t >> 3), g = t & 7;
        if (p0 + p < hw && c0 + 8 * g < ld) {
            bf16x8t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)tile[8 * g + e][p];
            *reinterpret_cast<bf16x8t *>(y + ((b * hw + p0 + p) * ld + c0 + 8 * g)) = o;
        }
    }
}

static int layout_check(const void *x, const void *y, int batch, int c, int64_t hw, const char *who) {
    if (!x || !y || batch <= 0 || batch > 65535 || c <= 0 || hw <= 0) {
        set_error("%s: bad argument", who);
        return S2D_ERR_INVALID_ARG;
    }
    if ((c & 7) || (hw & 3)) {
        set_error("%s: channels %d must be a multiple of 8 and pixels %lld a multiple of 4", who, c, (long long)hw);
        return S2D_ERR_UNSUPPORTED;
    }
    return 0;
}

}  // namespace s2d

using namespace s2d;

/* x bf16 [batch][hw][c] (NHWC) -> y fp32 [batch][c][hw] (NCHW) */
extern "C" int s2d_nhwc_bf16_to_nchw_f32(const void *x, int batch, int c, int64_t hw, float *y, s2d_stream_t stream) {
    int rc = layout_check(x, y, batch, c, hw, "nhwc_bf16_to_nchw_f32");
    if (rc) return rc;
    const dim3 grid((unsigned)ceil_div(hw, LT), (unsigned)ceil_div(c, LT), batch);
    hipLaunchKernelGGL(nhwc_bf16_to_nchw_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16 *)x, c, c, hw, y);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

/* the same from rows of `ld` >= c channels (ld % 8 == 0): x bf16 [batch][hw][ld], channels 0 .. c-1 -> y fp32 [batch][c][hw] */
extern "C" int s2d_nhwc_bf16_to_nchw_f32_ld(const void *x, int batch, int c, int ld, int64_t hw, float *y, s2d_stream_t stream) {
    int rc = layout_check(x, y, batch, c, hw, "nhwc_bf16_to_nchw_f32_ld");
    if (rc) return rc;
    S2D_CHECK_ARG(ld >= c && ld % 8 == 0, "nhwc_bf16_to_nchw_f32_ld: row stride must be a multiple of 8 and >= c");
    const dim3 grid((unsigned)ceil_div(hw, LT), (unsigned)ceil_div(c, LT), batch);
    hipLaunchKernelGGL(nhwc_bf16_to_nchw_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16 *)x, c, ld, hw, y);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

/* x fp32 [batch][c][hw] (NCHW) -> y bf16 [batch][hw][c] (NHWC) */
extern "C" int s2d_nchw_f32_to_nhwc_bf16(const float *x, int batch, int c, int64_t hw, void *y, s2d_stream_t stream) {
    int rc = layout_check(x, y, batch, c, hw, "nchw_f32_to_nhwc_bf16");
    if (rc) return rc;
    const dim3 grid((unsigned)ceil_div(hw, LT), (unsigned)ceil_div(c, LT), batch);
    hipLaunchKernelGGL(nchw_f32_to_nhwc_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, c, c, hw, (__bf16 *)y);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

/* the same into rows of `ld` >= c channels (ld % 8 == 0), channels c .. ld-1 zero-filled: y bf16 [batch][hw][ld] */
extern "C" int s2d_nchw_f32_to_nhwc_bf16_ld(const float *x, int batch, int c, int ld, int64_t hw, void *y, s2d_stream_t stream) {
    int rc = layout_check(x, y, batch, c, hw, "nchw_f32_to_nhwc_bf16_ld");
    if (rc) return rc;
    S2D_CHECK_ARG(ld >= c && ld % 8 == 0, "nchw_f32_to_nhwc_bf16_ld: row stride must be a multiple of 8 and >= c");
    const dim3 grid((unsigned)ceil_div(hw, LT), (unsigned)ceil_div(ld, LT), batch);
    hipLaunchKernelGGL(nchw_f32_to_nhwc_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, c, ld, hw, (__bf16 *)y);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

// =====================================================================================================================================
// 2 x 2 resampling of NHWC bf16 maps (r04; the pillar S2D module, /root/reference/det3d/models/readers/point_pillars.py S2D module:
// nn.MaxPool2d(2, 2) in front of encoder_1, nn.Upsample(scale_factor=2) behind decoder_2).  A thread moves one 16-byte group of 8
// channels; torch's NHWC kernels ran these at 0.8-1 TB/s (up-sampling 164 us forward / 89 us backward, max-pool 66 / 135 us at
// [4,64,468,468]).
// =====================================================================================================================================
namespace s2d {

__global__ __launch_bounds__(256) void upsample2x_nhwc_bf16_kernel(const __bf16 *__restrict__ x, int64_t groups_out, int h, int w, int c8, __bf16 *__restrict__ y) {
    // output element group index = ((n * 2h + oy) * 2w + ox) * c8 + g
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < groups_out; i += (int64_t)gridDim.x * 256) {
        const int g = (int)(i % c8);
        int64_t r = i / c8;
        const int ox = (int)(r % (2 * w));
        r /= 2 * w;
        const int oy = (int)(r % (2 * h));
        const int64_t n = r / (2 * h);
        reinterpret_cast<bf16x8t *>(y)[i] = reinterpret_cast<const bf16x8t *>(x)[((n * h + (oy >> 1)) * w + (ox >> 1)) * c8 + g];
    }
}

// dx[i][j] = sum of the four dy it was copied to (fp32 accumulation, one rounding: torch's backward accumulates in float too)
__global__ __launch_bounds__(256) void upsample2x_bwd_nhwc_bf16_kernel(const __bf16 *__restrict__ dy, int64_t groups_in, int h, int w, int c8,
                                                                       __bf16 *__restrict__ dx) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < groups_in; i += (int64_t)gridDim.x * 256) {
        const int g = (int)(i % c8);
        int64_t r = i / c8;
        const int xx = (int)(r % w);
        r /= w;
        const int yy = (int)(r % h);
        const int64_t n = r / h;
        const int64_t row0 = ((n * 2 * h + 2 * yy) * 2 * w + 2 * xx) * c8 + g, row1 = row0 + (int64_t)2 * w * c8;
        const bf16x8t a = reinterpret_cast<const bf16x8t *>(dy)[row0], b = reinterpret_cast<const bf16x8t *>(dy)[row0 + c8];
        const bf16x8t cc = reinterpret_cast<const bf16x8t *>(dy)[row1], d = reinterpret_cast<const bf16x8t *>(dy)[row1 + c8];
        bf16x8t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)((((float)a[e] + (float)b[e]) + (float)cc[e]) + (float)d[e]);
        reinterpret_cast<bf16x8t *>(dx)[i] = o;
    }
}

// torch's window scan: (0,0), (0,1), (1,0), (1,1); a later element replaces the maximum when it is larger or NaN
__device__ __forceinline__ bool pool_takes(float v, float m) { return v > m || v != v; }

__global__ __launch_bounds__(256) void maxpool2x2_nhwc_bf16_kernel(const __bf16 *__restrict__ x, int64_t groups_out, int ho, int wo, int w, int c8,
                                                                   int64_t in_plane_groups, __bf16 *__restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < groups_out; i += (int64_t)gridDim.x * 256) {
        const int g = (int)(i % c8);
        int64_t r = i / c8;
        const int ox = (int)(r % wo);
        r /= wo;
        const int oy = (int)(r % ho);
        const int64_t n = r / ho;
        const int64_t row0 = n * in_plane_groups + ((int64_t)(2 * oy) * w + 2 * ox) * c8 + g, row1 = row0 + (int64_t)w * c8;
        const bf16x8t a = reinterpret_cast<const bf16x8t *>(x)[row0], b = reinterpret_cast<const bf16x8t *>(x)[row0 + c8];
        const bf16x8t cc = reinterpret_cast<const bf16x8t *>(x)[row1], d = reinterpret_cast<const bf16x8t *>(x)[row1 + c8];
        bf16x8t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float m = (float)a[e];
            if (pool_takes((float)b[e], m)) m = (float)b[e];
            if (pool_takes((float)cc[e], m)) m = (float)cc[e];
            if (pool_takes((float)d[e], m)) m = (float)d[e];
            o[e] = (__bf16)m;
        }
        reinterpret_cast<bf16x8t *>(y)[i] = o;
    }
}

// the gradient goes to the window element the forward scan selected (re-derived from x: no index tensor); rows / columns of an odd-sized
// input that no window covers get zero
__global__ __launch_bounds__(256) void maxpool2x2_bwd_nhwc_bf16_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ dy, int64_t groups_in, int h,
                                                                       int w, int ho, int wo, int c8, __bf16 *__restrict__ dx) {
    // a thread owns one WINDOW position group (the four input groups of an output group); leftover rows / columns are zeroed by their own threads
    const int hw2 = (h + 1) / 2, ww2 = (w + 1) / 2;
    const int64_t items = groups_in / ((int64_t)h * w) * hw2 * ww2;   // groups_in / (h w) = n * c8
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (int64_t)gridDim.x * 256) {
        const int g = (int)(i % c8);
        int64_t r = i / c8;
        const int ox = (int)(r % ww2);
        r /= ww2;
        const int oy = (int)(r % hw2);
        const int64_t n = r / hw2;
        const int64_t row0 = ((n * h + 2 * oy) * w + 2 * ox) * c8 + g, row1 = row0 + (int64_t)w * c8;
        const bool in_x = 2 * ox + 1 < w, in_y = 2 * oy + 1 < h, covered = ox < wo && oy < ho;
        bf16x8t zero;
#pragma unroll
        for (int e = 0; e < 8; ++e) zero[e] = (__bf16)0.f;
        if (!covered) {
            reinterpret_cast<bf16x8t *>(dx)[row0] = zero;
            if (in_x) reinterpret_cast<bf16x8t *>(dx)[row0 + c8] = zero;
            if (in_y) reinterpret_cast<bf16x8t *>(dx)[row1] = zero;
            if (in_x && in_y) reinterpret_cast<bf16x8t *>(dx)[row1 + c8] = zero;
            continue;
        }
        const bf16x8t a = reinterpret_cast<const bf16x8t *>(x)[row0], b = reinterpret_cast<const bf16x8t *>(x)[row0 + c8];
        const bf16x8t cc = reinterpret_cast<const bf16x8t *>(x)[row1], d = reinterpret_cast<const bf16x8t *>(x)[row1 + c8];
        const bf16x8t gy = reinterpret_cast<const bf16x8t *>(dy)[((n * ho + oy) * wo + ox) * c8 + g];
        bf16x8t oa = zero, ob = zero, oc = zero, od = zero;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float m = (float)a[e];
            int sel = 0;
            if (pool_takes((float)b[e], m)) { m = (float)b[e]; sel = 1; }
            if (pool_takes((float)cc[e], m)) { m = (float)cc[e]; sel = 2; }
            if (pool_takes((float)d[e], m)) { m = (float)d[e]; sel = 3; }
            oa[e] = sel == 0 ? gy[e] : (__bf16)0.f;
            ob[e] = sel == 1 ? gy[e] : (__bf16)0.f;
            oc[e] = sel == 2 ? gy[e] : (__bf16)0.f;
            od[e] = sel == 3 ? gy[e] : (__bf16)0.f;
        }
        reinterpret_cast<bf16x8t *>(dx)[row0] = oa;
        reinterpret_cast<bf16x8t *>(dx)[row0 + c8] = ob;
        reinterpret_cast<bf16x8t *>(dx)[row1] = oc;
        reinterpret_cast<bf16x8t *>(dx)[row1 + c8] = od;
    }
}

// space-to-depth / depth-to-space with 2 x 2 blocks on NHWC bf16 (r06): the "depth" image is [n][h][w][(py, px, c)], the "space" image
// [n][2h][2w][c]; a thread moves one 16-byte group of 8 channels.  (torch's strided copy of the permuted view ran these 72 MB moves at
// 1.9 TB/s: four 39 us launches per step around the 2x2 / stride-2 conv and the 2x2 transposed conv.)
template <bool TO_DEPTH>
__global__ __launch_bounds__(256) void space_depth2_kernel(const __bf16 *__restrict__ src, int64_t groups, int h, int w, int c8, __bf16 *__restrict__ dst) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < groups; i += stride) {
        const int g = (int)(i % c8);
        int64_t r = i / c8;
        const int px = (int)(r & 1), py = (int)((r >> 1) & 1);
        r >>= 2;
        const int x = (int)(r % w);
        r /= w;
        const int y = (int)(r % h);
        const int64_t n = r / h;
        const int64_t sp = ((n * 2 * h + 2 * y + py) * (2 * (int64_t)w) + 2 * x + px) * c8 + g;   // the group's place in the space image
        if (TO_DEPTH) reinterpret_cast<bf16x8t *>(dst)[i] = reinterpret_cast<const bf16x8t *>(src)[sp];
        else reinterpret_cast<bf16x8t *>(dst)[sp] = reinterpret_cast<const bf16x8t *>(src)[i];
    }
}

static int resample_check(const void *a, const void *b, int n, int h, int w, int c, const char *who) {
    S2D_CHECK_ARG(a && b && n > 0 && h > 0 && w > 0 && c > 0, "%s: bad argument", who);
    if (c % 8) {
        set_error("%s: %d channels (multiples of 8 only)", who, c);
        return S2D_ERR_UNSUPPORTED;
    }
    return 0;
}
static unsigned resample_blocks(int64_t items) { return (unsigned)std::min<int64_t>(ceil_div(items, 256), 256 * 32); }

}  // namespace s2d

/* y[n][2h][2w][c] = x[n][h][w][c] copied to its 2 x 2 block (nn.Upsample(scale_factor=2, mode="nearest") on NHWC bf16) and its backward */
extern "C" int s2d_upsample2x_nhwc_bf16(const void *x, int n, int h, int w, int c, void *y, s2d_stream_t stream) {
    int rc = resample_check(x, y, n, h, w, c, "upsample2x_nhwc_bf16");
    if (rc) return rc;
    const int64_t groups = (int64_t)n * 4 * h * w * (c / 8);
    hipLaunchKernelGGL(upsample2x_nhwc_bf16_kernel, dim3(resample_blocks(groups)), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)x, groups, h, w, c / 8,
                       (__bf16 *)y);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
extern "C" int s2d_upsample2x_bwd_nhwc_bf16(const void *dy, int n, int h, int w, int c, void *dx, s2d_stream_t stream) {
    int rc = resample_check(dy, dx, n, h, w, c, "upsample2x_bwd_nhwc_bf16");
    if (rc) return rc;
    const int64_t groups = (int64_t)n * h * w * (c / 8);
    hipLaunchKernelGGL(upsample2x_bwd_nhwc_bf16_kernel, dim3(resample_blocks(groups)), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)dy, groups, h, w,
                       c / 8, (__bf16 *)dx);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
/* nn.MaxPool2d(2, 2) on NHWC bf16: y[n][h/2][w/2][c]; the backward re-derives the selected element from x (no index tensor) */
extern "C" int s2d_maxpool2x2_nhwc_bf16(const void *x, int n, int h, int w, int c, void *y, s2d_stream_t stream) {
    int rc = resample_check(x, y, n, h, w, c, "maxpool2x2_nhwc_bf16");
    if (rc) return rc;
    S2D_CHECK_ARG(h >= 2 && w >= 2, "maxpool2x2_nhwc_bf16: input smaller than the window");
    const int ho = h / 2, wo = w / 2;
    const int64_t groups = (int64_t)n * ho * wo * (c / 8);
    hipLaunchKernelGGL(maxpool2x2_nhwc_bf16_kernel, dim3(resample_blocks(groups)), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)x, groups, ho, wo, w,
                       c / 8, (int64_t)h * w * (c / 8), (__bf16 *)y);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
extern "C" int s2d_maxpool2x2_bwd_nhwc_bf16(const void *x, const void *dy, int n, int h, int w, int c, void *dx, s2d_stream_t stream) {
    int rc = resample_check(x, dx, n, h, w, c, "maxpool2x2_bwd_nhwc_bf16");
    if (rc) return rc;
    S2D_CHECK_ARG(dy && h >= 2 && w >= 2, "maxpool2x2_bwd_nhwc_bf16: bad argument");
    const int64_t groups_in = (int64_t)n * h * w * (c / 8);
    const int64_t items = (int64_t)n * ((h + 1) / 2) * ((w + 1) / 2) * (c / 8);
    hipLaunchKernelGGL(maxpool2x2_bwd_nhwc_bf16_kernel, dim3(resample_blocks(items)), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)x,
                       (const __bf16 *)dy, groups_in, h, w, h / 2, w / 2, c / 8, (__bf16 *)dx);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

/* 2 x 2 space-to-depth on NHWC bf16: x[n][2h][2w][c] -> y[n][h][w][(py, px, c)] (to_depth != 0) or the inverse (to_depth == 0: x is the depth image);
   h, w = the extent of the DEPTH image.  The rearrangement around nn.Conv2d(k=2, s=2) / nn.ConvTranspose2d(k=2, s=2) run as 1x1 tile kernels
   (rpn.py:188 encoder_1[0], rpn.py:92-104 deblocks; replaces torch permute + contiguous copies there). */
extern "C" int s2d_space_depth2_nhwc_bf16(const void *x, int n, int h, int w, int c, int to_depth, void *y, s2d_stream_t stream) {
    int rc = resample_check(x, y, n, h, w, c, "space_depth2_nhwc_bf16");
    if (rc) return rc;
    const int64_t groups = (int64_t)n * h * w * 4 * (c / 8);
    if (to_depth)
        hipLaunchKernelGGL(space_depth2_kernel<true>, dim3(resample_blocks(groups)), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)x, groups, h, w, c / 8,
                           (__bf16 *)y);
    else
        hipLaunchKernelGGL(space_depth2_kernel<false>, dim3(resample_blocks(groups)), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)x, groups, h, w, c / 8,
                           (__bf16 *)y);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
