// Layout + precision hand-over between the bf16 NHWC neck and the fp32 planar (NCDHW) PCR head of S2D_RPN
// (/root/reference/det3d/models/necks/rpn.py:283-285: `gen = out_conv(F_S_b).view(n, 128, 5, h, w)`): one tiled transpose pass
// per direction through LDS.  torch's strided copy kernels ran these at 0.8 TB/s (0.66 ms forward + 0.58 ms backward at
// [4,640,188,188]); both sides of the tile move as 16-byte accesses, 256 contiguous bytes per channel row.
#include "s2d_common.h"

namespace s2d {

typedef __bf16 bf16x8t __attribute__((ext_vector_type(8)));
constexpr int LT = 64;   // tile: 64 pixels x 64 channels

// grid (pixel tiles, channel tiles, batch).  c % 8 == 0, hw % 4 == 0.
__global__ __launch_bounds__(256) void nhwc_bf16_to_nchw_f32_kernel(const __bf16 *__restrict__ x, int c, int64_t hw, float *__restrict__ y) {
    __shared__ float tile[LT][LT + 1];   // [channel][pixel]
    const int64_t p0 = (int64_t)blockIdx.x * LT;
    const int c0 = blockIdx.y * LT;
    const int64_t b = blockIdx.z;
    const int t = threadIdx.x;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int p = pass * 32 + (t >> 3), g = t & 7;
        if (p0 + p < hw && c0 + 8 * g < c) {
            const bf16x8t v = *reinterpret_cast<const bf16x8t *>(x + ((b * hw + p0 + p) * c + c0 + 8 * g));
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

__global__ __launch_bounds__(256) void nchw_f32_to_nhwc_bf16_kernel(const float *__restrict__ x, int c, int64_t hw, __bf16 *__restrict__ y) {
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
        }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int p = pass * 32 + (t >> 3), g = t & 7;
        if (p0 + p < hw && c0 + 8 * g < c) {
            bf16x8t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)tile[8 * g + e][p];
            *reinterpret_cast<bf16x8t *>(y + ((b * hw + p0 + p) * c + c0 + 8 * g)) = o;
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
    hipLaunchKernelGGL(nhwc_bf16_to_nchw_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const __bf16 *)x, c, hw, y);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

/* x fp32 [batch][c][hw] (NCHW) -> y bf16 [batch][hw][c] (NHWC) */
extern "C" int s2d_nchw_f32_to_nhwc_bf16(const float *x, int batch, int c, int64_t hw, void *y, s2d_stream_t stream) {
    int rc = layout_check(x, y, batch, c, hw, "nchw_f32_to_nhwc_bf16");
    if (rc) return rc;
    const dim3 grid((unsigned)ceil_div(hw, LT), (unsigned)ceil_div(c, LT), batch);
    hipLaunchKernelGGL(nchw_f32_to_nhwc_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, c, hw, (__bf16 *)y);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
