// 3x3 convolutions with a handful of output channels: the final convs of the CenterHead branches
// (/root/reference/det3d/models/bbox_heads/center_head.py:33-61 SepHead: Conv2d(64, classes, 3, padding=1) with classes = 1..3 for
// reg / height / dim / rot / hm, on the 188 x 188 map).  With 1-3 output channels these are streaming reductions over a 64-channel
// NHWC bf16 map (18 MB), not GEMMs: MIOpen's implicit-GEMM kernels ran them at 65 us forward and 80-116 us backward each (0.77 ms
// per step for the five branches).  Here:
//   fwd   thread = output pixel: 9 taps x cin/8 16-byte loads, KO accumulators, weights broadcast from LDS; output written as fp32
//         planar [n][KO][h][w] - what the losses read (no bf16 round trip of the predictions)
//   dgrad thread = (pixel, 8-channel group): 9 taps x KO fp32 gradient values (coalesced planar reads) -> one 16-byte bf16 store
//   wgrad thread = (8-channel group, tap) worker x pixel lane: acc[8][KO] over a strip of pixels, fixed-order block fold, per-block
//         partials [blocks][KO][cin][9] (+ bias gradient) reduced by a second kernel
// x, dx: bf16 NHWC; weight fp32 [KO][cin][3][3] (torch layout); padding 1, stride 1; cin % 8 == 0, cin <= 128; KO <= 4.
#include "s2d_common.h"

namespace s2d {

typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
constexpr int SC_MAX_CIN = 128, SC_WG_BLOCKS = 512;

template <int KO>
__global__ __launch_bounds__(256) void smallconv_fwd_kernel(const __bf16 *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                            int n_img, int H, int W, int cin, float *__restrict__ y) {
    __shared__ float ws[9 * SC_MAX_CIN * KO];   // [tap][c][k]
    for (int i = threadIdx.x; i < 9 * cin * KO; i += 256) {
        const int k = i % KO, c = (i / KO) % cin, tap = i / (KO * cin);
        ws[i] = w[((int64_t)k * cin + c) * 9 + tap];
    }
    __syncthreads();
    const int64_t hw = (int64_t)H * W, total = hw * n_img;
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= total) return;
    const int px = (int)(m % W), py = (int)((m / W) % H);
    const int64_t img = m / hw;
    float acc[KO];
#pragma unroll
    for (int k = 0; k < KO; ++k) acc[k] = bias ? bias[k] : 0.f;
    const int groups = cin >> 3;
    for (int tap = 0; tap < 9; ++tap) {
        const int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
        if ((unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W) continue;
        const __bf16 *row = x + ((img * H + yy) * W + xx) * cin;
        const float *wt = ws + tap * cin * KO;
        for (int g = 0; g < groups; ++g) {
            const bf16x8s v = reinterpret_cast<const bf16x8s *>(row)[g];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xf = (float)v[e];
#pragma unroll
                for (int k = 0; k < KO; ++k) acc[k] = fmaf(xf, wt[(g * 8 + e) * KO + k], acc[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KO; ++k) y[(img * KO + k) * hw + (int64_t)py * W + px] = acc[k];
}

// dx[pix][c] = sum_tap sum_k dy[k][pix - (tap offset)] * w[k][c][tap]   (the output pixel pix - delta had pix as its tap `tap`)
template <int KO>
__global__ __launch_bounds__(256) void smallconv_dgrad_kernel(const float *__restrict__ dy, const float *__restrict__ w, int n_img, int H, int W,
                                                              int cin, __bf16 *__restrict__ dx) {
    __shared__ float ws[9 * SC_MAX_CIN * KO];   // [tap][k][c]
    for (int i = threadIdx.x; i < 9 * cin * KO; i += 256) {
        const int c = i % cin, k = (i / cin) % KO, tap = i / (KO * cin);
        ws[i] = w[((int64_t)k * cin + c) * 9 + tap];
    }
    __syncthreads();
    const int groups = cin >> 3;
    const int64_t hw = (int64_t)H * W, total = hw * n_img * groups;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int g = (int)(t % groups);
    const int64_t m = t / groups;
    const int px = (int)(m % W), py = (int)((m / W) % H);
    const int64_t img = m / hw;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
        const int oy = py - (tap / 3 - 1), ox = px - (tap % 3 - 1);   // output pixel whose tap `tap` reads this input pixel
        if ((unsigned)oy >= (unsigned)H || (unsigned)ox >= (unsigned)W) continue;
#pragma unroll
        for (int k = 0; k < KO; ++k) {
            const float d = dy[(img * KO + k) * hw + (int64_t)oy * W + ox];
            const float *wt = ws + (tap * KO + k) * cin + g * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(d, wt[e], acc[e]);
        }
    }
    bf16x8s o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)acc[e];
    reinterpret_cast<bf16x8s *>(dx + m * cin)[g] = o;
}

// block = strip of pixels; thread = worker (g, tap) x pixel lane.  partial[block][KO][cin][9] and partial_b[block][KO]
template <int KO>
__global__ __launch_bounds__(256) void smallconv_wgrad_kernel(const __bf16 *__restrict__ x, const float *__restrict__ dy, int n_img, int H, int W,
                                                              int cin, int64_t pix_per_block, float *__restrict__ partial,
                                                              float *__restrict__ partial_b) {
    extern __shared__ float red[];   // [lanes][workers][8][KO] fold buffer
    const int groups = cin >> 3, workers = groups * 9, lanes = 256 / workers;
    const int wk = threadIdx.x % workers, ln = threadIdx.x / workers;
    const int g = wk / 9, tap = wk % 9;
    const int64_t hw = (int64_t)H * W, total = hw * n_img;
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block, p1 = p0 + pix_per_block < total ? p0 + pix_per_block : total;
    float acc[8][KO], bsum[KO];
#pragma unroll
    for (int k = 0; k < KO; ++k) {
        bsum[k] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e][k] = 0.f;
    }
    if (ln < lanes) {
        for (int64_t m = p0 + ln; m < p1; m += lanes) {
            const int px = (int)(m % W), py = (int)((m / W) % H);
            const int64_t img = m / hw;
            float d[KO];
#pragma unroll
            for (int k = 0; k < KO; ++k) d[k] = dy[(img * KO + k) * hw + (int64_t)py * W + px];
            if (wk == 0) {
#pragma unroll
                for (int k = 0; k < KO; ++k) bsum[k] += d[k];
            }
            const int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
            if ((unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W) continue;
            const bf16x8s v = reinterpret_cast<const bf16x8s *>(x + ((img * H + yy) * W + xx) * cin)[g];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xf = (float)v[e];
#pragma unroll
                for (int k = 0; k < KO; ++k) acc[e][k] = fmaf(d[k], xf, acc[e][k]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int k = 0; k < KO; ++k) red[((ln * workers + wk) * 8 + e) * KO + k] = acc[e][k];
    }
    __syncthreads();
    // fold the pixel lanes in order; element (k, c = g*8+e, tap)
    for (int i = threadIdx.x; i < workers * 8 * KO; i += 256) {
        const int k = i % KO, e = (i / KO) % 8, w2 = i / (KO * 8);
        float s = 0.f;
        for (int l = 0; l < lanes; ++l) s += red[((l * workers + w2) * 8 + e) * KO + k];
        const int gg = w2 / 9, tt = w2 % 9;
        partial[(((int64_t)blockIdx.x * KO + k) * cin + gg * 8 + e) * 9 + tt] = s;
    }
    // bias gradient: worker 0 of every lane holds a partial sum
    __syncthreads();
    if (wk == 0 && ln < lanes) {
#pragma unroll
        for (int k = 0; k < KO; ++k) red[ln * KO + k] = bsum[k];
    }
    __syncthreads();
    if (threadIdx.x < KO) {
        float s = 0.f;
        for (int l = 0; l < lanes; ++l) s += red[l * KO + threadIdx.x];
        partial_b[(int64_t)blockIdx.x * KO + threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(256) void smallconv_wgrad_reduce_kernel(const float *__restrict__ partial, const float *__restrict__ partial_b, int blocks,
                                                                     int n_w, int ko, float *__restrict__ dw, float *__restrict__ db) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_w) {
        float s = 0.f;
        for (int b = 0; b < blocks; ++b) s += partial[(int64_t)b * n_w + i];
        dw[i] = s;
    } else if (i < n_w + ko && db) {
        const int k = i - n_w;
        float s = 0.f;
        for (int b = 0; b < blocks; ++b) s += partial_b[(int64_t)b * ko + k];
        db[k] = s;
    }
}

static bool sc_ok(int cin, int ko) { return cin >= 8 && cin % 8 == 0 && cin <= SC_MAX_CIN && ko >= 1 && ko <= 4; }

}  // namespace s2d

using namespace s2d;

extern "C" int s2d_smallconv3x3_supported(int cin, int cout) { return sc_ok(cin, cout); }

/* y fp32 [n][cout][h][w] = conv3x3(x bf16 NHWC [n][h][w][cin], weight fp32 [cout][cin][3][3], padding 1) + bias */
extern "C" int s2d_smallconv3x3_fwd(const void *x, const float *weight, const float *bias, int n_img, int h, int w, int cin, int cout, float *y,
                                    s2d_stream_t stream) {
    S2D_CHECK_ARG(x && weight && y && n_img > 0 && h > 0 && w > 0, "smallconv3x3_fwd: bad argument");
    if (!sc_ok(cin, cout)) {
        set_error("smallconv3x3: unsupported channels %d -> %d", cin, cout);
        return S2D_ERR_UNSUPPORTED;
    }
    const int64_t total = (int64_t)n_img * h * w;
    const dim3 grid((unsigned)ceil_div(total, 256)), blk(256);
    hipStream_t st = (hipStream_t)stream;
#define S2D_SC_FWD(K) hipLaunchKernelGGL(smallconv_fwd_kernel<K>, grid, blk, 0, st, (const __bf16 *)x, weight, bias, n_img, h, w, cin, y)
    switch (cout) {
        case 1: S2D_SC_FWD(1); break;
        case 2: S2D_SC_FWD(2); break;
        case 3: S2D_SC_FWD(3); break;
        default: S2D_SC_FWD(4); break;
    }
#undef S2D_SC_FWD
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

/* dx bf16 NHWC [n][h][w][cin] from dy fp32 [n][cout][h][w] */
extern "C" int s2d_smallconv3x3_dgrad(const float *dy, const float *weight, int n_img, int h, int w, int cin, int cout, void *dx,
                                      s2d_stream_t stream) {
    S2D_CHECK_ARG(dy && weight && dx && n_img > 0 && h > 0 && w > 0, "smallconv3x3_dgrad: bad argument");
    if (!sc_ok(cin, cout)) return S2D_ERR_UNSUPPORTED;
    const int64_t total = (int64_t)n_img * h * w * (cin / 8);
    const dim3 grid((unsigned)ceil_div(total, 256)), blk(256);
    hipStream_t st = (hipStream_t)stream;
#define S2D_SC_DG(K) hipLaunchKernelGGL(smallconv_dgrad_kernel<K>, grid, blk, 0, st, dy, weight, n_img, h, w, cin, (__bf16 *)dx)
    switch (cout) {
        case 1: S2D_SC_DG(1); break;
        case 2: S2D_SC_DG(2); break;
        case 3: S2D_SC_DG(3); break;
        default: S2D_SC_DG(4); break;
    }
#undef S2D_SC_DG
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" size_t s2d_smallconv3x3_wgrad_workspace_bytes(int cin, int cout) {
    if (!sc_ok(cin, cout)) return 0;
    return align_up((size_t)SC_WG_BLOCKS * cout * (cin * 9 + 1) * sizeof(float), 256);
}

/* dweight fp32 [cout][cin][3][3], dbias fp32 [cout] (optional) */
extern "C" int s2d_smallconv3x3_wgrad(const void *x, const float *dy, int n_img, int h, int w, int cin, int cout, float *dweight, float *dbias,
                                      void *ws, size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(x && dy && dweight && n_img > 0 && h > 0 && w > 0, "smallconv3x3_wgrad: bad argument");
    if (!sc_ok(cin, cout)) return S2D_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < s2d_smallconv3x3_wgrad_workspace_bytes(cin, cout)) {
        set_error("smallconv3x3_wgrad: workspace too small");
        return S2D_ERR_WORKSPACE;
    }
    const int64_t total = (int64_t)n_img * h * w;
    const int blocks = (int)std::min<int64_t>(SC_WG_BLOCKS, ceil_div(total, 64));
    const int64_t ppb = ceil_div(total, blocks);
    const int workers = (cin / 8) * 9, lanes = 256 / workers;
    S2D_CHECK_ARG(lanes >= 1, "smallconv3x3_wgrad: too many input channels");
    float *partial = (float *)ws, *partial_b = partial + (size_t)SC_WG_BLOCKS * cout * cin * 9;
    const size_t lds = (size_t)std::max(lanes * workers * 8 * cout, lanes * cout) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
#define S2D_SC_WG(K)                                                                                                                         \
    hipLaunchKernelGGL(smallconv_wgrad_kernel<K>, dim3(blocks), dim3(256), lds, st, (const __bf16 *)x, dy, n_img, h, w, cin, ppb, partial, \
                       partial_b)
    switch (cout) {
        case 1: S2D_SC_WG(1); break;
        case 2: S2D_SC_WG(2); break;
        case 3: S2D_SC_WG(3); break;
        default: S2D_SC_WG(4); break;
    }
#undef S2D_SC_WG
    const int n_w = cout * cin * 9;
    hipLaunchKernelGGL(smallconv_wgrad_reduce_kernel, dim3((unsigned)ceil_div(n_w + cout, 256)), dim3(256), 0, st, partial, partial_b, blocks, n_w, cout,
                       dweight, dbias);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
