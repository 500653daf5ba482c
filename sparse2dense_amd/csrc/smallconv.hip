// 3x3 convolutions with a handful of output channels: the final convs of the CenterHead branches
// (/root/reference/det3d/models/bbox_heads/center_head.py:33-61 SepHead: Conv2d(64, classes, 3, padding=1) with classes = 1..3 for
// reg / height / dim / rot / hm, on the 188 x 188 map).  With 1-3 output channels these are streaming reductions over a 64-channel
// NHWC bf16 map (18 MB), not GEMMs: MIOpen's implicit-GEMM kernels ran them at 65 us forward and 80-116 us backward each (0.77 ms
// per step for the five branches).  Here:
//   fwd   thread = (4 pixels of a row, 8-channel group): 3 x 6 16-byte loads issued together (the cin/8 lanes of a pixel read one
//         contiguous row), each tap's weights read from LDS once for the four pixels, cross-lane fold; output written as fp32 planar [n][KO][h][w] - what the losses read (no bf16
//         round trip of the predictions)
//   dgrad thread = (4 pixels of a row, 8-channel group): 3 x 6 x KO fp32 gradient values, weights from LDS once per four pixels ->
//         four 16-byte bf16 stores
//   wgrad thread = (kernel row, 8-channel group) worker x 8 pixel lanes: acc[3 taps][8][KO] over a run of image rows (the input row is
//         read 3x, not 9x), shuffle fold, per-block partials [blocks][KO*cin*9 + KO] (weights, bias) reduced by a second kernel
// x, dx: bf16 NHWC; weight fp32 [KO][cin][3][3] (torch layout); padding 1, stride 1; cin in {8, 16, 32, 64, 128}; KO <= 4.
#include "s2d_common.h"

namespace s2d {

typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
constexpr int SC_MAX_CIN = 128, SC_WG_BLOCKS = 384, SC_WG_LANES = 8;

// thread = (four consecutive pixels of a row, 8-channel group g): the 3 x 6 input vectors the four pixels touch are loaded once
// (the cin/8 lanes of a pixel read one contiguous 2*cin-byte row), every tap's 8 x KO weights are read from LDS once for the four
// pixels, and the cin/8 partial sums are folded with cross-lane shuffles (cin/8 is a power of two).
template <int KO>
__global__ __launch_bounds__(256) void smallconv_fwd_kernel(const __bf16 *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                            int n_img, int H, int W, int cin, int lg, float *__restrict__ y) {
    __shared__ float ws[9 * SC_MAX_CIN * KO];   // [tap][c][k]
    for (int i = threadIdx.x; i < 9 * cin * KO; i += 256) {
        const int k = i % KO, c = (i / KO) % cin, tap = i / (KO * cin);
        ws[i] = w[(k * cin + c) * 9 + tap];
    }
    __syncthreads();
    const unsigned qpr = (unsigned)(W + 3) >> 2, n_quads = qpr * H * n_img;
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    const unsigned g = t & ((1u << lg) - 1), q0 = t >> lg;
    const bool live = q0 < n_quads;
    const unsigned q = live ? q0 : 0;
    const unsigned line = q / qpr;
    const int px0 = (int)(q - line * qpr) * 4;
    const unsigned img = line / (unsigned)H;
    const int py = (int)(line - img * H);
    bf16x8s v[3][6];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int yy = py + r - 1, xx = px0 + c - 1;
            const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            const int64_t pix = ok ? ((int64_t)img * H + yy) * W + xx : ((int64_t)img * H + py) * W + px0;
            v[r][c] = reinterpret_cast<const bf16x8s *>(x + pix * cin)[g];
            if (!ok) v[r][c] = bf16x8s{};
        }
    float acc[4][KO];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < KO; ++k) acc[u][k] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        float wt[8][KO];
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int k = 0; k < KO; ++k) wt[e][k] = ws[(tap * cin + g * 8 + e) * KO + k];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xf = (float)v[tap / 3][u + tap % 3][e];
#pragma unroll
                for (int k = 0; k < KO; ++k) acc[u][k] = fmaf(xf, wt[e][k], acc[u][k]);
            }
    }
    for (int off = 1; off < (1 << lg); off <<= 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < KO; ++k) acc[u][k] += __shfl_xor(acc[u][k], off);
    }
    if (live && g == 0) {
        const int64_t hw = (int64_t)H * W;
#pragma unroll
        for (int k = 0; k < KO; ++k) {
            float *o = y + ((int64_t)img * KO + k) * hw + (int64_t)py * W + px0;
            const float b = bias ? bias[k] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (px0 + u < W) o[u] = acc[u][k] + b;
        }
    }
}

// dx[py][px][c] = sum_{ty,tx} sum_k dy[k][py-ty][px-tx] * w[k][c][tap(ty,tx)].  thread = (four consecutive pixels, 8-channel group)
template <int KO>
__global__ __launch_bounds__(256) void smallconv_dgrad_kernel(const float *__restrict__ dy, const float *__restrict__ w, int n_img, int H, int W,
                                                              int cin, int lg, __bf16 *__restrict__ dx) {
    __shared__ float ws[9 * SC_MAX_CIN * KO];   // [tap][k][c]
    for (int i = threadIdx.x; i < 9 * cin * KO; i += 256) {
        const int c = i % cin, k = (i / cin) % KO, tap = i / (KO * cin);
        ws[i] = w[(k * cin + c) * 9 + tap];
    }
    __syncthreads();
    const unsigned qpr = (unsigned)(W + 3) >> 2, n_quads = qpr * H * n_img;
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    const unsigned g = t & ((1u << lg) - 1), q = t >> lg;
    if (q >= n_quads) return;
    const unsigned line = q / qpr;
    const int px0 = (int)(q - line * qpr) * 4;
    const unsigned img = line / (unsigned)H;
    const int py = (int)(line - img * H);
    const int64_t hw = (int64_t)H * W;
    float d[3][6][KO];   // dy[k][py-1+r][px0-1+c]
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const int oy = py + r - 1, ox = px0 + c - 1;
            const bool ok = (unsigned)oy < (unsigned)H && (unsigned)ox < (unsigned)W;
            const int64_t o = ok ? (int64_t)oy * W + ox : (int64_t)py * W + px0;
#pragma unroll
            for (int k = 0; k < KO; ++k) {
                const float val = dy[((int64_t)img * KO + k) * hw + o];
                d[r][c][k] = ok ? val : 0.f;
            }
        }
    float acc[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[u][e] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int ty = tap / 3 - 1, tx = tap % 3 - 1;
#pragma unroll
        for (int k = 0; k < KO; ++k) {
            float wt[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) wt[e] = ws[(tap * KO + k) * cin + g * 8 + e];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float dv = d[1 - ty][u + 1 - tx][k];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[u][e] = fmaf(dv, wt[e], acc[u][e]);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (px0 + u >= W) break;
        bf16x8s o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)acc[u][e];
        reinterpret_cast<bf16x8s *>(dx + (((int64_t)img * H + py) * W + px0 + u) * cin)[g] = o;
    }
}

// block = a run of image rows; thread = worker (kernel row ty, 8-channel group g) x one of 8 pixel lanes (adjacent lanes of a wave).
// A worker reads the input row py+ty once for its three taps: dW[k][c][ty][tx] += dy[k][py][xx-tx] * x[py+ty][xx][c].  Two pixels
// per round; the 8 pixel lanes are folded with shuffles.  partial[block][KO*cin*9 + KO] (weights, then bias)
template <int KO>
__global__ __launch_bounds__(384) void smallconv_wgrad_kernel(const __bf16 *__restrict__ x, const float *__restrict__ dy, int n_img, int H, int W,
                                                              int cin, int lines_per_block, float *__restrict__ partial) {
    constexpr int LN = SC_WG_LANES;
    const int groups = cin >> 3;
    const int wk = threadIdx.x / LN, ln = threadIdx.x % LN;
    const int ty = wk / groups - 1, g = wk % groups;
    const int n_lines = n_img * H;
    const int64_t hw = (int64_t)H * W;
    const int l0 = blockIdx.x * lines_per_block, l1 = l0 + lines_per_block < n_lines ? l0 + lines_per_block : n_lines;
    float acc[3][8][KO], bsum[KO];
#pragma unroll
    for (int k = 0; k < KO; ++k) {
        bsum[k] = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[a][e][k] = 0.f;
    }
    for (int line = l0; line < l1; ++line) {
        const int img = line / H, py = line - img * H, yy = py + ty;
        if ((unsigned)yy >= (unsigned)H) continue;   // uniform per worker; the bias sum lives in the ty == 0 worker (always valid)
        const __bf16 *xrow = x + ((int64_t)img * H + yy) * W * cin + g * 8;
        const float *drow = dy + (int64_t)img * KO * hw + (int64_t)py * W;
        for (int p0 = ln; p0 < W; p0 += 2 * LN) {
            bf16x8s v[2];
            float d[2][3][KO];   // d[u][a][k] = dy[k][py][xx - (a-1)]
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int xx = p0 + u * LN;
                const bool in = xx < W;
                v[u] = *reinterpret_cast<const bf16x8s *>(xrow + (int64_t)(in ? xx : 0) * cin);
                if (!in) v[u] = bf16x8s{};
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const int ox = xx - (a - 1);
                    const bool ok = in && (unsigned)ox < (unsigned)W;
#pragma unroll
                    for (int k = 0; k < KO; ++k) {
                        const float val = drow[(int64_t)k * hw + (ok ? ox : 0)];
                        d[u][a][k] = ok ? val : 0.f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int k = 0; k < KO; ++k) bsum[k] += d[u][1][k];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xf = (float)v[u][e];
#pragma unroll
                    for (int a = 0; a < 3; ++a)
#pragma unroll
                        for (int k = 0; k < KO; ++k) acc[a][e][k] = fmaf(d[u][a][k], xf, acc[a][e][k]);
                }
            }
        }
    }
    for (int off = 1; off < LN; off <<= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int k = 0; k < KO; ++k) acc[a][e][k] += __shfl_xor(acc[a][e][k], off);
#pragma unroll
        for (int k = 0; k < KO; ++k) bsum[k] += __shfl_xor(bsum[k], off);
    }
    const int n_w = KO * cin * 9;
    float *out = partial + (int64_t)blockIdx.x * (n_w + KO);
    if (ln == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int k = 0; k < KO; ++k) out[(k * cin + g * 8 + e) * 9 + (ty + 1) * 3 + a] = acc[a][e][k];
        if (ty == 0 && g == 0) {
#pragma unroll
            for (int k = 0; k < KO; ++k) out[n_w + k] = bsum[k];
        }
    }
}

// 16 elements per block, 16 slices of the partial rows each, folded in a fixed order
__global__ __launch_bounds__(256) void smallconv_wgrad_reduce_kernel(const float *__restrict__ partial, int blocks, int n_w, int ko,
                                                                     float *__restrict__ dw, float *__restrict__ db) {
    __shared__ float fold[16][16];
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4, i = blockIdx.x * 16 + el, row = n_w + ko;
    float s = 0.f;
    if (i < row)
        for (int b = sl; b < blocks; b += 16) s += partial[(int64_t)b * row + i];
    fold[sl][el] = s;
    __syncthreads();
    if (sl == 0 && i < row) {
        float r = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) r += fold[q][el];
        if (i < n_w) dw[i] = r;
        else if (db) db[i - n_w] = r;
    }
}

static bool sc_ok(int cin, int ko) { return cin >= 8 && cin <= SC_MAX_CIN && (cin & (cin - 1)) == 0 && ko >= 1 && ko <= 4; }
static int sc_log2_groups(int cin) {
    int lg = 0;
    while ((8 << lg) < cin) ++lg;
    return lg;
}

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
    const int lg = sc_log2_groups(cin);
    const int64_t total = ((int64_t)n_img * h * ((w + 3) / 4)) << lg;
    S2D_CHECK_ARG(total < (int64_t)1 << 31, "smallconv3x3: map too large");
    const dim3 grid((unsigned)ceil_div(total, 256)), blk(256);
    hipStream_t st = (hipStream_t)stream;
#define S2D_SC_FWD(K) hipLaunchKernelGGL(smallconv_fwd_kernel<K>, grid, blk, 0, st, (const __bf16 *)x, weight, bias, n_img, h, w, cin, lg, y)
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
    const int lg = sc_log2_groups(cin);
    const int64_t total = ((int64_t)n_img * h * ((w + 3) / 4)) << lg;
    S2D_CHECK_ARG(total < (int64_t)1 << 31, "smallconv3x3: map too large");
    const dim3 grid((unsigned)ceil_div(total, 256)), blk(256);
    hipStream_t st = (hipStream_t)stream;
#define S2D_SC_DG(K) hipLaunchKernelGGL(smallconv_dgrad_kernel<K>, grid, blk, 0, st, dy, weight, n_img, h, w, cin, lg, (__bf16 *)dx)
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
    const int n_lines = n_img * h;
    const int lpb = (int)ceil_div(n_lines, SC_WG_BLOCKS), blocks = (int)ceil_div(n_lines, lpb);
    const int threads = 3 * (cin / 8) * SC_WG_LANES;
    float *partial = (float *)ws;
    hipStream_t st = (hipStream_t)stream;
#define S2D_SC_WG(K) \
    hipLaunchKernelGGL(smallconv_wgrad_kernel<K>, dim3(blocks), dim3(threads), 0, st, (const __bf16 *)x, dy, n_img, h, w, cin, lpb, partial)
    switch (cout) {
        case 1: S2D_SC_WG(1); break;
        case 2: S2D_SC_WG(2); break;
        case 3: S2D_SC_WG(3); break;
        default: S2D_SC_WG(4); break;
    }
#undef S2D_SC_WG
    const int n_w = cout * cin * 9;
    hipLaunchKernelGGL(smallconv_wgrad_reduce_kernel, dim3((unsigned)ceil_div(n_w + cout, 16)), dim3(256), 0, st, partial, blocks, n_w, cout, dweight,
                       dbias);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
