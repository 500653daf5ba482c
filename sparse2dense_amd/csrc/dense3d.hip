// PCR (point-cloud reconstruction) head of the S2D neck on gfx950: the 1x1x1 Conv3d layers and the
// two ConvTranspose3d(k=4, s=2, p=1) up-samplers of /root/reference/det3d/models/necks/rpn.py:263-296
// (and the 1x1x1 PCR heads of the pillar S2D backbone, readers/pillar_encoder.py:304-323).
// They move ~1 GB of fp32 activations per frame through channel counts of 1..32 (128 once): pure
// HBM-bound streaming, which MIOpen serves with im2col/naive kernels (6-7 ms per layer backward, and
// seconds of `naive_conv` search on a fresh box).  Layout: NCDHW fp32, contiguous, P = D*H*W.
//
//   pw_conv_kernel<CO>        out[n][co][p] = b[co] + sum_ci W[co][ci] * in[n][ci][p]
//                             thread = 4 consecutive positions (float4) x CO output channels; the
//                             weights are wave-uniform (scalar loads).  Serves forward and, with the
//                             transposed weight, the data gradient.  bytes = 4*P*(Cin + Cout) per sample.
//   convt3d_fwd_kernel<CO>    thread = one input cell h and one (pz,py) output-parity class, both px
//                             parities -> float2 stores; taps are the input cells h, h+-1; weights
//                             wave-uniform.
//   convt3d_dgrad_kernel<CI>  din[ci][h] = sum_co sum_{k in 4^3} dout[co][2h-1+k] * W[ci][co][k]
//                             thread = one input cell, float4 loads of the 4 x-taps.
//   convt3d_wgrad_kernel      LDS-staged rows contracted on the fp32 matrix cores (see below).
// The 1x1x1 weight gradients are plain GEMMs over the position axis (hipBLASLt via torch.matmul).
#include "s2d_common.h"
#include <cstdlib>

namespace s2d {

template <int CO>
__global__ __launch_bounds__(256) void pw_conv_kernel(const float *__restrict__ in, const float *__restrict__ w,
                                                      const float *__restrict__ bias, int64_t p4, int cin, int cout,
                                                      float *__restrict__ out) {
    // grid: x = position quads, y = output-channel tile, z = sample
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= p4) return;
    const int co0 = blockIdx.y * CO;
    const int n = blockIdx.z;
    const float4 *src = reinterpret_cast<const float4 *>(in) + (int64_t)n * cin * p4 + q;
    float4 acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) {
        const float b = (bias && co0 + c < cout) ? bias[co0 + c] : 0.f;
        acc[c] = float4{b, b, b, b};
    }
    for (int ci = 0; ci < cin; ++ci) {
        const float4 x = src[(int64_t)ci * p4];
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            const float wv = (co0 + c < cout) ? w[(int64_t)(co0 + c) * cin + ci] : 0.f;  // wave-uniform -> s_load
            acc[c].x = fmaf(wv, x.x, acc[c].x);
            acc[c].y = fmaf(wv, x.y, acc[c].y);
            acc[c].z = fmaf(wv, x.z, acc[c].z);
            acc[c].w = fmaf(wv, x.w, acc[c].w);
        }
    }
    float4 *dst = reinterpret_cast<float4 *>(out) + (int64_t)n * cout * p4 + q;
#pragma unroll
    for (int c = 0; c < CO; ++c)
        if (co0 + c < cout) dst[(int64_t)(co0 + c) * p4] = acc[c];
}

// 32 output channels per thread on position PAIRS (float2): one pass over the input for the PCR head's 128 -> 32 conv, four instead of
// eight for its 32 -> 128 data gradient, at 64 accumulator registers (the float4 variant with 32 channels drops to one wave per SIMD)
__global__ __launch_bounds__(256) void pw_conv32_kernel(const float *__restrict__ in, const float *__restrict__ w,
                                                        const float *__restrict__ bias, int64_t p2, int cin, int cout,
                                                        float *__restrict__ out) {
    constexpr int CO = 32;
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= p2) return;
    const int co0 = blockIdx.y * CO;
    const int n = blockIdx.z;
    const float2 *src = reinterpret_cast<const float2 *>(in) + (int64_t)n * cin * p2 + q;
    float2 acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) {
        const float b = bias ? bias[co0 + c] : 0.f;
        acc[c] = float2{b, b};
    }
#pragma unroll 4
    for (int ci = 0; ci < cin; ++ci) {
        const float2 x = src[(int64_t)ci * p2];
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            const float wv = w[(int64_t)(co0 + c) * cin + ci];  // wave-uniform -> s_load
            acc[c].x = fmaf(wv, x.x, acc[c].x);
            acc[c].y = fmaf(wv, x.y, acc[c].y);
        }
    }
    float2 *dst = reinterpret_cast<float2 *>(out) + (int64_t)n * cout * p2 + q;
#pragma unroll
    for (int c = 0; c < CO; ++c) dst[(int64_t)(co0 + c) * p2] = acc[c];
}

// scalar-position variant for P not divisible by 4
__global__ __launch_bounds__(256) void pw_conv_tail_kernel(const float *__restrict__ in, const float *__restrict__ w,
                                                           const float *__restrict__ bias, int64_t p, int cin, int cout,
                                                           float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.z;
    if (i >= p) return;
    for (int co = 0; co < cout; ++co) {
        float acc = bias ? bias[co] : 0.f;
        for (int ci = 0; ci < cin; ++ci) acc = fmaf(w[(int64_t)co * cin + ci], in[((int64_t)n * cin + ci) * p + i], acc);
        out[((int64_t)n * cout + co) * p + i] = acc;
    }
}

// ---- ConvTranspose3d k=4 s=2 p=1 ------------------------------------------------------------------
// torch weight layout [cin][cout][4][4][4].  Output o = 2h + par; taps per axis:
//   par 0: (k=1, i=h), (k=3, i=h-1)      par 1: (k=0, i=h+1), (k=2, i=h)
struct Dims3 {
    int d, h, w;  // INPUT extents; output is 2d x 2h x 2w
};

template <int CO>
__global__ __launch_bounds__(256) void convt3d_fwd_kernel(const float *__restrict__ in, const float *__restrict__ w,
                                                          const float *__restrict__ bias, Dims3 s, int cin, int cout,
                                                          float *__restrict__ out) {
    // grid: x = input cells (x fastest), y = (pz,py) class * co tiles, z = sample
    const int64_t cells = (int64_t)s.d * s.h * s.w;
    const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= cells) return;
    const int cls = blockIdx.y & 3, co0 = (blockIdx.y >> 2) * CO;
    const int pz = cls >> 1, py = cls & 1;
    const int n = blockIdx.z;
    const int hx = (int)(cell % s.w);
    const int hy = (int)((cell / s.w) % s.h);
    const int hz = (int)(cell / ((int64_t)s.w * s.h));
    // per-axis taps (kernel index, input offset)
    const int kz[2] = {pz ? 0 : 1, pz ? 2 : 3}, dz[2] = {pz ? 1 : 0, pz ? 0 : -1};
    const int ky[2] = {py ? 0 : 1, py ? 2 : 3}, dy[2] = {py ? 1 : 0, py ? 0 : -1};
    float acc0[CO], acc1[CO];  // px = 0 / 1
#pragma unroll
    for (int c = 0; c < CO; ++c) {
        const float b = (bias && co0 + c < cout) ? bias[co0 + c] : 0.f;
        acc0[c] = b;
        acc1[c] = b;
    }
    const float *inb = in + (int64_t)n * cin * cells;
    for (int ci = 0; ci < cin; ++ci) {
        const float *plane = inb + (int64_t)ci * cells;
        float x[2][2][3];  // [z tap][y tap][x = hx-1, hx, hx+1]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int z = hz + dz[a], y = hy + dy[b];
                const bool ok = (unsigned)z < (unsigned)s.d && (unsigned)y < (unsigned)s.h;
                const float *row = plane + ((int64_t)(ok ? z : 0) * s.h + (ok ? y : 0)) * s.w;
                x[a][b][0] = (ok && hx > 0) ? row[hx - 1] : 0.f;
                x[a][b][1] = ok ? row[hx] : 0.f;
                x[a][b][2] = (ok && hx + 1 < s.w) ? row[hx + 1] : 0.f;
            }
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            if (co0 + c < cout) {
                const float *wk = w + ((int64_t)ci * cout + co0 + c) * 64;  // wave-uniform
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const float *w4 = wk + (kz[a] * 4 + ky[b]) * 4;  // the 4 kx entries
                        // px=0: (kx=1, hx), (kx=3, hx-1) ; px=1: (kx=0, hx+1), (kx=2, hx)
                        acc0[c] = fmaf(w4[1], x[a][b][1], acc0[c]);
                        acc0[c] = fmaf(w4[3], x[a][b][0], acc0[c]);
                        acc1[c] = fmaf(w4[0], x[a][b][2], acc1[c]);
                        acc1[c] = fmaf(w4[2], x[a][b][1], acc1[c]);
                    }
            }
        }
    }
    const int od = 2 * s.d, oh = 2 * s.h, ow = 2 * s.w;
    const int oz = 2 * hz + pz, oy = 2 * hy + py;
    float *ob = out + (int64_t)n * cout * od * oh * ow;
#pragma unroll
    for (int c = 0; c < CO; ++c)
        if (co0 + c < cout)
            *reinterpret_cast<float2 *>(ob + (((int64_t)(co0 + c) * od + oz) * oh + oy) * ow + 2 * hx) = float2{acc0[c], acc1[c]};
}

template <int CI>
__global__ __launch_bounds__(256) void convt3d_dgrad_kernel(const float *__restrict__ dout, const float *__restrict__ w,
                                                            Dims3 s, int cin, int cout, float *__restrict__ din) {
    // grid: x = input cells, y = ci tiles, z = sample.  din[ci][h] = sum_co sum_k dout[co][2h-1+k] W[ci][co][k]
    const int64_t cells = (int64_t)s.d * s.h * s.w;
    const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= cells) return;
    const int ci0 = blockIdx.y * CI;
    const int n = blockIdx.z;
    const int hx = (int)(cell % s.w);
    const int hy = (int)((cell / s.w) % s.h);
    const int hz = (int)(cell / ((int64_t)s.w * s.h));
    const int od = 2 * s.d, oh = 2 * s.h, ow = 2 * s.w;
    float acc[CI];
#pragma unroll
    for (int c = 0; c < CI; ++c) acc[c] = 0.f;
    const float *db = dout + (int64_t)n * cout * od * oh * ow;
    const int x0 = 2 * hx - 1;  // taps x0 .. x0+3 ; x0+1, x0+2 always valid
    for (int co = 0; co < cout; ++co) {
        const float *plane = db + (int64_t)co * od * oh * ow;
        for (int kzz = 0; kzz < 4; ++kzz) {
            const int z = 2 * hz - 1 + kzz;
            if ((unsigned)z >= (unsigned)od) continue;
            for (int kyy = 0; kyy < 4; ++kyy) {
                const int y = 2 * hy - 1 + kyy;
                if ((unsigned)y >= (unsigned)oh) continue;
                const float *row = plane + ((int64_t)z * oh + y) * ow;
                const float2 mid = *reinterpret_cast<const float2 *>(row + x0 + 1);
                const float g0 = x0 >= 0 ? row[x0] : 0.f;
                const float g3 = x0 + 3 < ow ? row[x0 + 3] : 0.f;
#pragma unroll
                for (int c = 0; c < CI; ++c) {
                    if (ci0 + c < cin) {
                        const float *w4 = w + (((int64_t)(ci0 + c) * cout + co) * 16 + kzz * 4 + kyy) * 4;  // uniform
                        acc[c] = fmaf(w4[0], g0, acc[c]);
                        acc[c] = fmaf(w4[1], mid.x, acc[c]);
                        acc[c] = fmaf(w4[2], mid.y, acc[c]);
                        acc[c] = fmaf(w4[3], g3, acc[c]);
                    }
                }
            }
        }
    }
    float *dst = din + (int64_t)n * cin * cells + cell;
#pragma unroll
    for (int c = 0; c < CI; ++c)
        if (ci0 + c < cin) dst[(int64_t)(ci0 + c) * cells] = acc[c];
}

// ---- ConvTranspose3d k4s2p1 weight gradient ------------------------------------------------------
// dW[ci][co][kz][ky][kx] = sum_{n,h} x[n][ci][h] * dout[n][co][2h-1+k].  grid (row chunks, 16 (kz,ky) pairs);
// a block walks input rows (n, hz, hy), stages x[ci][0..W) and dout[co][2hz-1+kz][2hy-1+ky][-1..2W]
// in LDS with coalesced row loads, and contracts over the row's cells on the fp32 matrix cores
// (A[i=ci][k=cell], B[k=cell][j=co] at stride 2, one accumulator set per kx).  Waves split the
// cells; partial[chunk][ci][co][kz][ky][kx] is reduced by wgrad_reduce (fixed order).
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int CIT, int COT>  // number of 16-wide ci / co tiles (channels are zero padded in LDS)
__global__ __launch_bounds__(256) void convt3d_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dout,
                                                            Dims3 s, int batch, int cin, int cout, int rows_per_block,
                                                            float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int W = s.w, W2 = 2 * s.w + 2;           // dout tile holds positions -1 .. 2W
    const int xs_stride = W + 1, ds_stride = W2 + 1;   // +1: break power-of-two row strides
    float *xs = lds;                                 // [CIT*16][xs_stride]
    float *ds = lds + CIT * 16 * xs_stride;          // [COT*16][ds_stride]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int i16 = lane & 15, q = lane >> 4;
    const int kz = blockIdx.y >> 2, ky = blockIdx.y & 3;
    const int od = 2 * s.d, oh = 2 * s.h, ow = 2 * s.w;
    const int64_t total_rows = (int64_t)batch * s.d * s.h;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < total_rows ? r0 + rows_per_block : total_rows;

    f32x4_t acc[CIT][COT][4];
#pragma unroll
    for (int a = 0; a < CIT; ++a)
#pragma unroll
        for (int b = 0; b < COT; ++b)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[a][b][k] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int64_t row = r0; row < r1; ++row) {
        const int hy = (int)(row % s.h);
        const int hz = (int)((row / s.h) % s.d);
        const int n = (int)(row / ((int64_t)s.h * s.d));
        const int z = 2 * hz - 1 + kz, y = 2 * hy - 1 + ky;
        if ((unsigned)z >= (unsigned)od || (unsigned)y >= (unsigned)oh) continue;   // block-uniform
        __syncthreads();   // previous row fully consumed
        for (int e = threadIdx.x; e < CIT * 16 * W; e += 256) {
            const int ci = e / W, c = e - ci * W;
            xs[ci * xs_stride + c] = ci < cin ? x[(((int64_t)n * cin + ci) * s.d + hz) * s.h * (int64_t)W + (int64_t)hy * W + c] : 0.f;
        }
        for (int e = threadIdx.x; e < COT * 16 * W2; e += 256) {
            const int co = e / W2, p = e - co * W2;       // p = position + 1
            const int pos = p - 1;
            float v = 0.f;
            if (co < cout && (unsigned)pos < (unsigned)ow)
                v = dout[((((int64_t)n * cout + co) * od + z) * oh + y) * (int64_t)ow + pos];
            ds[co * ds_stride + p] = v;
        }
        __syncthreads();
        for (int c0 = 4 * wid; c0 < W; c0 += 16) {   // 4 cells per MFMA k-step, waves interleaved
            const int cell = c0 + q;
            float av[CIT], bv[COT][4];
#pragma unroll
            for (int a = 0; a < CIT; ++a) av[a] = cell < W ? xs[(a * 16 + i16) * xs_stride + cell] : 0.f;
#pragma unroll
            for (int b = 0; b < COT; ++b)
#pragma unroll
                for (int k = 0; k < 4; ++k) bv[b][k] = cell < W ? ds[(b * 16 + i16) * ds_stride + 2 * cell + k] : 0.f;   // pos+1 = 2c-1+k+1
#pragma unroll
            for (int a = 0; a < CIT; ++a)
#pragma unroll
                for (int b = 0; b < COT; ++b)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        acc[a][b][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b][k], acc[a][b][k], 0, 0, 0);
        }
    }
    // cross-wave reduction through LDS, then one partial slab per block
    __syncthreads();
    float *red = lds;   // [4 waves][CIT*COT*4 tiles][256]
    constexpr int TILES = CIT * COT * 4;
#pragma unroll
    for (int a = 0; a < CIT; ++a)
#pragma unroll
        for (int b = 0; b < COT; ++b)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    red[((wid * TILES + (a * COT + b) * 4 + k) * 4 + reg) * 64 + lane] = acc[a][b][k][reg];
    __syncthreads();
    float *dst = partial + (int64_t)blockIdx.x * cin * cout * 64;
    for (int e = threadIdx.x; e < TILES * 256; e += 256) {
        const int t = e / 256, rl = e % 256, reg = rl / 64, ln = rl % 64;
        const float v = (red[((0 * TILES + t) * 4 + reg) * 64 + ln] + red[((1 * TILES + t) * 4 + reg) * 64 + ln]) +
                        (red[((2 * TILES + t) * 4 + reg) * 64 + ln] + red[((3 * TILES + t) * 4 + reg) * 64 + ln]);
        const int k = t % 4, b = (t / 4) % COT, a = t / (4 * COT);
        const int ci = a * 16 + 4 * (ln >> 4) + reg, co = b * 16 + (ln & 15);   // C/D layout: row = 4*(lane>>4)+reg, col = lane&15
        if (ci < cin && co < cout) dst[(((int64_t)ci * cout + co) * 4 + kz) * 16 + ky * 4 + k] = v;
    }
}

__global__ __launch_bounds__(256) void slab_reduce_kernel(const float *__restrict__ partial, int n_slabs, int64_t size,
                                                          float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= size) return;
    float sum = 0.f;
    for (int sidx = 0; sidx < n_slabs; ++sidx) sum += partial[(int64_t)sidx * size + i];
    out[i] = sum;
}

struct CtWgradPlan {
    int blocks_x, rows_per_block, cit, cot;
    size_t lds, ws_bytes;
};
static CtWgradPlan ct_wgrad_plan(int batch, int cin, int cout, int d, int h, int w) {
    CtWgradPlan p;
    p.cit = (cin + 15) / 16;
    p.cot = (cout + 15) / 16;
    const int64_t rows = (int64_t)batch * d * h;
    int64_t bx = rows < 128 ? rows : 128;
    p.blocks_x = (int)bx;
    p.rows_per_block = (int)ceil_div(rows, bx);
    const size_t stage = ((size_t)p.cit * 16 * (w + 1) + (size_t)p.cot * 16 * (2 * w + 3)) * sizeof(float);
    const size_t red = (size_t)4 * p.cit * p.cot * 4 * 256 * sizeof(float);
    p.lds = stage > red ? stage : red;
    p.ws_bytes = align_up((size_t)p.blocks_x * cin * cout * 64 * sizeof(float), 256);
    return p;
}

static int pick_tile(int c) { return c >= 16 ? 16 : (c >= 8 ? 8 : (c >= 4 ? 4 : (c == 3 ? 3 : (c == 2 ? 2 : 1)))); }

}  // namespace s2d

using namespace s2d;

extern "C" int s2d_pointwise_conv_f32(const float *in, const float *weight, const float *bias, int batch, int cin, int cout,
                                      int64_t positions, float *out, s2d_stream_t stream) {
    S2D_CHECK_ARG(in && weight && out && batch > 0 && cin > 0 && cout > 0 && positions > 0, "pointwise_conv: bad argument");
    S2D_CHECK_ARG(batch <= 65535, "pointwise_conv: batch too large");
    hipStream_t st = (hipStream_t)stream;
    if (positions % 4) {
        hipLaunchKernelGGL(pw_conv_tail_kernel, dim3((unsigned)ceil_div(positions, 256), 1, batch), dim3(256), 0, st, in,
                           weight, bias, positions, cin, cout, out);
        S2D_LAUNCH_CHECK();
        return S2D_OK;
    }
    const int64_t p4 = positions / 4;
    static const bool use32 = [] { const char *e = getenv("S2D_PW32"); return !(e && e[0] == '0'); }();
    if (use32 && cout % 32 == 0 && cin >= 32) {   // (32 channels on float4 positions: 256 VGPRs, one wave per SIMD, 466 us vs 191 us)
        const int64_t p2 = positions / 2;
        hipLaunchKernelGGL(pw_conv32_kernel, dim3((unsigned)ceil_div(p2, 256), (unsigned)(cout / 32), batch), dim3(256), 0, st, in, weight, bias,
                           p2, cin, cout, out);
        S2D_LAUNCH_CHECK();
        return S2D_OK;
    }
    const int t = pick_tile(cout);
    const dim3 grid((unsigned)ceil_div(p4, 256), (unsigned)ceil_div(cout, t), batch), blk(256);
#define S2D_PW(T) hipLaunchKernelGGL(pw_conv_kernel<T>, grid, blk, 0, st, in, weight, bias, p4, cin, cout, out)
    switch (t) {
        case 16: S2D_PW(16); break;
        case 8: S2D_PW(8); break;
        case 4: S2D_PW(4); break;
        case 3: S2D_PW(3); break;
        case 2: S2D_PW(2); break;
        default: S2D_PW(1); break;
    }
#undef S2D_PW
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_convt3d_k4s2p1_fwd_f32(const float *in, const float *weight, const float *bias, int batch, int cin,
                                          int cout, int d, int h, int w, float *out, s2d_stream_t stream) {
    S2D_CHECK_ARG(in && weight && out && batch > 0 && batch <= 65535 && cin > 0 && cout > 0 && d > 0 && h > 0 && w > 0,
                  "convt3d_fwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    Dims3 s{d, h, w};
    const int64_t cells = (int64_t)d * h * w;
    const int t = pick_tile(cout);
    const dim3 grid((unsigned)ceil_div(cells, 256), (unsigned)(4 * ceil_div(cout, t)), batch), blk(256);
#define S2D_CT(T) hipLaunchKernelGGL(convt3d_fwd_kernel<T>, grid, blk, 0, st, in, weight, bias, s, cin, cout, out)
    switch (t) {
        case 16: S2D_CT(16); break;
        case 8: S2D_CT(8); break;
        case 4: S2D_CT(4); break;
        case 3: S2D_CT(3); break;
        case 2: S2D_CT(2); break;
        default: S2D_CT(1); break;
    }
#undef S2D_CT
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_convt3d_k4s2p1_dgrad_f32(const float *dout, const float *weight, int batch, int cin, int cout, int d,
                                            int h, int w, float *din, s2d_stream_t stream) {
    S2D_CHECK_ARG(dout && weight && din && batch > 0 && batch <= 65535 && cin > 0 && cout > 0 && d > 0 && h > 0 && w > 0,
                  "convt3d_dgrad: bad argument");
    hipStream_t st = (hipStream_t)stream;
    Dims3 s{d, h, w};
    const int64_t cells = (int64_t)d * h * w;
    const int t = pick_tile(cin);
    const dim3 grid((unsigned)ceil_div(cells, 256), (unsigned)ceil_div(cin, t), batch), blk(256);
#define S2D_CD(T) hipLaunchKernelGGL(convt3d_dgrad_kernel<T>, grid, blk, 0, st, dout, weight, s, cin, cout, din)
    switch (t) {
        case 16: S2D_CD(16); break;
        case 8: S2D_CD(8); break;
        case 4: S2D_CD(4); break;
        case 3: S2D_CD(3); break;
        case 2: S2D_CD(2); break;
        default: S2D_CD(1); break;
    }
#undef S2D_CD
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" size_t s2d_convt3d_k4s2p1_wgrad_workspace_bytes(int batch, int cin, int cout, int d, int h, int w) {
    if (batch <= 0 || cin <= 0 || cout <= 0 || d <= 0 || h <= 0 || w <= 0) return 0;
    return ct_wgrad_plan(batch, cin, cout, d, h, w).ws_bytes;
}

extern "C" int s2d_convt3d_k4s2p1_wgrad_f32(const float *in, const float *dout, int batch, int cin, int cout, int d, int h,
                                            int w, float *dweight, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(in && dout && dweight && batch > 0 && cin > 0 && cout > 0 && d > 0 && h > 0 && w > 0, "convt3d_wgrad: bad argument");
    if (cin > 32 || cout > 32) {
        set_error("convt3d_wgrad: channel counts above 32 unsupported (%d -> %d)", cin, cout);
        return S2D_ERR_UNSUPPORTED;
    }
    CtWgradPlan p = ct_wgrad_plan(batch, cin, cout, d, h, w);
    if (p.lds > 160 * 1024) {
        set_error("convt3d_wgrad: row of %d cells does not fit the LDS staging (%zu bytes)", w, p.lds);
        return S2D_ERR_UNSUPPORTED;
    }
    if (!ws || ws_bytes < p.ws_bytes) {
        set_error("convt3d_wgrad: workspace too small (%zu < %zu)", ws_bytes, p.ws_bytes);
        return S2D_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    Dims3 s{d, h, w};
    float *partial = (float *)ws;
    if (int rc = zero_async(partial, p.ws_bytes, st)) return rc;   // (kz,ky) blocks whose rows are all out of range write nothing
    const dim3 grid(p.blocks_x, 16), blk(256);
#define S2D_CW(A, B)                                                                                          \
    do {                                                                                                      \
        auto kern = convt3d_wgrad_kernel<A, B>;                                                               \
        S2D_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds)); \
        hipLaunchKernelGGL(kern, grid, blk, p.lds, st, in, dout, s, batch, cin, cout, p.rows_per_block, partial); \
    } while (0)
    if (p.cit == 1 && p.cot == 1) S2D_CW(1, 1);
    else if (p.cit == 1 && p.cot == 2) S2D_CW(1, 2);
    else if (p.cit == 2 && p.cot == 1) S2D_CW(2, 1);
    else S2D_CW(2, 2);
#undef S2D_CW
    const int64_t size = (int64_t)cin * cout * 64;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)ceil_div(size, 256)), dim3(256), 0, st, partial, p.blocks_x, size, dweight);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
