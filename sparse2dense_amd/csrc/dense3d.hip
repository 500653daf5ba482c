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
// Weight gradients are plain GEMMs over the position axis and go through hipBLASLt (torch.matmul).
#include "s2d_common.h"

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
