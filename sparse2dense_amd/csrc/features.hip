// Row-wise feature kernels of the sparse backbone on gfx950 (all HBM-bound, float4 vectorised):
//   * BatchNorm1d statistics / apply(+ReLU,+residual) / backward on `.features` [N,C]
//     (det3d/models/backbones/scn.py:69-85,104-152; BN1d eps=1e-3 momentum=0.01, scn.py:100-101)
//   * SparseConvTensor.dense() scatter and its gather backward (scn.py:173-176)
// Reductions are two-pass and order-fixed (deterministic; SyncBN all-reduces the [2C] vectors).
#include "s2d_common.h"
#include <cstdlib>

namespace s2d {

constexpr int RED_THREADS = 256;

// Each block reduces rows [blockIdx.x*rows_per_block, ...) for all channels; thread t owns the
// 4-channel group (t % (C/4)) and the row lane (t / (C/4)).  C must be a multiple of 4, <= 1024.
template <bool BWD>
__global__ __launch_bounds__(RED_THREADS) void col_reduce_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                                 const float *__restrict__ y, int relu, int64_t n, int c,
                                                                 int rows_per_block, float *__restrict__ g_out,
                                                                 float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [row_lanes][2][C]
    const int c4 = c >> 2;
    const int lanes = RED_THREADS / c4;  // row lanes per block (>= 1)
    const int grp = threadIdx.x % c4, rl = threadIdx.x / c4;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    float4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
    if (rl < lanes) {
        for (int64_t r = r0 + rl; r < r1; r += lanes) {
            const float4 xv = reinterpret_cast<const float4 *>(x + r * c)[grp];
            if (BWD) {
                float4 g = reinterpret_cast<const float4 *>(dy + r * c)[grp];
                if (relu) {
                    const float4 yv = reinterpret_cast<const float4 *>(y + r * c)[grp];
                    g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
                    g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
                }
                if (g_out) reinterpret_cast<float4 *>(g_out + r * c)[grp] = g;
                s0.x += g.x; s0.y += g.y; s0.z += g.z; s0.w += g.w;
                s1.x += g.x * xv.x; s1.y += g.y * xv.y; s1.z += g.z * xv.z; s1.w += g.w * xv.w;
            } else {
                s0.x += xv.x; s0.y += xv.y; s0.z += xv.z; s0.w += xv.w;
                s1.x += xv.x * xv.x; s1.y += xv.y * xv.y; s1.z += xv.z * xv.z; s1.w += xv.w * xv.w;
            }
        }
        reinterpret_cast<float4 *>(lds + (size_t)rl * 2 * c)[grp] = s0;
        reinterpret_cast<float4 *>(lds + (size_t)rl * 2 * c + c)[grp] = s1;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * c; e += RED_THREADS) {
        float s = 0.f;
        for (int l = 0; l < lanes; ++l) s += lds[(size_t)l * 2 * c + e];
        partial[(size_t)blockIdx.x * 2 * c + e] = s;
    }
}

// one wave per output element: lanes stride over the per-block partials, fixed-order butterfly
// out2 (optional): a second copy of the sums (the one that gets all-reduced); count >= 0: also written to out[width]
__global__ __launch_bounds__(256) void partial_sum_kernel(const float *__restrict__ partial, int nblocks, int width,
                                                          float *__restrict__ out, float *__restrict__ out2 = nullptr,
                                                          float count = -1.f, bool cm = false) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (e == 0 && lane == 0 && count >= 0.f) out[width] = count;
    if (e >= width) return;
    float s = 0.f;
    // cm: partials stored element-major [width][nblocks] (the row-major bf16 reduce kernels): the lanes' loads are then
    // one contiguous run instead of 64 sectors
    for (int b = lane; b < nblocks; b += 64) s += cm ? partial[(size_t)e * nblocks + b] : partial[(size_t)b * width + e];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) {
        out[e] = s;
        if (out2) out2[e] = s;
    }
}

// first stage for very long partial lists (a producer with one partial row per tile: 10^5..10^6 rows): block = a contiguous run of
// rows, thread = (row lane, element) so that every load of the block is one contiguous run; fixed-order LDS fold.  out[slice][width]
__global__ __launch_bounds__(256) void partial_sum_slices_kernel(const float *__restrict__ partial, int nblocks, int width, int rows_per_slice,
                                                                 float *__restrict__ out) {
    __shared__ float fold[256];
    const int lanes = 256 / width, e = threadIdx.x % width, r = threadIdx.x / width;
    const int64_t b0 = (int64_t)blockIdx.x * rows_per_slice, b1 = b0 + rows_per_slice < nblocks ? b0 + rows_per_slice : nblocks;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (r < lanes) {
        int64_t b = b0 + r;
        for (; b + 3 * lanes < b1; b += 4 * lanes) {
            s0 += partial[b * width + e];
            s1 += partial[(b + lanes) * width + e];
            s2 += partial[(b + 2 * lanes) * width + e];
            s3 += partial[(b + 3 * lanes) * width + e];
        }
        for (; b < b1; b += lanes) s0 += partial[b * width + e];
    }
    fold[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (threadIdx.x < width) {
        float s = 0.f;
        for (int l = 0; l < lanes; ++l) s += fold[l * width + threadIdx.x];
        out[(int64_t)blockIdx.x * width + threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float *__restrict__ x, const float *__restrict__ scale,
                                                       const float *__restrict__ shift, const float *__restrict__ res,
                                                       int relu, int64_t n4, int c4, float *__restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % c4);
        float4 v = reinterpret_cast<const float4 *>(x)[i];
        const float4 a = reinterpret_cast<const float4 *>(scale)[g];
        const float4 b = reinterpret_cast<const float4 *>(shift)[g];
        v.x = fmaf(v.x, a.x, b.x); v.y = fmaf(v.y, a.y, b.y); v.z = fmaf(v.z, a.z, b.z); v.w = fmaf(v.w, a.w, b.w);
        if (res) {
            const float4 rr = reinterpret_cast<const float4 *>(res)[i];
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        if (relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        reinterpret_cast<float4 *>(y)[i] = v;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float *__restrict__ g, const float *__restrict__ x,
                                                           const float *__restrict__ a, const float *__restrict__ b,
                                                           const float *__restrict__ d, int64_t n4, int c4,
                                                           float *__restrict__ dx) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int grp = (int)(i % c4);
        const float4 gv = reinterpret_cast<const float4 *>(g)[i];
        const float4 xv = reinterpret_cast<const float4 *>(x)[i];
        const float4 av = reinterpret_cast<const float4 *>(a)[grp];
        const float4 bv = reinterpret_cast<const float4 *>(b)[grp];
        const float4 dv = reinterpret_cast<const float4 *>(d)[grp];
        float4 o;
        o.x = fmaf(av.x, gv.x, fmaf(bv.x, xv.x, dv.x));
        o.y = fmaf(av.y, gv.y, fmaf(bv.y, xv.y, dv.y));
        o.z = fmaf(av.z, gv.z, fmaf(bv.z, xv.z, dv.z));
        o.w = fmaf(av.w, gv.w, fmaf(bv.w, xv.w, dv.w));
        reinterpret_cast<float4 *>(dx)[i] = o;
    }
}

// dense(): one thread per (channel, row) with the row index fastest, so that x-adjacent sites of
// the canonically ordered rows write adjacent addresses of the same channel plane.
__global__ __launch_bounds__(256) void densify_fwd_kernel(const float *__restrict__ feat, const int32_t *__restrict__ coors,
                                                          int64_t n, int batch, int D, int H, int W, int c,
                                                          float *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ch = t / n;
    const int64_t i = t - ch * n;
    if (ch >= c) return;
    const int4 co = reinterpret_cast<const int4 *>(coors)[i];
    if ((unsigned)co.x >= (unsigned)batch || (unsigned)co.y >= (unsigned)D || (unsigned)co.z >= (unsigned)H ||
        (unsigned)co.w >= (unsigned)W)
        return;
    out[((((int64_t)co.x * c + ch) * D + co.y) * H + co.z) * W + co.w] = feat[i * c + ch];
}

__global__ __launch_bounds__(256) void densify_bwd_kernel(const float *__restrict__ dout, const int32_t *__restrict__ coors,
                                                          int64_t n, int batch, int D, int H, int W, int c,
                                                          float *__restrict__ dfeat) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (row, channel), channel fastest
    const int64_t i = t / c;
    const int ch = (int)(t - i * c);
    if (i >= n) return;
    const int4 co = reinterpret_cast<const int4 *>(coors)[i];
    float v = 0.f;
    if ((unsigned)co.x < (unsigned)batch && (unsigned)co.y < (unsigned)D && (unsigned)co.z < (unsigned)H &&
        (unsigned)co.w < (unsigned)W)
        v = dout[((((int64_t)co.x * c + ch) * D + co.y) * H + co.z) * W + co.w];
    dfeat[t] = v;
}


// dense() straight into the BEV layout the bf16 neck consumes: [B][H][W][c*D] bf16 (channels_last view of
// ret.view(N, C*D, H, W), scn.py:173-176); BEV channel = ch*D + z.  Thread per (row, channel), channel fastest.
template <typename T>
__global__ __launch_bounds__(256) void densify_bev_fwd_kernel(const T *__restrict__ feat, const int32_t *__restrict__ coors,
                                                              int64_t n, int batch, int D, int H, int W, int c,
                                                              __bf16 *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t / c;
    const int ch = (int)(t - i * c);
    if (i >= n) return;
    const int4 co = reinterpret_cast<const int4 *>(coors)[i];
    if ((unsigned)co.x >= (unsigned)batch || (unsigned)co.y >= (unsigned)D || (unsigned)co.z >= (unsigned)H ||
        (unsigned)co.w >= (unsigned)W)
        return;
    out[((((int64_t)co.x * H + co.z) * W + co.w) * c + ch) * D + co.y] = (__bf16)(float)feat[t];
}

template <typename T>
__global__ __launch_bounds__(256) void densify_bev_bwd_kernel(const __bf16 *__restrict__ dout, const int32_t *__restrict__ coors,
                                                              int64_t n, int batch, int D, int H, int W, int c,
                                                              T *__restrict__ dfeat) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = t / c;
    const int ch = (int)(t - i * c);
    if (i >= n) return;
    const int4 co = reinterpret_cast<const int4 *>(coors)[i];
    float v = 0.f;
    if ((unsigned)co.x < (unsigned)batch && (unsigned)co.y < (unsigned)D && (unsigned)co.z < (unsigned)H &&
        (unsigned)co.w < (unsigned)W)
        v = (float)dout[((((int64_t)co.x * H + co.z) * W + co.w) * c + ch) * D + co.y];
    dfeat[t] = (T)v;
}

// ---- per-channel finalisation (one tiny launch instead of ~15 elementwise torch kernels) ---------
// forward: stats[2C] (+count) -> mean, invstd, scale, shift; running stats updated in place.
__global__ __launch_bounds__(256) void bn_finalize_fwd_kernel(const float *__restrict__ stats, const float *__restrict__ count_p,
                                                              const float *__restrict__ gamma, const float *__restrict__ beta,
                                                              float eps, float momentum, int c, float *__restrict__ mean_out,
                                                              float *__restrict__ invstd_out, float *__restrict__ scale,
                                                              float *__restrict__ shift, float *__restrict__ running_mean,
                                                              float *__restrict__ running_var,
                                                              long long *__restrict__ batches_tracked) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    if (i == 0 && batches_tracked) *batches_tracked += 1;   // nn.BatchNorm's num_batches_tracked, no extra launch
    const float n = *count_p;
    const float mean = stats[i] / n;
    float var = stats[c + i] / n - mean * mean;
    var = var > 0.f ? var : 0.f;
    const float invstd = rsqrtf(var + eps);
    const float sc = gamma[i] * invstd;
    mean_out[i] = mean;
    invstd_out[i] = invstd;
    scale[i] = sc;
    shift[i] = beta[i] - mean * sc;
    if (running_mean) {
        const float unbiased = var * (n / fmaxf(n - 1.f, 1.f));
        running_mean[i] = (1.f - momentum) * running_mean[i] + momentum * mean;
        running_var[i] = (1.f - momentum) * running_var[i] + momentum * unbiased;
    }
}

// backward: local sums (for dgamma/dbeta) and global sums (for dx) -> dgamma, dbeta, a, b, d
__global__ __launch_bounds__(256) void bn_finalize_bwd_kernel(const float *__restrict__ sums_local,
                                                              const float *__restrict__ sums_global,
                                                              const float *__restrict__ count_p, const float *__restrict__ gamma,
                                                              const float *__restrict__ mean, const float *__restrict__ invstd,
                                                              int c, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                              float *__restrict__ a, float *__restrict__ b, float *__restrict__ d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    const float n = *count_p;
    const float m = mean[i], is = invstd[i];
    dbeta[i] = sums_local[i];
    dgamma[i] = is * (sums_local[c + i] - m * sums_local[i]);
    const float sg = sums_global[i];
    const float dg_all = is * (sums_global[c + i] - m * sg);
    const float av = gamma[i] * is;
    const float bv = -(av * is) * dg_all / n;
    a[i] = av;
    b[i] = bv;
    d[i] = -(av * sg) / n - bv * m;
}


// ---- channel-major (NC[D]HW) batch norm for the PCR head --------------------------------------
// x[n][c][p], few channels (1..32) and up to 1.1e7 positions per plane.  MIOpen assigns one workgroup
// per channel to such shapes (1.2 ms per call); here every (channel, position-chunk) gets a block.
// grid (chunks, C); partial[chunk][2C] is reduced by partial_sum_kernel like the row-major case.
// relu with y == nullptr (r04): the mask y > 0 is re-derived from x as fma(x, scale, shift) > 0 - the expression the forward apply
// evaluated, so the same mask bit for bit - instead of reading the 362 MB output plane set a third time
// 4 consecutive positions of a plane as fp32, from an fp32 or a bf16-stored tensor (r06: the 16-channel PCR volume z, its gradient and the
// up-sampler's input gradient dx' are stored in bf16 - half the bytes of the twelve passes that cross them per step)
template <typename T> __device__ __forceinline__ float4 cm_load4(const T *p, int64_t i4) {
    if constexpr (sizeof(T) == 4) {
        return reinterpret_cast<const float4 *>(p)[i4];
    } else {
        const uint2 u = reinterpret_cast<const uint2 *>(p)[i4];
        return float4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
    }
}
template <typename T> __device__ __forceinline__ void cm_store4(T *p, int64_t i4, const float4 v) {
    if constexpr (sizeof(T) == 4) {
        reinterpret_cast<float4 *>(p)[i4] = v;
    } else {
        typedef __bf16 bf16x4s __attribute__((ext_vector_type(4)));
        bf16x4s o;
        o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
        reinterpret_cast<bf16x4s *>(p)[i4] = o;
    }
}

template <bool BWD, typename TX = float, typename TG = float>
__global__ __launch_bounds__(256) void cm_reduce_kernel(const TX *__restrict__ x, const TG *__restrict__ dy,
                                                        const float *__restrict__ y, int relu, int n, int c, int64_t p4,
                                                        int64_t chunk4, float *__restrict__ partial, const float *__restrict__ scale = nullptr,
                                                        const float *__restrict__ shift = nullptr) {
    __shared__ float lds[2][4];
    const int ch = blockIdx.y;
    const float msc = (BWD && scale) ? scale[ch] : 0.f, msh = (BWD && shift) ? shift[ch] : 0.f;
    const int64_t q0 = (int64_t)blockIdx.x * chunk4;
    const int64_t q1 = q0 + chunk4 < p4 ? q0 + chunk4 : p4;
    float s0 = 0.f, s1 = 0.f;
    for (int b = 0; b < n; ++b) {
        const int64_t base = ((int64_t)b * c + ch) * p4;
        for (int64_t q = q0 + threadIdx.x; q < q1; q += 256) {
            const float4 xv = cm_load4(x, base + q);
            if (BWD) {
                float4 g = cm_load4(dy, base + q);
                if (relu) {
                    const float4 yv = y ? reinterpret_cast<const float4 *>(y)[base + q]
                                        : float4{fmaf(xv.x, msc, msh), fmaf(xv.y, msc, msh), fmaf(xv.z, msc, msh), fmaf(xv.w, msc, msh)};
                    g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
                    g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
                }
                s0 += (g.x + g.y) + (g.z + g.w);
                s1 += (g.x * xv.x + g.y * xv.y) + (g.z * xv.z + g.w * xv.w);
            } else {
                s0 += (xv.x + xv.y) + (xv.z + xv.w);
                s1 += (xv.x * xv.x + xv.y * xv.y) + (xv.z * xv.z + xv.w * xv.w);
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s0 += __shfl_xor(s0, d, 64);
        s1 += __shfl_xor(s1, d, 64);
    }
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { lds[0][wid] = s0; lds[1][wid] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[(size_t)blockIdx.x * 2 * c + ch] = (lds[0][0] + lds[0][1]) + (lds[0][2] + lds[0][3]);
        partial[(size_t)blockIdx.x * 2 * c + c + ch] = (lds[1][0] + lds[1][1]) + (lds[1][2] + lds[1][3]);
    }
}

// fwd: y = x*scale[c] + shift[c] (relu optional).  bwd (dy given): dx = a[c]*g + b[c]*x + d[c], g = dy*(y>0 if relu)
template <bool BWD, typename TX = float, typename TG = float, typename TO = float>
__global__ __launch_bounds__(256) void cm_apply_kernel(const TX *__restrict__ x, const TG *__restrict__ dy,
                                                       const float *__restrict__ y, const float *__restrict__ v0,
                                                       const float *__restrict__ v1, const float *__restrict__ v2, int relu,
                                                       int c, int64_t p4, TO *__restrict__ out, const float *__restrict__ scale = nullptr,
                                                       const float *__restrict__ shift = nullptr) {
    // grid (position blocks, N*C)
    const int plane = blockIdx.y, ch = plane % c;
    const float a = v0[ch], b = v1[ch], d = BWD ? v2[ch] : 0.f;
    const float msc = (BWD && scale) ? scale[ch] : 0.f, msh = (BWD && shift) ? shift[ch] : 0.f;   // y == nullptr: mask from x (see cm_reduce_kernel)
    const int64_t base = (int64_t)plane * p4;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < p4; q += (int64_t)gridDim.x * 256) {
        const float4 xv = cm_load4(x, base + q);
        float4 o;
        if (BWD) {
            float4 g = cm_load4(dy, base + q);
            if (relu) {
                const float4 yv = y ? reinterpret_cast<const float4 *>(y)[base + q]
                                    : float4{fmaf(xv.x, msc, msh), fmaf(xv.y, msc, msh), fmaf(xv.z, msc, msh), fmaf(xv.w, msc, msh)};
                g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
                g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
            }
            o.x = fmaf(a, g.x, fmaf(b, xv.x, d)); o.y = fmaf(a, g.y, fmaf(b, xv.y, d));
            o.z = fmaf(a, g.z, fmaf(b, xv.z, d)); o.w = fmaf(a, g.w, fmaf(b, xv.w, d));
        } else {
            o.x = fmaf(xv.x, a, b); o.y = fmaf(xv.y, a, b); o.z = fmaf(xv.z, a, b); o.w = fmaf(xv.w, a, b);
            if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        }
        cm_store4(out, base + q, o);
    }
}

struct CmPlan {
    int chunks;
    int64_t chunk4;
    size_t ws_bytes;
};
static CmPlan cm_plan(int n, int c, int64_t p4) {
    CmPlan pl;
    int64_t want = ceil_div(2048, c > 0 ? c : 1);            // ~2048 blocks in total
    int64_t maxc = ceil_div(p4, 1024);                         // at least 1024 float4 per block and sample
    int64_t chunks = want < maxc ? want : maxc;
    if (chunks < 1) chunks = 1;
    pl.chunks = (int)chunks;
    pl.chunk4 = ceil_div(p4, chunks);
    pl.ws_bytes = align_up((size_t)chunks * 2 * c * sizeof(float), 256);
    return pl;
}


// ---- single-GPU fast path: reduction of the per-block partials fused with the finalisation ------
// one workgroup per channel: its lanes stride over the partials of (sum, sumsq) resp. (sum g, sum g*x).
__global__ __launch_bounds__(256) void bn_reduce_finalize_fwd_kernel(const float *__restrict__ partial, int nblocks, float n,
                                                                     const float *__restrict__ gamma,
                                                                     const float *__restrict__ beta, float eps, float momentum,
                                                                     int c, float *__restrict__ mean_out,
                                                                     float *__restrict__ invstd_out, float *__restrict__ scale,
                                                                     float *__restrict__ shift, float *__restrict__ running_mean,
                                                                     float *__restrict__ running_var,
                                                                     long long *__restrict__ batches_tracked, bool cm = false) {
    // one workgroup per channel (r05; one wave per channel before): the four waves stride over the partial rows together - four times the
    // loads in flight for a launch whose duration is the latency of its ~18 dependent-free load rounds - and fold in a fixed order
    const int i = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (i >= c) return;
    if (i == 0 && threadIdx.x == 0 && batches_tracked) *batches_tracked += 1;
    float s0 = 0.f, s1 = 0.f;
    for (int b = threadIdx.x; b < nblocks; b += 256) {
        s0 += cm ? partial[(size_t)i * nblocks + b] : partial[(size_t)b * 2 * c + i];
        s1 += cm ? partial[(size_t)(c + i) * nblocks + b] : partial[(size_t)b * 2 * c + c + i];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s0 += __shfl_xor(s0, d, 64);
        s1 += __shfl_xor(s1, d, 64);
    }
    __shared__ float red[2][4];
    if (lane == 0) {
        red[0][wave] = s0;
        red[1][wave] = s1;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    s0 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    s1 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const float mean = s0 / n;
    float var = s1 / n - mean * mean;
    var = var > 0.f ? var : 0.f;
    const float invstd = rsqrtf(var + eps);
    const float sc = gamma[i] * invstd;
    mean_out[i] = mean;
    invstd_out[i] = invstd;
    scale[i] = sc;
    shift[i] = beta[i] - mean * sc;
    if (running_mean) {
        const float unbiased = var * (n / fmaxf(n - 1.f, 1.f));
        running_mean[i] = (1.f - momentum) * running_mean[i] + momentum * mean;
        running_var[i] = (1.f - momentum) * running_var[i] + momentum * unbiased;
    }
}

__global__ __launch_bounds__(256) void bn_reduce_finalize_bwd_kernel(const float *__restrict__ partial, int nblocks, float n,
                                                                     const float *__restrict__ gamma,
                                                                     const float *__restrict__ mean,
                                                                     const float *__restrict__ invstd, int c,
                                                                     float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                                     float *__restrict__ a, float *__restrict__ b,
                                                                     float *__restrict__ d, bool cm = false) {
    const int i = blockIdx.x;   // one workgroup per channel, as in the forward kernel
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (i >= c) return;
    float sg = 0.f, sgx = 0.f;
    for (int k = threadIdx.x; k < nblocks; k += 256) {
        sg += cm ? partial[(size_t)i * nblocks + k] : partial[(size_t)k * 2 * c + i];
        sgx += cm ? partial[(size_t)(c + i) * nblocks + k] : partial[(size_t)k * 2 * c + c + i];
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        sg += __shfl_xor(sg, s, 64);
        sgx += __shfl_xor(sgx, s, 64);
    }
    __shared__ float red[2][4];
    if (lane == 0) {
        red[0][wave] = sg;
        red[1][wave] = sgx;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    sg = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    sgx = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const float m = mean[i], is = invstd[i];
    const float dg = is * (sgx - m * sg);
    dbeta[i] = sg;
    dgamma[i] = dg;
    const float av = gamma[i] * is;
    const float bv = -(av * is) * dg / n;
    a[i] = av;
    b[i] = bv;
    d[i] = -(av * sg) / n - bv * m;
}

struct RedPlan {
    int nblocks;
    int rows_per_block;
    size_t lds;
    size_t ws_bytes;
};
static RedPlan red_plan(int64_t n, int c) {
    RedPlan p;
    const int c4 = c / 4;
    const int lanes = RED_THREADS / c4 > 0 ? RED_THREADS / c4 : 1;
    int64_t nb = ceil_div(n > 0 ? n : 1, (int64_t)lanes * 16);
    if (nb > 1024) nb = 1024;
    p.nblocks = (int)nb;
    p.rows_per_block = (int)ceil_div(n > 0 ? n : 1, nb);
    p.lds = (size_t)lanes * 2 * c * sizeof(float);
    p.ws_bytes = align_up((size_t)nb * 2 * c * sizeof(float), 256);
    return p;
}

static int check_c(int c, const char *who) {
    if (c <= 0 || (c & 3) || c > 1024) {
        set_error("%s: channel count %d must be a positive multiple of 4 (<=1024)", who, c);
        return S2D_ERR_UNSUPPORTED;
    }
    return 0;
}

// ---- row-major bf16 batch norm (BatchNorm2d on NHWC bf16 activations of the BEV neck / head) ------
// x, y, dy, dx are [n rows = N*H*W][c] bf16; statistics, scale/shift and the reductions are fp32.
// A thread owns 8 channels (one 16-byte access).  The ReLU mask of the backward is recomputed from
// x (y > 0  <=>  fma(x, scale, shift) > 0), so the forward output need not be kept for it.
typedef __bf16 bf16x8r __attribute__((ext_vector_type(8)));
constexpr int ROW_UNROLL = 4;   // rows per thread and trip in the row-major bf16 kernels

// activation fused behind the normalisation: 0 none, 1 ReLU, 2 exact (erf) GELU = nn.GELU() of the S2D module's conv-BN-GELU groups
// (gelu_f / gelu_grad_f: s2d_common.h - the conv epilogue that emits this layer's backward sums evaluates the same expression)

// ACT / HAS_Y are compile-time: with run-time flags hipcc keeps a uniform branch per element in the unrolled bodies
template <bool BWD, int ACT, bool HAS_Y>
__global__ __launch_bounds__(RED_THREADS) void row_reduce_bf16_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ dy,
                                                                      const __bf16 *__restrict__ y,
                                                                      const float *__restrict__ scale,
                                                                      const float *__restrict__ shift, int64_t n, int c,
                                                                      int rows_per_block, float *__restrict__ partial, int dy_ld) {
    // dy_ld (r04): row stride of dy in elements (>= c, a multiple of 8) - the gradient may be a channel slice of a wider row-major
    // tensor (the RPN's concatenated deblock outputs: no contiguous copy of the slice is made)
    constexpr bool relu = ACT == 1, gelu = ACT == 2, act = ACT != 0;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [row_lanes][2][C]
    const int c8 = c >> 3;
    const int lanes = RED_THREADS / c8;
    const int grp = threadIdx.x % c8, rl = threadIdx.x / c8;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    float s0[8], s1[8], sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        s0[e] = 0.f; s1[e] = 0.f;
        sc[e] = (BWD && act && !HAS_Y) ? scale[grp * 8 + e] : 0.f;
        sh[e] = (BWD && act && !HAS_Y) ? shift[grp * 8 + e] : 0.f;
    }
    if (rl < lanes) {
        constexpr int RU = BWD ? ROW_UNROLL : 2 * ROW_UNROLL;   // the statistics pass carries one tensor: twice the rows in flight
        for (int64_t rb = r0 + rl; rb < r1; rb += (int64_t)RU * lanes) {
            bf16x8r xv[RU], gv[BWD ? RU : 1], yv[(BWD && HAS_Y) ? RU : 1];
#pragma unroll
            for (int u = 0; u < RU; ++u) {   // all loads of the trip first; rows past the end re-read row rb
                const int64_t r = rb + (int64_t)u * lanes < r1 ? rb + (int64_t)u * lanes : rb;
                xv[u] = reinterpret_cast<const bf16x8r *>(x + r * c)[grp];
                if constexpr (BWD) {
                    gv[u] = reinterpret_cast<const bf16x8r *>(dy + r * dy_ld)[grp];
                    if constexpr (relu && HAS_Y) yv[u] = reinterpret_cast<const bf16x8r *>(y + r * c)[grp];
                }
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                if (rb + (int64_t)u * lanes >= r1) break;
                if constexpr (BWD) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float xf = (float)xv[u][e];
                        float g = (float)gv[u][e];
                        if constexpr (relu && HAS_Y) g = (float)yv[u][e] > 0.f ? g : 0.f;
                        else if constexpr (relu) g = fmaf(xf, sc[e], sh[e]) > 0.f ? g : 0.f;
                        else if constexpr (gelu) g *= gelu_grad_f(fmaf(xf, sc[e], sh[e]));
                        s0[e] += g;
                        s1[e] += g * xf;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float xf = (float)xv[u][e];
                        s0[e] += xf;
                        s1[e] += xf * xf;
                    }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            lds[(size_t)rl * 2 * c + grp * 8 + e] = s0[e];
            lds[(size_t)rl * 2 * c + c + grp * 8 + e] = s1[e];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * c; e += RED_THREADS) {
        float s = 0.f;
        for (int l = 0; l < lanes; ++l) s += lds[(size_t)l * 2 * c + e];
        partial[(size_t)e * gridDim.x + blockIdx.x] = s;   // element-major: the finalize kernels' lanes stride over the blocks
    }
}

// apply kernels: blockDim.x = lanes*c8 (a multiple of c8), thread -> (row lane, 8-channel group); the per-channel constants
// live in registers, rows are strided over the grid.
template <bool RES, int ACT>
__global__ __launch_bounds__(256) void row_apply_bf16_kernel(const __bf16 *__restrict__ x, const float *__restrict__ scale,
                                                             const float *__restrict__ shift, const __bf16 *__restrict__ res_p,
                                                             int64_t n, int c8, __bf16 *__restrict__ y, int y_c8) {
    // y_c8 (r04): row stride of y in 8-channel groups (>= c8) - y may be a channel slice of a wider row-major tensor
    constexpr bool relu = ACT == 1, gelu = ACT == 2;
    const __bf16 *__restrict__ res = RES ? res_p : nullptr;
    const int g = threadIdx.x % c8, rl = threadIdx.x / c8, lanes = blockDim.x / c8;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = scale[g * 8 + e];
        sh[e] = shift[g * 8 + e];
    }
    // ROW_UNROLL rows per trip with every load issued before the first use (the kernel is a pure HBM stream: what
    // matters is bytes in flight per wave); rows past the end re-read the trip's first row and are not stored
    const int64_t stride = (int64_t)gridDim.x * lanes;
    for (int64_t r0 = (int64_t)blockIdx.x * lanes + rl; r0 < n; r0 += ROW_UNROLL * stride) {
        bf16x8r xv[ROW_UNROLL], rv[ROW_UNROLL];
#pragma unroll
        for (int u = 0; u < ROW_UNROLL; ++u) {
            const int64_t r = r0 + u * stride < n ? r0 + u * stride : r0;
            xv[u] = reinterpret_cast<const bf16x8r *>(x)[r * c8 + g];
            if (RES) rv[u] = reinterpret_cast<const bf16x8r *>(res)[r * c8 + g];
        }
#pragma unroll
        for (int u = 0; u < ROW_UNROLL; ++u) {
            const int64_t r = r0 + u * stride;
            if (r >= n) break;
            bf16x8r o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = fmaf((float)xv[u][e], sc[e], sh[e]);
                if (RES) v += (float)rv[u][e];
                if (relu) v = fmaxf(v, 0.f);
                if (gelu) v = gelu_f(v);
                o[e] = (__bf16)v;
            }
            reinterpret_cast<bf16x8r *>(y)[r * y_c8 + g] = o;
        }
    }
}

template <int ACT, bool HAS_Y, bool HAS_DRES>
__global__ __launch_bounds__(256) void row_bwd_apply_bf16_kernel(const __bf16 *__restrict__ dy, const __bf16 *__restrict__ x,
                                                                 const __bf16 *__restrict__ y,
                                                                 const float *__restrict__ scale, const float *__restrict__ shift,
                                                                 const float *__restrict__ a, const float *__restrict__ b,
                                                                 const float *__restrict__ d, int64_t n, int c8,
                                                                 __bf16 *__restrict__ dx, __bf16 *__restrict__ dres, int dy_c8) {
    constexpr bool relu = ACT == 1, gelu = ACT == 2, act = ACT != 0;
    const int g = threadIdx.x % c8, rl = threadIdx.x / c8, lanes = blockDim.x / c8;
    float sc[8], sh[8], av[8], bv[8], dv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = g * 8 + e;
        sc[e] = (act && !HAS_Y) ? scale[ch] : 0.f;
        sh[e] = (act && !HAS_Y) ? shift[ch] : 0.f;
        av[e] = a[ch];
        bv[e] = b[ch];
        dv[e] = d[ch];
    }
    const int64_t stride = (int64_t)gridDim.x * lanes;
    for (int64_t r0 = (int64_t)blockIdx.x * lanes + rl; r0 < n; r0 += ROW_UNROLL * stride) {
        bf16x8r xv[ROW_UNROLL], gv[ROW_UNROLL], yv[ROW_UNROLL];
#pragma unroll
        for (int u = 0; u < ROW_UNROLL; ++u) {
            const int64_t r = r0 + u * stride < n ? r0 + u * stride : r0;
            xv[u] = reinterpret_cast<const bf16x8r *>(x)[r * c8 + g];
            gv[u] = reinterpret_cast<const bf16x8r *>(dy)[r * dy_c8 + g];
            if (relu && HAS_Y) yv[u] = reinterpret_cast<const bf16x8r *>(y)[r * c8 + g];
        }
#pragma unroll
        for (int u = 0; u < ROW_UNROLL; ++u) {
            const int64_t r = r0 + u * stride;
            if (r >= n) break;
            bf16x8r o, gm;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xf = (float)xv[u][e];
                float gg = (float)gv[u][e];
                if (relu) gg = (HAS_Y ? (float)yv[u][e] : fmaf(xf, sc[e], sh[e])) > 0.f ? gg : 0.f;
                if (gelu) gg *= gelu_grad_f(fmaf(xf, sc[e], sh[e]));
                if (HAS_DRES) gm[e] = (__bf16)gg;
                o[e] = (__bf16)fmaf(av[e], gg, fmaf(bv[e], xf, dv[e]));
            }
            reinterpret_cast<bf16x8r *>(dx)[r * c8 + g] = o;
            if (HAS_DRES) reinterpret_cast<bf16x8r *>(dres)[r * c8 + g] = gm;   // gradient of the residual branch: masked dy
        }
    }
}

struct RowLaunch {
    unsigned blocks, threads;
};
static RowLaunch row_launch(int64_t n, int c8) {
    const int lanes = 256 / c8 > 0 ? 256 / c8 : 1;
    RowLaunch l;
    l.threads = (unsigned)(lanes * c8);
    int64_t blocks = ceil_div(n, (int64_t)lanes * 4);   // ~4 rows per thread
    if (blocks > 16384) blocks = 16384;
    l.blocks = (unsigned)blocks;
    return l;
}

static RedPlan row_plan_bf16(int64_t n, int c) {
    RedPlan p;
    const int c8 = c / 8;
    const int lanes = RED_THREADS / c8 > 0 ? RED_THREADS / c8 : 1;
    // rows per thread (S2D_ROW_RPT: A/B hook).  r06: 8 instead of 16 - a 128-channel map of 141 376 pixels then runs as 1 105 workgroups
    // instead of 553 (two trips of four rows per thread instead of four): the statistics / backward-sum passes are latency-bound at these
    // sizes, not bandwidth-bound (measured 4 / 6 / 8 / 16 / 32 rows: 14.1 / 13.0 / 13.3 / 16.2 / 23.9 us for the backward sums of that map)
    static int rpt = 0;
    if (!rpt) {
        const char *e = getenv("S2D_ROW_RPT");
        rpt = e ? atoi(e) : 8;
        if (rpt < 4 || rpt > 64) rpt = 8;
    }
    int64_t nb = ceil_div(n > 0 ? n : 1, (int64_t)lanes * rpt);
    if (nb > 2048) nb = 2048;
    p.nblocks = (int)nb;
    p.rows_per_block = (int)ceil_div(n > 0 ? n : 1, nb);
    p.lds = (size_t)lanes * 2 * c * sizeof(float);
    p.ws_bytes = align_up((size_t)nb * 2 * c * sizeof(float), 256);
    return p;
}

static int check_c8(int c, const char *who) {
    if (c <= 0 || (c & 7) || c > 1024) {
        set_error("%s: channel count %d must be a positive multiple of 8 (<=1024)", who, c);
        return S2D_ERR_UNSUPPORTED;
    }
    return 0;
}

}  // namespace s2d

using namespace s2d;

extern "C" size_t s2d_bn1d_workspace_bytes(int64_t n, int c) {
    if (n < 0 || c <= 0 || (c & 3) || c > 1024) return 0;
    return red_plan(n, c).ws_bytes;
}

extern "C" int s2d_bn1d_stats_f32(const float *x, int64_t n, int c, float *stats, void *ws, size_t ws_bytes,
                                  s2d_stream_t stream) {
    int rc = check_c(c, "bn1d_stats");
    if (rc) return rc;
    S2D_CHECK_ARG(n >= 0 && stats && (n == 0 || x), "bn1d_stats: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        S2D_HIP(hipMemsetAsync(stats, 0, 2 * c * sizeof(float), st));
        return S2D_OK;
    }
    RedPlan p = red_plan(n, c);
    if (!ws || ws_bytes < p.ws_bytes) {
        set_error("bn1d_stats: workspace too small (%zu < %zu)", ws_bytes, p.ws_bytes);
        return S2D_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(col_reduce_kernel<false>, dim3(p.nblocks), dim3(RED_THREADS), p.lds, st, x, nullptr, nullptr, 0, n, c,
                       p.rows_per_block, nullptr, (float *)ws);
    hipLaunchKernelGGL(partial_sum_kernel, dim3((2 * c + 3) / 4), dim3(256), 0, st, (const float *)ws, p.nblocks, 2 * c,
                       stats);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_bn1d_apply_f32(const float *x, const float *scale, const float *shift, const float *residual, int relu,
                                  int64_t n, int c, float *y, s2d_stream_t stream) {
    int rc = check_c(c, "bn1d_apply");
    if (rc) return rc;
    S2D_CHECK_ARG(n >= 0 && scale && shift && (n == 0 || (x && y)), "bn1d_apply: bad argument");
    if (n == 0) return S2D_OK;
    const int64_t n4 = n * (c / 4);
    int64_t blocks = ceil_div(n4, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, scale, shift, residual,
                       relu, n4, c / 4, y);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_bn1d_bwd_reduce_f32(const float *dy, const float *y, const float *x, int relu, int64_t n, int c,
                                       float *g_out, float *sums, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    int rc = check_c(c, "bn1d_bwd_reduce");
    if (rc) return rc;
    S2D_CHECK_ARG(n >= 0 && sums && (n == 0 || (dy && x)) && (!relu || n == 0 || y), "bn1d_bwd_reduce: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        S2D_HIP(hipMemsetAsync(sums, 0, 2 * c * sizeof(float), st));
        return S2D_OK;
    }
    RedPlan p = red_plan(n, c);
    if (!ws || ws_bytes < p.ws_bytes) {
        set_error("bn1d_bwd_reduce: workspace too small (%zu < %zu)", ws_bytes, p.ws_bytes);
        return S2D_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(col_reduce_kernel<true>, dim3(p.nblocks), dim3(RED_THREADS), p.lds, st, x, dy, y, relu, n, c,
                       p.rows_per_block, g_out, (float *)ws);
    hipLaunchKernelGGL(partial_sum_kernel, dim3((2 * c + 3) / 4), dim3(256), 0, st, (const float *)ws, p.nblocks, 2 * c,
                       sums);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_bn1d_bwd_apply_f32(const float *g, const float *x, const float *a, const float *b, const float *d,
                                      int64_t n, int c, float *dx, s2d_stream_t stream) {
    int rc = check_c(c, "bn1d_bwd_apply");
    if (rc) return rc;
    S2D_CHECK_ARG(n >= 0 && a && b && d && (n == 0 || (g && x && dx)), "bn1d_bwd_apply: bad argument");
    if (n == 0) return S2D_OK;
    const int64_t n4 = n * (c / 4);
    int64_t blocks = ceil_div(n4, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, x, a, b, d, n4,
                       c / 4, dx);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_densify_fwd_f32(const float *feat, const int32_t *coors, int64_t n, int batch, const int32_t shape[3],
                                   int c, float *out, s2d_stream_t stream) {
    S2D_CHECK_ARG(n >= 0 && batch > 0 && shape && c > 0 && out, "densify_fwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const size_t bytes = (size_t)batch * c * shape[0] * shape[1] * shape[2] * sizeof(float);
    S2D_HIP(hipMemsetAsync(out, 0, bytes, st));
    if (n == 0) return S2D_OK;
    S2D_CHECK_ARG(feat && coors, "densify_fwd: null input");
    const int64_t threads = n * c;
    hipLaunchKernelGGL(densify_fwd_kernel, dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0, st, feat, coors, n, batch,
                       shape[0], shape[1], shape[2], c, out);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_densify_bwd_f32(const float *dout, const int32_t *coors, int64_t n, int batch, const int32_t shape[3],
                                   int c, float *dfeat, s2d_stream_t stream) {
    S2D_CHECK_ARG(n >= 0 && batch > 0 && shape && c > 0, "densify_bwd: bad argument");
    if (n == 0) return S2D_OK;
    S2D_CHECK_ARG(dout && coors && dfeat, "densify_bwd: null argument");
    const int64_t threads = n * c;
    hipLaunchKernelGGL(densify_bwd_kernel, dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0, (hipStream_t)stream, dout,
                       coors, n, batch, shape[0], shape[1], shape[2], c, dfeat);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}


extern "C" int s2d_densify_bev_fwd_bf16(const void *feat, int feat_bf16, const int32_t *coors, int64_t n, int batch,
                                        const int32_t shape[3], int c, void *out, s2d_stream_t stream) {
    S2D_CHECK_ARG(n >= 0 && batch > 0 && shape && c > 0 && out, "densify_bev_fwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const size_t bytes = (size_t)batch * c * shape[0] * shape[1] * shape[2] * 2;
    S2D_HIP(hipMemsetAsync(out, 0, bytes, st));
    if (n == 0) return S2D_OK;
    S2D_CHECK_ARG(feat && coors, "densify_bev_fwd: null input");
    if (feat_bf16)
        hipLaunchKernelGGL(densify_bev_fwd_kernel<__bf16>, dim3((unsigned)ceil_div(n * c, 256)), dim3(256), 0, st,
                           (const __bf16 *)feat, coors, n, batch, shape[0], shape[1], shape[2], c, (__bf16 *)out);
    else
        hipLaunchKernelGGL(densify_bev_fwd_kernel<float>, dim3((unsigned)ceil_div(n * c, 256)), dim3(256), 0, st,
                           (const float *)feat, coors, n, batch, shape[0], shape[1], shape[2], c, (__bf16 *)out);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_densify_bev_bwd_bf16(const void *dout, const int32_t *coors, int64_t n, int batch, const int32_t shape[3],
                                        int c, void *dfeat, int feat_bf16, s2d_stream_t stream) {
    S2D_CHECK_ARG(n >= 0 && batch > 0 && shape && c > 0, "densify_bev_bwd: bad argument");
    if (n == 0) return S2D_OK;
    S2D_CHECK_ARG(dout && coors && dfeat, "densify_bev_bwd: null argument");
    if (feat_bf16)
        hipLaunchKernelGGL(densify_bev_bwd_kernel<__bf16>, dim3((unsigned)ceil_div(n * c, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const __bf16 *)dout, coors, n, batch, shape[0], shape[1], shape[2], c, (__bf16 *)dfeat);
    else
        hipLaunchKernelGGL(densify_bev_bwd_kernel<float>, dim3((unsigned)ceil_div(n * c, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const __bf16 *)dout, coors, n, batch, shape[0], shape[1], shape[2], c, (float *)dfeat);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_bn1d_finalize_fwd_f32(const float *stats, const float *count, const float *gamma, const float *beta,
                                         float eps, float momentum, int c, float *mean, float *invstd, float *scale,
                                         float *shift, float *running_mean, float *running_var, int64_t *batches_tracked,
                                         s2d_stream_t stream) {
    S2D_CHECK_ARG(c > 0 && stats && count && gamma && beta && mean && invstd && scale && shift, "bn1d_finalize_fwd: null argument");
    S2D_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "bn1d_finalize_fwd: running stats must come in pairs");
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, count, gamma,
                       beta, eps, momentum, c, mean, invstd, scale, shift, running_mean, running_var,
                       (long long *)batches_tracked);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_bn1d_finalize_bwd_f32(const float *sums_local, const float *sums_global, const float *count,
                                         const float *gamma, const float *mean, const float *invstd, int c, float *dgamma,
                                         float *dbeta, float *a, float *b, float *d, s2d_stream_t stream) {
    S2D_CHECK_ARG(c > 0 && sums_local && sums_global && count && gamma && mean && invstd && dgamma && dbeta && a && b && d,
                  "bn1d_finalize_bwd: null argument");
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums_local,
                       sums_global, count, gamma, mean, invstd, c, dgamma, dbeta, a, b, d);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

// ---- channel-major batch norm entry points -------------------------------------------------------
extern "C" size_t s2d_bncm_workspace_bytes(int batch, int c, int64_t positions) {
    if (batch <= 0 || c <= 0 || positions <= 0 || (positions & 3)) return 0;
    return cm_plan(batch, c, positions / 4).ws_bytes;
}

static int bncm_reduce(bool bwd, const float *x, const float *dy, const float *y, int relu, int batch, int c,
                       int64_t positions, float *sums, void *ws, size_t ws_bytes, hipStream_t st, const float *scale = nullptr,
                       const float *shift = nullptr) {
    S2D_CHECK_ARG(batch > 0 && c > 0 && c <= 65535 && positions > 0 && x && sums, "bncm: bad argument");
    if (positions & 3) {
        set_error("bncm: positions per plane (%lld) must be a multiple of 4", (long long)positions);
        return S2D_ERR_UNSUPPORTED;
    }
    CmPlan pl = cm_plan(batch, c, positions / 4);
    if (!ws || ws_bytes < pl.ws_bytes) {
        set_error("bncm: workspace too small (%zu < %zu)", ws_bytes, pl.ws_bytes);
        return S2D_ERR_WORKSPACE;
    }
    if (bwd)
        hipLaunchKernelGGL(cm_reduce_kernel<true>, dim3(pl.chunks, c), dim3(256), 0, st, x, dy, y, relu, batch, c,
                           positions / 4, pl.chunk4, (float *)ws, scale, shift);
    else
        hipLaunchKernelGGL(cm_reduce_kernel<false>, dim3(pl.chunks, c), dim3(256), 0, st, x, (const float *)nullptr, (const float *)nullptr, 0, batch, c,
                           positions / 4, pl.chunk4, (float *)ws, (const float *)nullptr, (const float *)nullptr);
    hipLaunchKernelGGL(partial_sum_kernel, dim3((2 * c + 3) / 4), dim3(256), 0, st, (const float *)ws, pl.chunks, 2 * c, sums);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_bncm_stats_f32(const float *x, int batch, int c, int64_t positions, float *stats, void *ws,
                                  size_t ws_bytes, s2d_stream_t stream) {
    return bncm_reduce(false, x, nullptr, nullptr, 0, batch, c, positions, stats, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int s2d_bncm_bwd_reduce_f32(const float *dy, const float *y, const float *x, int relu, int batch, int c,
                                       int64_t positions, float *sums, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(dy && (!relu || y), "bncm_bwd_reduce: null argument");
    return bncm_reduce(true, x, dy, y, relu, batch, c, positions, sums, ws, ws_bytes, (hipStream_t)stream);
}

/* the same sums with the ReLU mask re-derived from x: y > 0 <=> fma(x, scale[c], shift[c]) > 0 (what s2d_bncm_apply_f32 evaluated) */
extern "C" int s2d_bncm_bwd_reduce_x_f32(const float *dy, const float *x, const float *scale, const float *shift, int batch, int c,
                                         int64_t positions, float *sums, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(dy && scale && shift, "bncm_bwd_reduce_x: null argument");
    return bncm_reduce(true, x, dy, nullptr, 1, batch, c, positions, sums, ws, ws_bytes, (hipStream_t)stream, scale, shift);
}

extern "C" int s2d_bncm_apply_f32(const float *x, const float *scale, const float *shift, int relu, int batch, int c,
                                  int64_t positions, float *y, s2d_stream_t stream) {
    S2D_CHECK_ARG(x && scale && shift && y && batch > 0 && c > 0 && positions > 0 && !(positions & 3) &&
                      (int64_t)batch * c <= 65535, "bncm_apply: bad argument");
    const int64_t p4 = positions / 4;
    int64_t bx = ceil_div(p4, 256);
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(cm_apply_kernel<false>, dim3((unsigned)bx, batch * c), dim3(256), 0, (hipStream_t)stream, x, (const float *)nullptr,
                       (const float *)nullptr, scale, shift, (const float *)nullptr, relu, c, p4, y, (const float *)nullptr, (const float *)nullptr);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_bncm_bwd_apply_f32(const float *dy, const float *y, const float *x, const float *a, const float *b,
                                      const float *d, int relu, int batch, int c, int64_t positions, float *dx,
                                      s2d_stream_t stream) {
    S2D_CHECK_ARG(dy && x && a && b && d && dx && (!relu || y) && batch > 0 && c > 0 && positions > 0 && !(positions & 3) &&
                      (int64_t)batch * c <= 65535, "bncm_bwd_apply: bad argument");
    const int64_t p4 = positions / 4;
    int64_t bx = ceil_div(p4, 256);
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(cm_apply_kernel<true>, dim3((unsigned)bx, batch * c), dim3(256), 0, (hipStream_t)stream, x, dy, y, a, b,
                       d, relu, c, p4, dx, (const float *)nullptr, (const float *)nullptr);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

/* r06: the same two passes on bf16-STORED tensors (x_bf16 / dy_bf16 / dx_bf16: 0 = fp32, 1 = bf16 [n][c][positions]; fp32 arithmetic and sums):
 * the 16-channel PCR volume z [4,16,10,376,376], the up-sampler's input gradient and the batch norm's own output gradient in bf16 halve
 * the 1.8 GB these two passes move per step.  Supported type combinations: (x, dy, dx) all fp32 | (fp32, bf16, fp32) | all bf16. */
template <typename TX, typename TG>
static int bncm_reduce_x_t(const void *dy, const void *x, const float *scale, const float *shift, int batch, int c, int64_t positions, float *sums,
                           void *ws, size_t ws_bytes, hipStream_t st) {
    S2D_CHECK_ARG(batch > 0 && c > 0 && c <= 65535 && positions > 0 && x && dy && sums && scale && shift, "bncm_bwd_reduce_x_t: bad argument");
    if (positions & 3) {
        set_error("bncm: positions per plane (%lld) must be a multiple of 4", (long long)positions);
        return S2D_ERR_UNSUPPORTED;
    }
    CmPlan pl = cm_plan(batch, c, positions / 4);
    if (!ws || ws_bytes < pl.ws_bytes) {
        set_error("bncm: workspace too small (%zu < %zu)", ws_bytes, pl.ws_bytes);
        return S2D_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL((cm_reduce_kernel<true, TX, TG>), dim3(pl.chunks, c), dim3(256), 0, st, (const TX *)x, (const TG *)dy, (const float *)nullptr, 1,
                       batch, c, positions / 4, pl.chunk4, (float *)ws, scale, shift);
    hipLaunchKernelGGL(partial_sum_kernel, dim3((2 * c + 3) / 4), dim3(256), 0, st, (const float *)ws, pl.chunks, 2 * c, sums);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
extern "C" int s2d_bncm_bwd_reduce_x_typed(const void *dy, int dy_bf16, const void *x, int x_bf16, const float *scale, const float *shift, int batch,
                                           int c, int64_t positions, float *sums, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!x_bf16 && !dy_bf16) return bncm_reduce_x_t<float, float>(dy, x, scale, shift, batch, c, positions, sums, ws, ws_bytes, st);
    if (!x_bf16 && dy_bf16) return bncm_reduce_x_t<float, __bf16>(dy, x, scale, shift, batch, c, positions, sums, ws, ws_bytes, st);
    if (x_bf16 && dy_bf16) return bncm_reduce_x_t<__bf16, __bf16>(dy, x, scale, shift, batch, c, positions, sums, ws, ws_bytes, st);
    set_error("bncm_bwd_reduce_x_typed: unsupported storage combination");
    return S2D_ERR_UNSUPPORTED;
}

template <typename TX, typename TG, typename TO>
static int bncm_apply_x_t(const void *dy, const void *x, const float *scale, const float *shift, const float *a, const float *b, const float *d,
                          int batch, int c, int64_t positions, void *dx, hipStream_t st) {
    S2D_CHECK_ARG(dy && x && scale && shift && a && b && d && dx && batch > 0 && c > 0 && positions > 0 && !(positions & 3) &&
                      (int64_t)batch * c <= 65535, "bncm_bwd_apply_x_typed: bad argument");
    const int64_t p4 = positions / 4;
    int64_t bx = ceil_div(p4, 256);
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL((cm_apply_kernel<true, TX, TG, TO>), dim3((unsigned)bx, batch * c), dim3(256), 0, st, (const TX *)x, (const TG *)dy,
                       (const float *)nullptr, a, b, d, 1, c, p4, (TO *)dx, scale, shift);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
extern "C" int s2d_bncm_bwd_apply_x_typed(const void *dy, int dy_bf16, const void *x, int x_bf16, const float *scale, const float *shift,
                                          const float *a, const float *b, const float *d, int batch, int c, int64_t positions, void *dx, int dx_bf16,
                                          s2d_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!x_bf16 && !dy_bf16 && !dx_bf16) return bncm_apply_x_t<float, float, float>(dy, x, scale, shift, a, b, d, batch, c, positions, dx, st);
    if (!x_bf16 && dy_bf16 && !dx_bf16) return bncm_apply_x_t<float, __bf16, float>(dy, x, scale, shift, a, b, d, batch, c, positions, dx, st);
    if (x_bf16 && dy_bf16 && dx_bf16) return bncm_apply_x_t<__bf16, __bf16, __bf16>(dy, x, scale, shift, a, b, d, batch, c, positions, dx, st);
    set_error("bncm_bwd_apply_x_typed: unsupported storage combination");
    return S2D_ERR_UNSUPPORTED;
}

/* dx = a g + b x + d with g = dy where fma(x, scale, shift) > 0 (the ReLU mask re-derived from x, see s2d_bncm_bwd_reduce_x_f32) */
extern "C" int s2d_bncm_bwd_apply_x_f32(const float *dy, const float *x, const float *scale, const float *shift, const float *a, const float *b,
                                        const float *d, int batch, int c, int64_t positions, float *dx, s2d_stream_t stream) {
    S2D_CHECK_ARG(dy && x && scale && shift && a && b && d && dx && batch > 0 && c > 0 && positions > 0 && !(positions & 3) &&
                      (int64_t)batch * c <= 65535, "bncm_bwd_apply_x: bad argument");
    const int64_t p4 = positions / 4;
    int64_t bx = ceil_div(p4, 256);
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(cm_apply_kernel<true>, dim3((unsigned)bx, batch * c), dim3(256), 0, (hipStream_t)stream, x, dy, (const float *)nullptr, a, b, d, 1, c,
                       p4, dx, scale, shift);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

// ---- fused single-GPU entry points (no cross-rank statistics exchange) -----------------------------
extern "C" int s2d_bn1d_stats_finalize_f32(const float *x, int64_t n, int c, const float *gamma, const float *beta, float eps,
                                           float momentum, float *mean, float *invstd, float *scale, float *shift,
                                           float *running_mean, float *running_var, int64_t *batches_tracked, void *ws,
                                           size_t ws_bytes, s2d_stream_t stream) {
    int rc = check_c(c, "bn1d_stats_finalize");
    if (rc) return rc;
    S2D_CHECK_ARG(n > 0 && x && gamma && beta && mean && invstd && scale && shift, "bn1d_stats_finalize: bad argument");
    hipStream_t st = (hipStream_t)stream;
    RedPlan p = red_plan(n, c);
    if (!ws || ws_bytes < p.ws_bytes) {
        set_error("bn1d_stats_finalize: workspace too small (%zu < %zu)", ws_bytes, p.ws_bytes);
        return S2D_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(col_reduce_kernel<false>, dim3(p.nblocks), dim3(RED_THREADS), p.lds, st, x, nullptr, nullptr, 0, n, c,
                       p.rows_per_block, nullptr, (float *)ws);
    hipLaunchKernelGGL(bn_reduce_finalize_fwd_kernel, dim3(c), dim3(256), 0, st, (const float *)ws, p.nblocks, (float)n,
                       gamma, beta, eps, momentum, c, mean, invstd, scale, shift, running_mean, running_var,
                       (long long *)batches_tracked);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_bn1d_bwd_reduce_finalize_f32(const float *dy, const float *y, const float *x, int relu, int64_t n, int c,
                                                const float *gamma, const float *mean, const float *invstd, float *g_out,
                                                float *dgamma, float *dbeta, float *a, float *b, float *d, void *ws,
                                                size_t ws_bytes, s2d_stream_t stream) {
    int rc = check_c(c, "bn1d_bwd_reduce_finalize");
    if (rc) return rc;
    S2D_CHECK_ARG(n > 0 && dy && x && (!relu || y) && gamma && mean && invstd && dgamma && dbeta && a && b && d,
                  "bn1d_bwd_reduce_finalize: bad argument");
    hipStream_t st = (hipStream_t)stream;
    RedPlan p = red_plan(n, c);
    if (!ws || ws_bytes < p.ws_bytes) {
        set_error("bn1d_bwd_reduce_finalize: workspace too small (%zu < %zu)", ws_bytes, p.ws_bytes);
        return S2D_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(col_reduce_kernel<true>, dim3(p.nblocks), dim3(RED_THREADS), p.lds, st, x, dy, y, relu, n, c,
                       p.rows_per_block, g_out, (float *)ws);
    hipLaunchKernelGGL(bn_reduce_finalize_bwd_kernel, dim3(c), dim3(256), 0, st, (const float *)ws, p.nblocks, (float)n,
                       gamma, mean, invstd, c, dgamma, dbeta, a, b, d);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

// ---- row-major bf16 batch norm entry points ---------------------------------------------------------
extern "C" size_t s2d_bnrow_workspace_bytes(int64_t n, int c) {
    if (n < 0 || c <= 0 || (c & 7) || c > 1024) return 0;
    return row_plan_bf16(n, c).ws_bytes;
}

// fwd=1: x -> (sum x, sum x^2); fwd=0: (dy, x, scale, shift, relu) -> (sum g, sum g*x).  out: [2c] sums when
// fin == nullptr-style split mode is wanted (stats != nullptr), else fused finalisation.
static int bnrow_reduce(bool bwd, const void *x, const void *dy, const void *y, const float *scale, const float *shift, int relu,
                        int64_t n, int c, void *ws, size_t ws_bytes, hipStream_t st, RedPlan *plan_out, const char *who, int dy_ld = 0) {
    int rc = check_c8(c, who);
    if (rc) return rc;
    dy_ld = dy_ld ? dy_ld : c;
    S2D_CHECK_ARG(dy_ld >= c && dy_ld % 8 == 0, "bnrow reduce: the row stride of dy must be a multiple of 8 and >= c");
    S2D_CHECK_ARG(n > 0 && x && (!bwd || dy) && (!(bwd && relu) || (y && relu == 1) || (scale && shift)) && relu >= 0 && relu <= 2,
                  "bnrow reduce: bad argument");
    RedPlan p = row_plan_bf16(n, c);
    if (!ws || ws_bytes < p.ws_bytes) {
        set_error("%s: workspace too small (%zu < %zu)", who, ws_bytes, p.ws_bytes);
        return S2D_ERR_WORKSPACE;
    }
#define S2D_ROW_REDUCE(B, R, Y)                                                                                              \
    hipLaunchKernelGGL((row_reduce_bf16_kernel<B, R, Y>), dim3(p.nblocks), dim3(RED_THREADS), p.lds, st, (const __bf16 *)x, \
                       (const __bf16 *)dy, (const __bf16 *)y, scale, shift, n, c, p.rows_per_block, (float *)ws, dy_ld)
    if (!bwd) S2D_ROW_REDUCE(false, 0, false);
    else if (!relu) S2D_ROW_REDUCE(true, 0, false);
    else if (relu == 2) S2D_ROW_REDUCE(true, 2, false);
    else if (y) S2D_ROW_REDUCE(true, 1, true);
    else S2D_ROW_REDUCE(true, 1, false);
#undef S2D_ROW_REDUCE
    *plan_out = p;
    return S2D_OK;
}

extern "C" int s2d_bnrow_stats_bf16(const void *x, int64_t n, int c, float *stats, int write_count, void *ws, size_t ws_bytes,
                                    s2d_stream_t stream) {
    S2D_CHECK_ARG(stats, "bnrow_stats: null stats");
    hipStream_t st = (hipStream_t)stream;
    RedPlan p;
    int rc = bnrow_reduce(false, x, nullptr, nullptr, nullptr, nullptr, 0, n, c, ws, ws_bytes, st, &p, "bnrow_stats");
    if (rc) return rc;
    hipLaunchKernelGGL(partial_sum_kernel, dim3((2 * c + 3) / 4), dim3(256), 0, st, (const float *)ws, p.nblocks, 2 * c, stats,
                       (float *)nullptr, write_count ? (float)n : -1.f, true);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_bnrow_stats_finalize_bf16(const void *x, int64_t n, int c, const float *gamma, const float *beta, float eps,
                                             float momentum, float *mean, float *invstd, float *scale, float *shift,
                                             float *running_mean, float *running_var, int64_t *batches_tracked, void *ws,
                                             size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(gamma && beta && mean && invstd && scale && shift, "bnrow_stats_finalize: bad argument");
    hipStream_t st = (hipStream_t)stream;
    RedPlan p;
    int rc = bnrow_reduce(false, x, nullptr, nullptr, nullptr, nullptr, 0, n, c, ws, ws_bytes, st, &p, "bnrow_stats_finalize");
    if (rc) return rc;
    hipLaunchKernelGGL(bn_reduce_finalize_fwd_kernel, dim3(c), dim3(256), 0, st, (const float *)ws, p.nblocks, (float)n,
                       gamma, beta, eps, momentum, c, mean, invstd, scale, shift, running_mean, running_var,
                       (long long *)batches_tracked, true);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

/* y_ld: row stride of y in elements (a multiple of 8, >= c): y may be a channel slice of a wider row-major tensor */
extern "C" int s2d_bnrow_apply_ld_bf16(const void *x, const float *scale, const float *shift, const void *residual, int relu,
                                       int64_t n, int c, void *y, int y_ld, s2d_stream_t stream);
extern "C" int s2d_bnrow_apply_bf16(const void *x, const float *scale, const float *shift, const void *residual, int relu,
                                    int64_t n, int c, void *y, s2d_stream_t stream) {
    return s2d_bnrow_apply_ld_bf16(x, scale, shift, residual, relu, n, c, y, c, stream);
}
extern "C" int s2d_bnrow_apply_ld_bf16(const void *x, const float *scale, const float *shift, const void *residual, int relu,
                                       int64_t n, int c, void *y, int y_ld, s2d_stream_t stream) {
    int rc = check_c8(c, "bnrow_apply");
    if (rc) return rc;
    S2D_CHECK_ARG(n > 0 && x && y && scale && shift && y_ld >= c && y_ld % 8 == 0, "bnrow_apply: bad argument");
    const RowLaunch l = row_launch(n, c / 8);
#define S2D_ROW_APPLY(RS, RL)                                                                                                   \
    hipLaunchKernelGGL((row_apply_bf16_kernel<RS, RL>), dim3(l.blocks), dim3(l.threads), 0, (hipStream_t)stream, (const __bf16 *)x, \
                       scale, shift, (const __bf16 *)residual, n, c / 8, (__bf16 *)y, y_ld / 8)
    S2D_CHECK_ARG(relu >= 0 && relu <= 2, "bnrow_apply: activation code must be 0 (none), 1 (ReLU) or 2 (GELU)");
    if (residual) {
        if (relu == 2) S2D_ROW_APPLY(true, 2); else if (relu) S2D_ROW_APPLY(true, 1); else S2D_ROW_APPLY(true, 0);
    } else {
        if (relu == 2) S2D_ROW_APPLY(false, 2); else if (relu) S2D_ROW_APPLY(false, 1); else S2D_ROW_APPLY(false, 0);
    }
#undef S2D_ROW_APPLY
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

/* the _ld variants: dy_ld = row stride of dy in elements (a multiple of 8, >= c): dy may be a channel slice of a wider row-major tensor */
extern "C" int s2d_bnrow_bwd_reduce_ld_bf16(const void *dy, int dy_ld, const void *x, const void *y, const float *scale, const float *shift,
                                            int relu, int64_t n, int c, float *sums, float *sums_copy, void *ws, size_t ws_bytes,
                                            s2d_stream_t stream);
extern "C" int s2d_bnrow_bwd_reduce_bf16(const void *dy, const void *x, const void *y, const float *scale, const float *shift,
                                         int relu, int64_t n, int c, float *sums, float *sums_copy, void *ws, size_t ws_bytes,
                                         s2d_stream_t stream) {
    return s2d_bnrow_bwd_reduce_ld_bf16(dy, c, x, y, scale, shift, relu, n, c, sums, sums_copy, ws, ws_bytes, stream);
}
extern "C" int s2d_bnrow_bwd_reduce_ld_bf16(const void *dy, int dy_ld, const void *x, const void *y, const float *scale, const float *shift,
                                            int relu, int64_t n, int c, float *sums, float *sums_copy, void *ws, size_t ws_bytes,
                                            s2d_stream_t stream) {
    S2D_CHECK_ARG(sums, "bnrow_bwd_reduce: null sums");
    hipStream_t st = (hipStream_t)stream;
    RedPlan p;
    int rc = bnrow_reduce(true, x, dy, y, scale, shift, relu, n, c, ws, ws_bytes, st, &p, "bnrow_bwd_reduce", dy_ld);
    if (rc) return rc;
    hipLaunchKernelGGL(partial_sum_kernel, dim3((2 * c + 3) / 4), dim3(256), 0, st, (const float *)ws, p.nblocks, 2 * c, sums,
                       sums_copy, -1.f, true);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_bnrow_bwd_reduce_finalize_ld_bf16(const void *dy, int dy_ld, const void *x, const void *y, const float *scale,
                                                     const float *shift, int relu, int64_t n, int c, const float *gamma, const float *mean,
                                                     const float *invstd, float *dgamma, float *dbeta, float *a, float *b, float *d, void *ws,
                                                     size_t ws_bytes, s2d_stream_t stream);
extern "C" int s2d_bnrow_bwd_reduce_finalize_bf16(const void *dy, const void *x, const void *y, const float *scale,
                                                  const float *shift, int relu, int64_t n, int c, const float *gamma, const float *mean, const float *invstd,
                                                  float *dgamma, float *dbeta, float *a, float *b, float *d, void *ws,
                                                  size_t ws_bytes, s2d_stream_t stream) {
    return s2d_bnrow_bwd_reduce_finalize_ld_bf16(dy, c, x, y, scale, shift, relu, n, c, gamma, mean, invstd, dgamma, dbeta, a, b, d, ws, ws_bytes,
                                                 stream);
}
extern "C" int s2d_bnrow_bwd_reduce_finalize_ld_bf16(const void *dy, int dy_ld, const void *x, const void *y, const float *scale,
                                                     const float *shift, int relu, int64_t n, int c, const float *gamma, const float *mean,
                                                     const float *invstd, float *dgamma, float *dbeta, float *a, float *b, float *d, void *ws,
                                                     size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(gamma && mean && invstd && dgamma && dbeta && a && b && d, "bnrow_bwd_reduce_finalize: bad argument");
    hipStream_t st = (hipStream_t)stream;
    RedPlan p;
    int rc = bnrow_reduce(true, x, dy, y, scale, shift, relu, n, c, ws, ws_bytes, st, &p, "bnrow_bwd_reduce_finalize", dy_ld);
    if (rc) return rc;
    hipLaunchKernelGGL(bn_reduce_finalize_bwd_kernel, dim3(c), dim3(256), 0, st, (const float *)ws, p.nblocks, (float)n,
                       gamma, mean, invstd, c, dgamma, dbeta, a, b, d, true);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_bnrow_bwd_apply_ld_bf16(const void *dy, int dy_ld, const void *x, const void *y, const float *scale, const float *shift,
                                           int relu, const float *a, const float *b, const float *d, int64_t n, int c, void *dx,
                                           void *dres, s2d_stream_t stream);
extern "C" int s2d_bnrow_bwd_apply_bf16(const void *dy, const void *x, const void *y, const float *scale, const float *shift,
                                        int relu, const float *a, const float *b, const float *d, int64_t n, int c, void *dx,
                                        void *dres, s2d_stream_t stream) {
    return s2d_bnrow_bwd_apply_ld_bf16(dy, c, x, y, scale, shift, relu, a, b, d, n, c, dx, dres, stream);
}
extern "C" int s2d_bnrow_bwd_apply_ld_bf16(const void *dy, int dy_ld, const void *x, const void *y, const float *scale, const float *shift,
                                           int relu, const float *a, const float *b, const float *d, int64_t n, int c, void *dx,
                                           void *dres, s2d_stream_t stream) {
    int rc = check_c8(c, "bnrow_bwd_apply");
    if (rc) return rc;
    S2D_CHECK_ARG(n > 0 && dy && x && dx && a && b && d && (!relu || (y && relu == 1) || (scale && shift)) && relu >= 0 && relu <= 2 &&
                      dy_ld >= c && dy_ld % 8 == 0, "bnrow_bwd_apply: bad argument");
    const RowLaunch l = row_launch(n, c / 8);
#define S2D_ROW_BWD(RL, Y, DR)                                                                                                     \
    hipLaunchKernelGGL((row_bwd_apply_bf16_kernel<RL, Y, DR>), dim3(l.blocks), dim3(l.threads), 0, (hipStream_t)stream,         \
                       (const __bf16 *)dy, (const __bf16 *)x, (const __bf16 *)y, scale, shift, a, b, d, n, c / 8, (__bf16 *)dx,    \
                       (__bf16 *)dres, dy_ld / 8)
    const bool use_y = relu == 1 && y;
    if (!relu) {
        if (dres) S2D_ROW_BWD(0, false, true); else S2D_ROW_BWD(0, false, false);
    } else if (relu == 2) {
        if (dres) S2D_ROW_BWD(2, false, true); else S2D_ROW_BWD(2, false, false);
    } else if (use_y) {
        if (dres) S2D_ROW_BWD(1, true, true); else S2D_ROW_BWD(1, true, false);
    } else {
        if (dres) S2D_ROW_BWD(1, false, true); else S2D_ROW_BWD(1, false, false);
    }
#undef S2D_ROW_BWD
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

// ---- batch-norm statistics whose reduction pass ran in a producer's epilogue -----------------------------------------
constexpr int PS_LONG = 1536, PS_SLICES = 512;   // partial lists longer than PS_LONG rows are folded in two stages

extern "C" size_t s2d_bn_partials_sum_workspace_bytes(int nblocks, int c) {
    if (nblocks <= PS_LONG || c <= 0 || 2 * c > 256) return 0;
    return align_up((size_t)PS_SLICES * 2 * c * sizeof(float), 256);
}

// first stage of a long list: [nblocks][2c] -> [slices][2c] in ws; returns the new row count (or nblocks when not applicable)
static int partials_prefold(const float *&partial, int nblocks, int c, void *ws, size_t ws_bytes, hipStream_t st) {
    const size_t need = s2d_bn_partials_sum_workspace_bytes(nblocks, c);
    if (!need || !ws || ws_bytes < need) return nblocks;
    const int rps = (int)ceil_div(nblocks, PS_SLICES), slices = (int)ceil_div(nblocks, rps);
    hipLaunchKernelGGL(partial_sum_slices_kernel, dim3(slices), dim3(256), 0, st, partial, nblocks, 2 * c, rps, (float *)ws);
    partial = (const float *)ws;
    return slices;
}

/* ws (optional, s2d_bn_partials_sum_workspace_bytes): lists of more than 1536 rows (a sparse conv writes one row per 64-row tile: 4500 rows
 * at the first stage) are folded in two stages - one wave per channel walking 70 x 64 scattered rows took 20-30 us */
extern "C" int s2d_bn_partials_finalize_ws_f32(const float *partial, int nblocks, int64_t n, int c, const float *gamma, const float *beta,
                                               float eps, float momentum, float *mean, float *invstd, float *scale, float *shift,
                                               float *running_mean, float *running_var, int64_t *batches_tracked, void *ws, size_t ws_bytes,
                                               s2d_stream_t stream) {
    S2D_CHECK_ARG(partial && nblocks > 0 && n > 0 && c > 0 && gamma && beta && mean && invstd && scale && shift,
                  "bn_partials_finalize: bad argument");
    hipStream_t st = (hipStream_t)stream;
    nblocks = partials_prefold(partial, nblocks, c, ws, ws_bytes, st);
    hipLaunchKernelGGL(bn_reduce_finalize_fwd_kernel, dim3(c), dim3(256), 0, st, partial, nblocks, (float)n, gamma, beta, eps,
                       momentum, c, mean, invstd, scale, shift, running_mean, running_var, (long long *)batches_tracked);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

/* Backward counterpart: partial[nblocks][2][c] = per-tile (sum g, sum g x) written by the epilogue of the data-gradient conv that produced
 * this batch norm's output gradient (s2d_conv2d3x3_nhwc_bf16_bnbwd / s2d_conv2d1x1_nhwc_bf16_bnbwd) -> dgamma, dbeta and the per-channel
 * coefficients a, b, d of s2d_bnrow_bwd_apply_ld_bf16: the reduction pass over (dY, x) of s2d_bnrow_bwd_reduce_finalize_ld_bf16 is gone. */
extern "C" int s2d_bn_partials_bwd_finalize_ws_f32(const float *partial, int nblocks, int64_t n, int c, const float *gamma, const float *mean,
                                                   const float *invstd, float *dgamma, float *dbeta, float *a, float *b, float *d, void *ws,
                                                   size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(partial && nblocks > 0 && n > 0 && c > 0 && gamma && mean && invstd && dgamma && dbeta && a && b && d,
                  "bn_partials_bwd_finalize: bad argument");
    hipStream_t st = (hipStream_t)stream;
    nblocks = partials_prefold(partial, nblocks, c, ws, ws_bytes, st);
    hipLaunchKernelGGL(bn_reduce_finalize_bwd_kernel, dim3(c), dim3(256), 0, st, partial, nblocks, (float)n, gamma, mean, invstd, c, dgamma,
                       dbeta, a, b, d, false);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_bn_partials_finalize_f32(const float *partial, int nblocks, int64_t n, int c, const float *gamma, const float *beta,
                                            float eps, float momentum, float *mean, float *invstd, float *scale, float *shift,
                                            float *running_mean, float *running_var, int64_t *batches_tracked,
                                            s2d_stream_t stream) {
    return s2d_bn_partials_finalize_ws_f32(partial, nblocks, n, c, gamma, beta, eps, momentum, mean, invstd, scale, shift, running_mean,
                                           running_var, batches_tracked, nullptr, 0, stream);
}



/* stats[2c] (+ count at [2c] when write_count) = column sums of partial[nblocks][2c].  ws (optional, s2d_bn_partials_sum_workspace_bytes):
 * lets long lists (one row per producer tile) be folded in two stages */
extern "C" int s2d_bn_partials_sum_ws_f32(const float *partial, int nblocks, int64_t n, int c, float *stats, int write_count, void *ws,
                                          size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(partial && nblocks > 0 && n > 0 && c > 0 && stats, "bn_partials_sum: bad argument");
    hipStream_t st = (hipStream_t)stream;
    nblocks = partials_prefold(partial, nblocks, c, ws, ws_bytes, st);
    hipLaunchKernelGGL(partial_sum_kernel, dim3((2 * c + 3) / 4), dim3(256), 0, st, partial, nblocks, 2 * c, stats, (float *)nullptr,
                       write_count ? (float)n : -1.f);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_bn_partials_sum_f32(const float *partial, int nblocks, int64_t n, int c, float *stats, int write_count,
                                       s2d_stream_t stream) {
    return s2d_bn_partials_sum_ws_f32(partial, nblocks, n, c, stats, write_count, nullptr, 0, stream);
}
