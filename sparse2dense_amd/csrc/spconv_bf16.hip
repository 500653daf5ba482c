// bf16-input / fp32-accumulate variant of the output-stationary sparse-conv implicit GEMM
// (v_mfma_f32_16x16x32_bf16: 16x the rate of the fp32 MFMA).  Features stay fp32 in HBM (the rest
// of the stack — BN statistics, residuals, weight gradients — is fp32); a lane gathers its 8
// consecutive fp32 channels of a neighbour row with two float4 loads and packs them to one bf16x8
// A fragment in registers (v_cvt_pk_bf16_f32).  Weights are pre-packed once per call into the exact
// LDS image the B fragments are read from ([k][co-block][t][n][q][c][8], 1 KiB contiguous per
// (t,n) -> conflict-free ds_read_b128), so the per-offset staging is a linear global_load_lds copy.
// Software pipeline per offset k: gather(k+1) and stage W[k+1] are issued before the MFMAs of k.
#include "s2d_common.h"

namespace s2d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---- weight pre-pack ---------------------------------------------------------------------------
// w[K][cin][cout] fp32 -> image[K][cout/CT][T][NT][4][16][8] bf16 with
//   ci = 32 t + 8 q + e ,  co = y*CT + c*NT + n   (CT = 16*NT)
__global__ __launch_bounds__(256) void pack_weights_bf16_kernel(const float *__restrict__ w, int kvol, int cin, int cout,
                                                                int nt, int transpose, int flip, __bf16 *__restrict__ out) {
    // (cin, cout) are the dimensions of the PACKED operand; with transpose=1 the source tensor is
    // [K][cout][cin] (data gradient: W^T), with flip=1 offsets are mirrored (SubM transposed map).
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)kvol * cin * cout;
    if (i >= total) return;
    const int ct = 16 * nt, T = cin / 32;
    int64_t r = i;
    const int e = r % 8; r /= 8;
    const int c = r % 16; r /= 16;
    const int q = r % 4; r /= 4;
    const int n = r % nt; r /= nt;
    const int t = r % T; r /= T;
    const int y = r % (cout / ct); r /= (cout / ct);
    const int k = (int)r;
    const int ci = 32 * t + 8 * q + e;
    const int co = y * ct + c * nt + n;
    const int ks = flip ? kvol - 1 - k : k;
    const int64_t src = transpose ? ((int64_t)ks * cout + co) * cin + ci : ((int64_t)ks * cin + ci) * cout + co;
    out[i] = (__bf16)w[src];
}

template <int BYTES>
__device__ __forceinline__ void stage_linear(char *lds, const char *__restrict__ src, int wid, int lane) {
    constexpr int UNITS = BYTES / 1024;  // wave-instructions of 64 lanes x 16 B
#pragma unroll
    for (int u = 0; u < (UNITS + 3) / 4; ++u) {
        const int unit = u * 4 + wid;
        if (unit < UNITS) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + unit * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(lds + unit * 1024), 16, 0, 0);
        }
    }
}

__device__ __forceinline__ bf16x8 pack8(const float4 &lo, const float4 &hi) {
    bf16x8 v;
    v[0] = (__bf16)lo.x; v[1] = (__bf16)lo.y; v[2] = (__bf16)lo.z; v[3] = (__bf16)lo.w;
    v[4] = (__bf16)hi.x; v[5] = (__bf16)hi.y; v[6] = (__bf16)hi.z; v[7] = (__bf16)hi.w;
    return v;
}

template <int CIN, int NT, int MT>
__global__ __launch_bounds__(256) void spconv_fwd_bf16(const float *__restrict__ in, const __bf16 *__restrict__ wpack,
                                                       const float *__restrict__ bias, const int32_t *__restrict__ nbr,
                                                       int n_out, int kvol, int cout, float *__restrict__ out) {
    constexpr int CT = 16 * NT;
    constexpr int T = CIN / 32;               // K chunks of 32 per offset
    constexpr int SLAB = CIN * CT * 2;        // bytes of one offset's weight image for this block
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *wbuf0 = smem, *wbuf1 = smem + SLAB;

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int ytiles = cout / CT;
    const int co0 = blockIdx.y * CT;
    const int row0 = (blockIdx.x * 4 + wid) * (16 * MT);
    const char *wsrc = reinterpret_cast<const char *>(wpack) + (int64_t)blockIdx.y * SLAB;
    const int64_t wstride = (int64_t)ytiles * SLAB;  // bytes between consecutive offsets

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: W[0] -> LDS, gather indices and rows of offset 0
    stage_linear<SLAB>(wbuf0, wsrc, wid, lane);
    int j[MT];
    float4 raw[MT][T][2];
    bool any_cur;
    {
        bool mine = false;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int row = row0 + 16 * m + r;
            j[m] = row < n_out ? nbr[row] : -1;
            mine = mine || (j[m] >= 0);
        }
        any_cur = __ballot(mine) != 0ull;
        if (any_cur) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float4 *src = reinterpret_cast<const float4 *>(in + (int64_t)(j[m] >= 0 ? j[m] : 0) * CIN) + 2 * q;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    raw[m][t][0] = j[m] >= 0 ? src[8 * t] : float4{0.f, 0.f, 0.f, 0.f};
                    raw[m][t][1] = j[m] >= 0 ? src[8 * t + 1] : float4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    }

    for (int k = 0; k < kvol; ++k) {
        char *wb = (k & 1) ? wbuf1 : wbuf0;
        // A fragments of offset k (bf16) from the rows gathered one iteration ago
        bf16x8 a[MT][T];
        if (any_cur) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int t = 0; t < T; ++t) a[m][t] = pack8(raw[m][t][0], raw[m][t][1]);
        }
        // issue next offset's weight staging and row gather before this offset's MFMAs
        bool any_next = false;
        if (k + 1 < kvol) {
            stage_linear<SLAB>((k & 1) ? wbuf0 : wbuf1, wsrc + (int64_t)(k + 1) * wstride, wid, lane);
            bool mine = false;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int row = row0 + 16 * m + r;
                j[m] = row < n_out ? nbr[(int64_t)(k + 1) * n_out + row] : -1;
                mine = mine || (j[m] >= 0);
            }
            any_next = __ballot(mine) != 0ull;
            if (any_next) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float4 *src = reinterpret_cast<const float4 *>(in + (int64_t)(j[m] >= 0 ? j[m] : 0) * CIN) + 2 * q;
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        raw[m][t][0] = j[m] >= 0 ? src[8 * t] : float4{0.f, 0.f, 0.f, 0.f};
                        raw[m][t][1] = j[m] >= 0 ? src[8 * t + 1] : float4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
        }
        if (k == 0) __syncthreads();  // W[0] landed
        if (any_cur) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const bf16x8 b = *reinterpret_cast<const bf16x8 *>(wb + ((t * NT + n) * 64 + lane) * 16);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][t], b, acc[m][n], 0, 0, 0);
                }
            }
        }
        __syncthreads();  // done with wb; W[k+1] landed (barrier release drains the LDS-DMA)
        any_cur = any_next;
    }

    // C/D layout: column = lane&15 -> channels co0 + r*NT + n, row = (lane>>4)*4 + reg
    float bv[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) bv[n] = bias ? bias[co0 + r * NT + n] : 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = row0 + 16 * m + 4 * q + reg;
            if (row < n_out) {
                float *dst = out + (int64_t)row * cout + co0 + r * NT;
                if (NT == 4) {
                    *reinterpret_cast<float4 *>(dst) = float4{acc[m][0][reg] + bv[0], acc[m][1 % NT][reg] + bv[1 % NT],
                                                             acc[m][2 % NT][reg] + bv[2 % NT], acc[m][3 % NT][reg] + bv[3 % NT]};
                } else if (NT == 2) {
                    *reinterpret_cast<float2 *>(dst) = float2{acc[m][0][reg] + bv[0], acc[m][1 % NT][reg] + bv[1 % NT]};
                } else {
                    dst[0] = acc[m][0][reg] + bv[0];
                }
            }
        }
    }
}

template <int CIN, int NT>
static int launch_bf16(const float *in, const __bf16 *wpack, const float *bias, const int32_t *nbr, int n_out, int kvol,
                       int cout, float *out, hipStream_t st) {
    constexpr int CT = 16 * NT;
    const size_t lds = 2 * (size_t)CIN * CT * 2;
    const int ytiles = cout / CT;
    const bool small = ceil_div(n_out, 128) * ytiles < 512;
    if (small) {
        auto kern = spconv_fwd_bf16<CIN, NT, 1>;
        hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(n_out, 64), ytiles), dim3(256), lds, st, in, wpack, bias, nbr, n_out,
                           kvol, cout, out);
    } else {
        auto kern = spconv_fwd_bf16<CIN, NT, 2>;
        hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(n_out, 128), ytiles), dim3(256), lds, st, in, wpack, bias, nbr, n_out,
                           kvol, cout, out);
    }
    S2D_LAUNCH_CHECK();
    return 0;
}

static int bf16_nt(int cout) { return cout == 16 ? 1 : (cout == 32 ? 2 : 4); }

}  // namespace s2d

using namespace s2d;

extern "C" int s2d_spconv_bf16_supported(int cin, int cout) {
    return (cin == 32 || cin == 64 || cin == 128) && (cout == 16 || cout == 32 || cout == 64 || cout == 128);
}

extern "C" int s2d_spconv_pack_weights_bf16(const float *weight, int kvol, int cin, int cout, int transpose, int flip,
                                            void *packed, s2d_stream_t stream) {
    S2D_CHECK_ARG(weight && packed && kvol > 0, "pack_weights_bf16: null argument");
    if (!s2d_spconv_bf16_supported(cin, cout)) {
        set_error("pack_weights_bf16: unsupported channels %d -> %d", cin, cout);
        return S2D_ERR_UNSUPPORTED;
    }
    const int64_t total = (int64_t)kvol * cin * cout;
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       weight, kvol, cin, cout, bf16_nt(cout), transpose, flip, (__bf16 *)packed);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_spconv_fwd_bf16(const float *in_feat, int64_t n_in, const void *packed_weight, const float *bias,
                                   const int32_t *nbr, int64_t n_out, int kvol, int cin, int cout, float *out_feat,
                                   s2d_stream_t stream) {
    S2D_CHECK_ARG(n_in >= 0 && n_out >= 0 && n_out < 0x7fffffff && kvol > 0, "spconv_fwd_bf16: bad sizes");
    if (!s2d_spconv_bf16_supported(cin, cout)) {
        set_error("spconv_fwd_bf16: unsupported channels %d -> %d", cin, cout);
        return S2D_ERR_UNSUPPORTED;
    }
    if (n_out == 0) return S2D_OK;
    S2D_CHECK_ARG(in_feat && packed_weight && nbr && out_feat && n_in > 0, "spconv_fwd_bf16: null argument");
    hipStream_t st = (hipStream_t)stream;
    const __bf16 *wp = (const __bf16 *)packed_weight;
    const int no = (int)n_out;
#define S2D_BF16_CASE(CI)                                                                                   \
    case CI:                                                                                                \
        switch (bf16_nt(cout)) {                                                                            \
            case 1: return launch_bf16<CI, 1>(in_feat, wp, bias, nbr, no, kvol, cout, out_feat, st);        \
            case 2: return launch_bf16<CI, 2>(in_feat, wp, bias, nbr, no, kvol, cout, out_feat, st);        \
            default: return launch_bf16<CI, 4>(in_feat, wp, bias, nbr, no, kvol, cout, out_feat, st);       \
        }
    switch (cin) {
        S2D_BF16_CASE(32)
        S2D_BF16_CASE(64)
        S2D_BF16_CASE(128)
    }
#undef S2D_BF16_CASE
    return S2D_ERR_UNSUPPORTED;
}
