// CenterHead losses on the device maps, one pass per direction:
//   FastFocalLoss (/root/reference/det3d/models/losses/centernet_loss.py:33-54): on the clamped sigmoid map `out` [B,C,H,W]
//       neg = sum log(1-out) out^2 (1-target)^4 over every pixel,  pos = sum_m log(p_m) (1-p_m)^2 mask_m with p_m = out[b, cat_m, ind_m],
//       loss = -(pos + neg) / max(sum mask, 1)          (the reference's num_pos == 0 branch gives the same value)
//   RegLoss (centernet_loss.py:9-31): pred = out[b, :, ind_m]; loss_c = sum_{b,m} |pred*mask - target*mask| / (sum mask + 1e-4)
// composed from torch ops these are ~45 + ~20 launches of 3-5 us per step (permute / gather / pow / log / mul / sum chains and their
// backward); here 2 + 2 and 1 + 2.  Sums are folded in a fixed order; the backward's scatter uses atomicAdd only where two objects share
// a centre cell (as torch's gather backward does).
#include "s2d_common.h"

namespace s2d {

constexpr int FOCAL_BLOCKS = 256;

__device__ __forceinline__ float block_sum_256(float v, float *sh) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wid] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void focal_neg_kernel(const float *__restrict__ out, const float *__restrict__ target, int64_t n,
                                                        float *__restrict__ partial) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float p = out[i], t = 1.f - target[i];
        const float t2 = t * t;
        s += logf(1.f - p) * p * p * (t2 * t2);
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// one block: positives + fold of the negative partials.  res = {loss, pos, neg, num_pos}
__global__ __launch_bounds__(256) void focal_finalize_kernel(const float *__restrict__ out, const int64_t *__restrict__ ind, const uint8_t *__restrict__ mask,
                                                             const int64_t *__restrict__ cat, int batch, int classes, int64_t hw, int max_objs,
                                                             const float *__restrict__ partial, int n_partial, float *__restrict__ res) {
    __shared__ float sh[4];
    float pos = 0.f, cnt = 0.f, neg = 0.f;
    for (int i = threadIdx.x; i < batch * max_objs; i += 256) {
        if (!mask[i]) continue;
        const int b = i / max_objs;
        const float p = out[((int64_t)b * classes + cat[i]) * hw + ind[i]];
        pos += logf(p) * (1.f - p) * (1.f - p);
        cnt += 1.f;
    }
    for (int i = threadIdx.x; i < n_partial; i += 256) neg += partial[i];
    pos = block_sum_256(pos, sh);
    cnt = block_sum_256(cnt, sh);
    neg = block_sum_256(neg, sh);
    if (threadIdx.x == 0) {
        res[0] = -(pos + neg) / fmaxf(cnt, 1.f);
        res[1] = pos;
        res[2] = neg;
        res[3] = cnt;
    }
}

// d loss / d out over every pixel (negatives)
__global__ __launch_bounds__(256) void focal_bwd_neg_kernel(const float *__restrict__ out, const float *__restrict__ target, int64_t n,
                                                            const float *__restrict__ res, const float *__restrict__ go, float *__restrict__ dout) {
    const float coef = -go[0] / fmaxf(res[3], 1.f);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float p = out[i], t = 1.f - target[i];
        const float t2 = t * t, q = 1.f - p;
        dout[i] = coef * (t2 * t2) * (2.f * p * logf(q) - p * p / q);
    }
}

__global__ __launch_bounds__(256) void focal_bwd_pos_kernel(const float *__restrict__ out, const int64_t *__restrict__ ind, const uint8_t *__restrict__ mask,
                                                            const int64_t *__restrict__ cat, int batch, int classes, int64_t hw, int max_objs,
                                                            const float *__restrict__ res, const float *__restrict__ go, float *__restrict__ dout) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= batch * max_objs || !mask[i]) return;
    const int b = i / max_objs;
    const int64_t at = ((int64_t)b * classes + cat[i]) * hw + ind[i];
    const float p = out[at], q = 1.f - p;
    const float coef = -go[0] / fmaxf(res[3], 1.f);
    atomicAdd(dout + at, coef * (q * q / p - 2.f * q * logf(p)));
}

// res[c] = loss_c (c < channels), res[channels] = sum mask + 1e-4
__global__ __launch_bounds__(256) void regloss_fwd_kernel(const float *__restrict__ feat, const int64_t *__restrict__ ind, const uint8_t *__restrict__ mask,
                                                          const float *__restrict__ target, int batch, int channels, int64_t hw, int max_objs,
                                                          float *__restrict__ res) {
    __shared__ float sh[4];
    float cnt = 0.f;
    for (int i = threadIdx.x; i < batch * max_objs; i += 256) cnt += mask[i] ? 1.f : 0.f;
    cnt = block_sum_256(cnt, sh) + 1e-4f;
    for (int c = 0; c < channels; ++c) {
        float s = 0.f;
        for (int i = threadIdx.x; i < batch * max_objs; i += 256) {
            if (!mask[i]) continue;
            const int b = i / max_objs;
            s += fabsf(feat[((int64_t)b * channels + c) * hw + ind[i]] - target[(int64_t)i * channels + c]);
        }
        s = block_sum_256(s, sh);
        if (threadIdx.x == 0) res[c] = s / cnt;
    }
    if (threadIdx.x == 0) res[channels] = cnt;
}

__global__ __launch_bounds__(256) void regloss_bwd_kernel(const float *__restrict__ feat, const int64_t *__restrict__ ind, const uint8_t *__restrict__ mask,
                                                          const float *__restrict__ target, int batch, int channels, int64_t hw, int max_objs,
                                                          const float *__restrict__ res, const float *__restrict__ go, float *__restrict__ dfeat) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= batch * max_objs * channels) return;
    const int c = t % channels, i = t / channels;
    if (!mask[i]) return;
    const int b = i / max_objs;
    const int64_t at = ((int64_t)b * channels + c) * hw + ind[i];
    const float d = feat[at] - target[(int64_t)i * channels + c];
    const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    atomicAdd(dfeat + at, go[c] * sgn / res[channels]);
}

}  // namespace s2d

using namespace s2d;

extern "C" size_t s2d_focal_workspace_bytes(void) { return (size_t)FOCAL_BLOCKS * sizeof(float) + 256; }

/* out / target: fp32 [batch][classes][hw] contiguous; ind, cat: int64 [batch][max_objs]; mask: uint8 [batch][max_objs];
 * res (device, 4 floats): loss, positive sum, negative sum, number of positives */
extern "C" int s2d_focal_fwd(const float *out, const float *target, const int64_t *ind, const uint8_t *mask, const int64_t *cat, int batch,
                             int classes, int64_t hw, int max_objs, float *res, void *ws, size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(out && target && ind && mask && cat && res && batch > 0 && classes > 0 && hw > 0 && max_objs > 0, "focal_fwd: bad argument");
    if (!ws || ws_bytes < s2d_focal_workspace_bytes()) {
        set_error("focal_fwd: workspace too small");
        return S2D_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)batch * classes * hw;
    const int nb = (int)std::min<int64_t>(FOCAL_BLOCKS, ceil_div(n, 256));
    hipLaunchKernelGGL(focal_neg_kernel, dim3(nb), dim3(256), 0, st, out, target, n, (float *)ws);
    hipLaunchKernelGGL(focal_finalize_kernel, dim3(1), dim3(256), 0, st, out, ind, mask, cat, batch, classes, hw, max_objs, (const float *)ws, nb,
                       res);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

/* dout (fp32, same shape as out) = go[0] * d loss / d out; res = the forward's result vector */
extern "C" int s2d_focal_bwd(const float *out, const float *target, const int64_t *ind, const uint8_t *mask, const int64_t *cat, int batch,
                             int classes, int64_t hw, int max_objs, const float *res, const float *go, float *dout, s2d_stream_t stream) {
    S2D_CHECK_ARG(out && target && ind && mask && cat && res && go && dout && batch > 0 && classes > 0 && hw > 0 && max_objs > 0,
                  "focal_bwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = (int64_t)batch * classes * hw;
    hipLaunchKernelGGL(focal_bwd_neg_kernel, dim3((unsigned)std::min<int64_t>(2048, ceil_div(n, 256))), dim3(256), 0, st, out, target, n, res, go,
                       dout);
    hipLaunchKernelGGL(focal_bwd_pos_kernel, dim3((unsigned)ceil_div(batch * max_objs, 256)), dim3(256), 0, st, out, ind, mask, cat, batch, classes,
                       hw, max_objs, res, go, dout);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

/* feat: fp32 [batch][channels][hw]; target: fp32 [batch][max_objs][channels]; res (device, channels + 1 floats): the per-channel
 * losses, then the denominator sum(mask) + 1e-4 */
extern "C" int s2d_regloss_fwd(const float *feat, const int64_t *ind, const uint8_t *mask, const float *target, int batch, int channels,
                               int64_t hw, int max_objs, float *res, s2d_stream_t stream) {
    S2D_CHECK_ARG(feat && ind && mask && target && res && batch > 0 && channels > 0 && hw > 0 && max_objs > 0, "regloss_fwd: bad argument");
    hipLaunchKernelGGL(regloss_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, feat, ind, mask, target, batch, channels, hw, max_objs, res);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

/* dfeat (fp32, zero-filled here) = sum_c go[c] * d loss_c / d feat */
extern "C" int s2d_regloss_bwd(const float *feat, const int64_t *ind, const uint8_t *mask, const float *target, int batch, int channels,
                               int64_t hw, int max_objs, const float *res, const float *go, float *dfeat, s2d_stream_t stream) {
    S2D_CHECK_ARG(feat && ind && mask && target && res && go && dfeat && batch > 0 && channels > 0 && hw > 0 && max_objs > 0,
                  "regloss_bwd: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = zero_async(dfeat, (size_t)batch * channels * hw * sizeof(float), st)) return rc;   // (a kernel, not a memset node: see zero_async)
    hipLaunchKernelGGL(regloss_bwd_kernel, dim3((unsigned)ceil_div(batch * max_objs * channels, 256)), dim3(256), 0, st, feat, ind, mask, target,
                       batch, channels, hw, max_objs, res, go, dfeat);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
