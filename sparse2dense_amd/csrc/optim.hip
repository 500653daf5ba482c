// Fused multi-tensor Adam step with decoupled ("true") weight decay and the gradient-clip coefficient folded in: the update of
// the reference's training step (det3d/torchie/apis/train.py:168-186 build_one_cycle_optimizer -> fastai OptimWrapper,
// det3d/solver/fastai_optim.py:158-171: p *= 1 - wd*lr for every parameter, then torch.optim.Adam(betas=(mom, 0.99)) with
// weight_decay = 0; gradient clipping clip_grad_norm_(35) in hooks/optimizer.py:15-21).
//   g     = grad * clip_coef                      (clip_coef = min(1, max_norm / (total_norm + 1e-6)), a DEVICE scalar: no host sync)
//   p     = p * (1 - lr * wd)
//   m     = beta1 * m + (1 - beta1) * g ;  v = beta2 * v + (1 - beta2) * g * g
//   p     = p - lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// One launch handles up to ADAM_MAX_TENSORS tensors (pointer table in the kernel arguments).
#include "s2d_common.h"

namespace s2d {

constexpr int ADAM_MAX_TENSORS = 48;
constexpr int ADAM_CHUNK = 256 * 4 * 4;   // elements per block: 256 threads x 4 float4

struct AdamTable {
    float *p[ADAM_MAX_TENSORS];
    const float *g[ADAM_MAX_TENSORS];
    float *m[ADAM_MAX_TENSORS];
    float *v[ADAM_MAX_TENSORS];
    int64_t n[ADAM_MAX_TENSORS];
    int first_block[ADAM_MAX_TENSORS + 1];
    int count;
};

__global__ __launch_bounds__(256) void adam_step_kernel(AdamTable tb, float lr, float beta1, float beta2, float eps, float wd, float bc1,
                                                        float bc2_sqrt, const float *__restrict__ clip_coef) {
    int ti = 0;
    while (ti + 1 < tb.count && (int)blockIdx.x >= tb.first_block[ti + 1]) ++ti;   // block-uniform scan of <= 48 entries
    const int64_t base = (int64_t)(blockIdx.x - tb.first_block[ti]) * ADAM_CHUNK;
    float *p = tb.p[ti], *m = tb.m[ti], *v = tb.v[ti];
    const float *g = tb.g[ti];
    const int64_t n = tb.n[ti];
    const float cc = clip_coef ? clip_coef[0] : 1.f;
    const float decay = 1.f - lr * wd, step = lr / bc1;
    for (int j = 0; j < 4; ++j) {
        const int64_t i = base + ((int64_t)j * 256 + threadIdx.x) * 4;
        if (i + 3 < n && ((reinterpret_cast<uintptr_t>(p + i) | reinterpret_cast<uintptr_t>(g + i) | reinterpret_cast<uintptr_t>(m + i) |
                           reinterpret_cast<uintptr_t>(v + i)) & 15) == 0) {
            float4 pp = *reinterpret_cast<float4 *>(p + i), mm = *reinterpret_cast<float4 *>(m + i), vv = *reinterpret_cast<float4 *>(v + i);
            const float4 gg = *reinterpret_cast<const float4 *>(g + i);
            float *pf = &pp.x, *mf = &mm.x, *vf = &vv.x;
            const float *gf = &gg.x;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ge = gf[e] * cc;
                mf[e] = beta1 * mf[e] + (1.f - beta1) * ge;
                vf[e] = beta2 * vf[e] + (1.f - beta2) * ge * ge;
                pf[e] = pf[e] * decay - step * mf[e] / (sqrtf(vf[e]) / bc2_sqrt + eps);
            }
            *reinterpret_cast<float4 *>(p + i) = pp;
            *reinterpret_cast<float4 *>(m + i) = mm;
            *reinterpret_cast<float4 *>(v + i) = vv;
        } else {
            for (int e = 0; e < 4; ++e) {
                if (i + e >= n) break;
                const float ge = g[i + e] * cc;
                const float me = beta1 * m[i + e] + (1.f - beta1) * ge;
                const float ve = beta2 * v[i + e] + (1.f - beta2) * ge * ge;
                m[i + e] = me;
                v[i + e] = ve;
                p[i + e] = p[i + e] * decay - step * me / (sqrtf(ve) / bc2_sqrt + eps);
            }
        }
    }
}

// sum of squares of up to ADAM_MAX_TENSORS gradients -> partial[block]; second kernel: total norm and the clip coefficient
struct NormTable {
    const float *g[ADAM_MAX_TENSORS];
    int64_t n[ADAM_MAX_TENSORS];
    int first_block[ADAM_MAX_TENSORS + 1];
    int count;
};

__global__ __launch_bounds__(256) void grad_sumsq_kernel(NormTable tb, float *__restrict__ partial) {
    __shared__ float red[4];
    int ti = 0;
    while (ti + 1 < tb.count && (int)blockIdx.x >= tb.first_block[ti + 1]) ++ti;
    const int64_t base = (int64_t)(blockIdx.x - tb.first_block[ti]) * ADAM_CHUNK;
    const float *g = tb.g[ti];
    const int64_t n = tb.n[ti];
    float s = 0.f;
    for (int j = 0; j < 16; ++j) {
        const int64_t i = base + (int64_t)j * 256 + threadIdx.x;
        if (i < n) { const float x = g[i]; s += x * x; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = total L2 norm, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))   (torch.nn.utils.clip_grad_norm_)
__global__ __launch_bounds__(256) void grad_norm_finalize_kernel(const float *__restrict__ partial, int n, float max_norm, float *__restrict__ out) {
    __shared__ double red[256];
    double s = 0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(red[0]);
        out[0] = norm;
        const float c = max_norm / (norm + 1e-6f);
        out[1] = c < 1.f ? c : 1.f;
    }
}

}  // namespace s2d

using namespace s2d;

extern "C" int s2d_adam_max_tensors(void) { return ADAM_MAX_TENSORS; }

extern "C" int s2d_adam_step_f32(int count, float *const *params, const float *const *grads, float *const *exp_avg, float *const *exp_avg_sq,
                                 const int64_t *numel, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                 const float *clip_coef, s2d_stream_t stream) {
    S2D_CHECK_ARG(count >= 0 && count <= ADAM_MAX_TENSORS && step >= 1, "adam_step: bad count / step");
    if (count == 0) return S2D_OK;
    S2D_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && numel, "adam_step: null table");
    AdamTable tb;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        S2D_CHECK_ARG(params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i] && numel[i] >= 0, "adam_step: null tensor %d", i);
        tb.p[i] = params[i]; tb.g[i] = grads[i]; tb.m[i] = exp_avg[i]; tb.v[i] = exp_avg_sq[i]; tb.n[i] = numel[i];
        tb.first_block[i] = blocks;
        blocks += (int)ceil_div(numel[i], ADAM_CHUNK);
    }
    tb.first_block[count] = blocks;
    tb.count = count;
    if (blocks == 0) return S2D_OK;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_step_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tb, lr, beta1, beta2, eps, weight_decay, bc1,
                       sqrtf(bc2), clip_coef);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" size_t s2d_grad_norm_workspace_floats(int count, const int64_t *numel) {
    size_t blocks = 0;
    for (int i = 0; i < count; ++i) blocks += (size_t)ceil_div(numel[i], ADAM_CHUNK);
    return blocks + 8;
}

/* sum of squares of `count` gradients accumulated into partial[offset ...]; returns the number of partials written through *written */
extern "C" int s2d_grad_sumsq_f32(int count, const float *const *grads, const int64_t *numel, float *partial, int *written, s2d_stream_t stream) {
    S2D_CHECK_ARG(count >= 0 && count <= ADAM_MAX_TENSORS && written, "grad_sumsq: bad count");
    *written = 0;
    if (count == 0) return S2D_OK;
    S2D_CHECK_ARG(grads && numel && partial, "grad_sumsq: null argument");
    NormTable tb;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        tb.g[i] = grads[i]; tb.n[i] = numel[i];
        tb.first_block[i] = blocks;
        blocks += (int)ceil_div(numel[i], ADAM_CHUNK);
    }
    tb.first_block[count] = blocks;
    tb.count = count;
    *written = blocks;
    if (blocks == 0) return S2D_OK;
    hipLaunchKernelGGL(grad_sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tb, partial);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_grad_norm_finalize_f32(const float *partial, int n, float max_norm, float *out2, s2d_stream_t stream) {
    S2D_CHECK_ARG(partial && out2 && n >= 0, "grad_norm_finalize: bad argument");
    hipLaunchKernelGGL(grad_norm_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n, max_norm, out2);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
