// Fused PointPillars feature net (one PFN layer, the configuration of configs/waymo/pp/*: 5 point features + 5 decorations -> 64,
// /root/reference/det3d/models/readers/pillar_encoder.py:41-56 PFNLayer.forward, :114-154 PillarFeatureNet.forward):
//   decorate (x - mean_xyz, x - pillar centre) -> Linear(10 -> 64, no bias) -> BatchNorm1d over the P*20 rows -> ReLU -> max over the
//   20 slots
// without the [P,20,10] / [P,20,64] intermediates the layer-by-layer version writes and re-reads (3 x 188 MB per step at 36 k pillars):
// the per-row products are RECOMPUTED from the 400-byte pillar in every pass.
//   forward  pass 1 (training): per-channel sums of h and h^2 over all rows -> per-workgroup partial slabs [blocks][2][64] (the
//                               layout s2d_bn_partials_* folds; SyncBN all-reduces the folded vector)
//            pass 2           : y = relu(h * scale + shift), out[p][c] = max over the 20 slots (empty slots hold h = 0), the slot
//                               index of the maximum is kept (first maximum, like torch.max) for the backward
//   backward one pass         : g = dout at the kept slot where y > 0; per channel sum g, sum g*h (batch-norm backward), and the three
//                               matrices the weight gradient is assembled from once the batch-norm terms (a, b, d) are known:
//                               dW[c][k] = a_c * M1[c][k] + b_c * M2[c][k] + d_c * M3[k],
//                               M1 = sum g_row f_row[k], M2 = sum h_row[c] f_row[k] (valid rows), M3 = sum f_row[k]
// A wave owns a pillar, lane = output channel; the pillar's points are wave-uniform (scalar loads).  fp32 throughout; the decoration
// arithmetic follows the reference's operation order (no contraction: the file is built with -ffp-contract=off).
#include "s2d_common.h"

namespace s2d {

constexpr int PFN_C = 64;      // output channels = lanes
constexpr int PFN_T = 20;      // max slots supported in registers
constexpr int PFN_F = 10;      // 5 point features + 3 cluster offsets + 2 centre offsets
constexpr int PFN_BWD_COLS = (2 + 2 * PFN_F) * PFN_C + PFN_F;   // sum g | sum gh | M1[10][64] | M2[10][64] | M3[10]

struct PfnGeo {
    float vx, vy, x_offset, y_offset;
    int slots;      // T (<= PFN_T)
    int ndim;       // 5
};

// decorated features of slot t of a pillar (wave-uniform); returns false for an empty slot
__device__ __forceinline__ void pfn_pillar_head(const float *__restrict__ vox, int n, const PfnGeo &g, const int4 c, float (&mean)[3], float &cx, float &cy) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int t = 0; t < g.slots; ++t) {   // torch sums all slots (empty ones are zero)
        s0 += vox[t * g.ndim + 0];
        s1 += vox[t * g.ndim + 1];
        s2 += vox[t * g.ndim + 2];
    }
    const float fn = (float)n;
    mean[0] = __fdiv_rn(s0, fn); mean[1] = __fdiv_rn(s1, fn); mean[2] = __fdiv_rn(s2, fn);
    cx = __fadd_rn(__fmul_rn((float)c.w, g.vx), g.x_offset);
    cy = __fadd_rn(__fmul_rn((float)c.z, g.vy), g.y_offset);
}
__device__ __forceinline__ void pfn_slot_feats(const float *__restrict__ p, const float (&mean)[3], float cx, float cy, float (&f)[PFN_F]) {
    f[0] = p[0]; f[1] = p[1]; f[2] = p[2]; f[3] = p[3]; f[4] = p[4];
    f[5] = p[0] - mean[0]; f[6] = p[1] - mean[1]; f[7] = p[2] - mean[2];
    f[8] = p[0] - cx; f[9] = p[1] - cy;
}
__device__ __forceinline__ float pfn_dot(const float (&w)[PFN_F], const float (&f)[PFN_F]) {
    float h = 0.f;
#pragma unroll
    for (int k = 0; k < PFN_F; ++k) h = fmaf(w[k], f[k], h);
    return h;
}

// block-level fold of per-lane (= per-channel) values of the 4 waves, written as one row of `cols` floats
template <int K>
__device__ __forceinline__ void pfn_block_fold(const float (&v)[K], float *__restrict__ out_row, float *lds /*[4][K][64]*/) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) lds[(wid * K + k) * 64 + lane] = v[k];
    __syncthreads();
    for (int e = threadIdx.x; e < K * 64; e += 256) out_row[e] = (lds[e] + lds[K * 64 + e]) + (lds[2 * K * 64 + e] + lds[3 * K * 64 + e]);
    __syncthreads();
}

__global__ __launch_bounds__(256) void pfn_stats_kernel(const float *__restrict__ voxels, const int32_t *__restrict__ num, const int32_t *__restrict__ coors,
                                                        const float *__restrict__ weight, int64_t pillars, PfnGeo g, float *__restrict__ partial) {
    __shared__ float lds[4 * 2 * 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float w[PFN_F];
#pragma unroll
    for (int k = 0; k < PFN_F; ++k) w[k] = weight[lane * PFN_F + k];
    float acc[2] = {0.f, 0.f};
    for (int64_t p = (int64_t)blockIdx.x * 4 + wid; p < pillars; p += (int64_t)gridDim.x * 4) {
        const float *vox = voxels + p * g.slots * g.ndim;
        const int n = min(max(num[p], 0), g.slots);
        if (n == 0) continue;
        float mean[3], cx, cy;
        pfn_pillar_head(vox, n, g, reinterpret_cast<const int4 *>(coors)[p], mean, cx, cy);
        for (int t = 0; t < n; ++t) {
            float f[PFN_F];
            pfn_slot_feats(vox + t * g.ndim, mean, cx, cy, f);
            const float h = pfn_dot(w, f);
            acc[0] += h;
            acc[1] = fmaf(h, h, acc[1]);
        }
    }
    pfn_block_fold<2>(acc, partial + (int64_t)blockIdx.x * 2 * PFN_C, lds);
}

__global__ __launch_bounds__(256) void pfn_apply_max_kernel(const float *__restrict__ voxels, const int32_t *__restrict__ num, const int32_t *__restrict__ coors,
                                                            const float *__restrict__ weight, const float *__restrict__ scale,
                                                            const float *__restrict__ shift, int64_t pillars, PfnGeo g, float *__restrict__ out,
                                                            uint8_t *__restrict__ arg) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float w[PFN_F];
#pragma unroll
    for (int k = 0; k < PFN_F; ++k) w[k] = weight[lane * PFN_F + k];
    const float sc = scale[lane], sh = shift[lane];
    const float y_empty = fmaxf(sh, 0.f);
    for (int64_t p = (int64_t)blockIdx.x * 4 + wid; p < pillars; p += (int64_t)gridDim.x * 4) {
        const float *vox = voxels + p * g.slots * g.ndim;
        const int n = min(max(num[p], 0), g.slots);
        float best = -1.f;   // y >= 0
        int bi = 0;
        if (n > 0) {
            float mean[3], cx, cy;
            pfn_pillar_head(vox, n, g, reinterpret_cast<const int4 *>(coors)[p], mean, cx, cy);
            for (int t = 0; t < n; ++t) {
                float f[PFN_F];
                pfn_slot_feats(vox + t * g.ndim, mean, cx, cy, f);
                const float y = fmaxf(fmaf(pfn_dot(w, f), sc, sh), 0.f);
                if (y > best) { best = y; bi = t; }
            }
        }
        if (n < g.slots && y_empty > best) { best = y_empty; bi = n; }   // first empty slot (torch.max keeps the first maximum)
        out[p * PFN_C + lane] = best;
        if (arg) arg[p * PFN_C + lane] = (uint8_t)bi;
    }
}

__global__ __launch_bounds__(256) void pfn_bwd_kernel(const float *__restrict__ voxels, const int32_t *__restrict__ num, const int32_t *__restrict__ coors,
                                                      const float *__restrict__ weight, const float *__restrict__ dout, const uint8_t *__restrict__ arg,
                                                      int64_t pillars, PfnGeo g, float *__restrict__ partial) {
    __shared__ float lds[4 * (2 + 2 * PFN_F) * 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float w[PFN_F];
#pragma unroll
    for (int k = 0; k < PFN_F; ++k) w[k] = weight[lane * PFN_F + k];
    float acc[2 + 2 * PFN_F];     // sum g, sum g*h, M1[k], M2[k] of this lane's channel
    float m3[PFN_F];              // wave-uniform: sum of f over the valid rows of the wave's pillars
#pragma unroll
    for (int k = 0; k < 2 + 2 * PFN_F; ++k) acc[k] = 0.f;
#pragma unroll
    for (int k = 0; k < PFN_F; ++k) m3[k] = 0.f;
    for (int64_t p = (int64_t)blockIdx.x * 4 + wid; p < pillars; p += (int64_t)gridDim.x * 4) {
        const float *vox = voxels + p * g.slots * g.ndim;
        const int n = min(max(num[p], 0), g.slots);
        const float go = dout[p * PFN_C + lane];     // out > 0 is implied by the gradient being routed: relu'(0) = 0 is applied by the caller
        const int bi = arg[p * PFN_C + lane];
        if (bi >= n) acc[0] += go;                   // the maximum sits in an empty slot: h = 0, f = 0 - only the bias-like sum sees it
        if (n == 0) continue;
        float mean[3], cx, cy;
        pfn_pillar_head(vox, n, g, reinterpret_cast<const int4 *>(coors)[p], mean, cx, cy);
        for (int t = 0; t < n; ++t) {
            float f[PFN_F];
            pfn_slot_feats(vox + t * g.ndim, mean, cx, cy, f);
            const float h = pfn_dot(w, f);
            const float gr = t == bi ? go : 0.f;
            acc[0] += gr;
            acc[1] = fmaf(gr, h, acc[1]);
#pragma unroll
            for (int k = 0; k < PFN_F; ++k) {
                acc[2 + k] = fmaf(gr, f[k], acc[2 + k]);
                acc[2 + PFN_F + k] = fmaf(h, f[k], acc[2 + PFN_F + k]);
                m3[k] += f[k];
            }
        }
    }
    float *row = partial + (int64_t)blockIdx.x * PFN_BWD_COLS;
    pfn_block_fold<2 + 2 * PFN_F>(acc, row, lds);
    // M3: identical in every lane of a wave; lanes 0..9 of the four waves fold their wave's value
    if (lane < PFN_F) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < PFN_F; ++k) v = lane == k ? m3[k] : v;
        lds[wid * 16 + lane] = v;
    }
    __syncthreads();
    if (threadIdx.x < PFN_F) row[(2 + 2 * PFN_F) * PFN_C + threadIdx.x] = (lds[threadIdx.x] + lds[16 + threadIdx.x]) + (lds[32 + threadIdx.x] + lds[48 + threadIdx.x]);
}

static int pfn_blocks(int64_t pillars) {
    const int64_t b = ceil_div(pillars, 4);
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace s2d

using namespace s2d;

extern "C" int s2d_pfn_supported(int ndim, int slots, int feats, int cout) { return ndim == 5 && slots >= 1 && slots <= PFN_T && feats == PFN_F && cout == PFN_C; }
extern "C" int s2d_pfn_blocks(int64_t pillars) { return pfn_blocks(pillars); }
extern "C" int s2d_pfn_bwd_cols(void) { return PFN_BWD_COLS; }

static int pfn_check(const void *voxels, const void *num, const void *coors, const void *weight, int64_t pillars, int slots, int ndim) {
    S2D_CHECK_ARG(voxels && num && coors && weight && pillars > 0, "pfn: null argument / no pillars");
    if (!s2d_pfn_supported(ndim, slots, PFN_F, PFN_C)) {
        set_error("pfn: %d point features x %d slots unsupported", ndim, slots);
        return S2D_ERR_UNSUPPORTED;
    }
    return 0;
}

extern "C" int s2d_pfn_stats_f32(const float *voxels, const int32_t *num_points, const int32_t *coors, const float *weight, int64_t pillars,
                                 int slots, int ndim, float vx, float vy, float x_offset, float y_offset, float *partial, s2d_stream_t stream) {
    int rc = pfn_check(voxels, num_points, coors, weight, pillars, slots, ndim);
    if (rc) return rc;
    S2D_CHECK_ARG(partial, "pfn_stats: null output");
    const PfnGeo g{vx, vy, x_offset, y_offset, slots, ndim};
    hipLaunchKernelGGL(pfn_stats_kernel, dim3(pfn_blocks(pillars)), dim3(256), 0, (hipStream_t)stream, voxels, num_points, coors, weight, pillars, g, partial);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_pfn_apply_max_f32(const float *voxels, const int32_t *num_points, const int32_t *coors, const float *weight, const float *scale,
                                     const float *shift, int64_t pillars, int slots, int ndim, float vx, float vy, float x_offset, float y_offset,
                                     float *out, uint8_t *argmax, s2d_stream_t stream) {
    int rc = pfn_check(voxels, num_points, coors, weight, pillars, slots, ndim);
    if (rc) return rc;
    S2D_CHECK_ARG(scale && shift && out, "pfn_apply_max: null argument");
    const PfnGeo g{vx, vy, x_offset, y_offset, slots, ndim};
    hipLaunchKernelGGL(pfn_apply_max_kernel, dim3(pfn_blocks(pillars)), dim3(256), 0, (hipStream_t)stream, voxels, num_points, coors, weight, scale, shift,
                       pillars, g, out, argmax);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_pfn_bwd_f32(const float *voxels, const int32_t *num_points, const int32_t *coors, const float *weight, const float *dout,
                               const uint8_t *argmax, int64_t pillars, int slots, int ndim, float vx, float vy, float x_offset, float y_offset,
                               float *partial, s2d_stream_t stream) {
    int rc = pfn_check(voxels, num_points, coors, weight, pillars, slots, ndim);
    if (rc) return rc;
    S2D_CHECK_ARG(dout && argmax && partial, "pfn_bwd: null argument");
    const PfnGeo g{vx, vy, x_offset, y_offset, slots, ndim};
    hipLaunchKernelGGL(pfn_bwd_kernel, dim3(pfn_blocks(pillars)), dim3(256), 0, (hipStream_t)stream, voxels, num_points, coors, weight, dout, argmax,
                       pillars, g, partial);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
