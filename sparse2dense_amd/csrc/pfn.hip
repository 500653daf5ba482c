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

// =====================================================================================================================================
// Two PFN layers (the Waymo pp reader, configs/waymo/pp/*: num_filters = [64, 64]):
//   layer 1: h1 = W1 f (10 -> 32), x1 = relu(bn1(h1)), m1 = max over the slots, z = [x1 | m1]        (pillar_encoder.py:41-56, not last: concat)
//   layer 2: h2 = W2 z (64 -> 64), x2 = relu(bn2(h2)), out = max over the slots                         (last layer)
// Empty slots carry f = 0, hence h1 = 0 and x1 = relu(shift1) =: e1 - they take part in m1, in h2 (= A e1 + B m1), in both batch norms'
// statistics and in the final max exactly as in the reference, but all empty slots of a pillar are identical: evaluated once, weighted
// by their number.  A wave owns a pillar; lane l = layer-2 channel l and layer-1 channel l & 31 (the upper half duplicates the lower);
// h2[l] = sum_j A[l][j] x1[j] + (B m1)[l] with x1[j] broadcast from lane j by v_readlane and the rows A = W2[:, :32], B = W2[:, 32:]
// in registers.  Passes (per-row products recomputed, nothing of size [P,20,*] stored):
//   F1 sums of h1, h1^2 | F2 sums of h2, h2^2 (all P*T rows) | F3 out, slot of the maximum, h2 at the maximum
//   B  (after the host finalised bn2's backward from dout and the kept h2): dh2 for every row, dW2 = sum dh2 (x) z in registers,
//      dz = W2^T dh2 (W2 columns read from LDS), g1 through max-1 / relu / concat, and the sums + matrices from which the host assembles
//      bn1's backward and dW1 (the one-layer scheme above).  One partial row per WAVE (1024 rows), folded by the host.
// =====================================================================================================================================
namespace s2d {

constexpr int PFN2_C1 = 32;
constexpr int PFN2_BWD_COLS = PFN_C * PFN_C + (2 + 2 * PFN_F) * PFN_C + PFN_F;   // dW2[64][64] | sum g1 | sum g1 h1 | M1[10][64] | M2[10][64] | M3[10]
constexpr int PFN2_PTS = PFN_T * 5;                                               // floats of one pillar

// The pillar of the NEXT loop iteration is fetched into two registers per lane while the current one is processed out of LDS: with one
// or two waves per SIMD nothing else hides the global-memory latency.  Lane e holds floats e and e + 64 of the [slots][5] pillar;
// the second register's spare lanes carry num_points and the four coordinates, so that everything arrives through vector loads (a
// scalar load would share LDS's wait counter and be waited for at the first LDS read).
struct PfnFetch {
    float v0;
    uint32_t v1;
};
__device__ __forceinline__ PfnFetch pfn2_fetch(const float *__restrict__ voxels, const int32_t *__restrict__ num, const int32_t *__restrict__ coors,
                                               int64_t p, int64_t pillars, const PfnGeo &g, int lane) {
    PfnFetch r{0.f, 0u};
    if (p >= pillars) return r;
    const int len = g.slots * g.ndim;
    const float *vox = voxels + p * len;
    if (lane < len) r.v0 = vox[lane];
    const int e = lane + 64;
    const uint32_t *src = e < len ? reinterpret_cast<const uint32_t *>(vox + e)
                          : e == PFN2_PTS ? reinterpret_cast<const uint32_t *>(num + p)
                          : (e > PFN2_PTS && e <= PFN2_PTS + 4) ? reinterpret_cast<const uint32_t *>(coors + p * 4 + (e - PFN2_PTS - 1))
                                                                : nullptr;
    if (src) r.v1 = *src;
    return r;
}
// per-wave LDS image of a pillar: pts[0..len) | [100] num_points | [101..104] b, z, y, x
struct PfnStage {
    float pts[PFN2_PTS + 8];
};
struct PfnHead {
    int n;
    float mean[3], cx, cy;
};
__device__ __forceinline__ PfnHead pfn2_stage(PfnStage &st, const PfnFetch &r, const PfnGeo &g, int lane) {
    __builtin_amdgcn_wave_barrier();
    st.pts[lane] = r.v0;
    if (lane + 64 < PFN2_PTS + 8) st.pts[lane + 64] = __builtin_bit_cast(float, r.v1);
    __builtin_amdgcn_wave_barrier();
    PfnHead h;
    h.n = __builtin_amdgcn_readfirstlane(min(max(__builtin_bit_cast(int, st.pts[PFN2_PTS]), 0), g.slots));
    const int cy_i = __builtin_bit_cast(int, st.pts[PFN2_PTS + 3]), cx_i = __builtin_bit_cast(int, st.pts[PFN2_PTS + 4]);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < PFN_T; ++t) {   // torch sums all slots (empty ones are zero); unrolled: the 60 LDS reads issue back to back
        if (t < g.slots) {
            s0 += st.pts[t * 5 + 0];
            s1 += st.pts[t * 5 + 1];
            s2 += st.pts[t * 5 + 2];
        }
    }
    const float fn = (float)h.n;
    h.mean[0] = __fdiv_rn(s0, fn); h.mean[1] = __fdiv_rn(s1, fn); h.mean[2] = __fdiv_rn(s2, fn);
    h.cx = __fadd_rn(__fmul_rn((float)cx_i, g.vx), g.x_offset);
    h.cy = __fadd_rn(__fmul_rn((float)cy_i, g.vy), g.y_offset);
    return h;
}
__device__ __forceinline__ float pfn2_h1(const PfnStage &st, int t, int ndim, const PfnHead &h, const float (&w)[PFN_F], float (&f)[PFN_F]) {
    pfn_slot_feats(st.pts + t * 5, h.mean, h.cx, h.cy, f);   // ndim == 5 (s2d_pfn2_supported)
    return pfn_dot(w, f);
}

// per-wave LDS of the layer-2 passes
struct Pfn2Wave {
    PfnStage st;
    float x1[PFN_T][PFN2_C1];      // relu(bn1(h1)) of every slot; row n = the empty slots' value when n < T
    float m1[PFN2_C1];
};
// (sum_j a[j] x[j]) + init with x[0..31] read from LDS at a wave- (or half-wave-) uniform address
__device__ __forceinline__ float pfn2_dot_lds(const float (&a)[PFN2_C1], const float *x, float init) {
    float h = init;
#pragma unroll
    for (int q = 0; q < PFN2_C1 / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4 *>(x + 4 * q);
        h = fmaf(a[4 * q + 0], v.x, h);
        h = fmaf(a[4 * q + 1], v.y, h);
        h = fmaf(a[4 * q + 2], v.z, h);
        h = fmaf(a[4 * q + 3], v.w, h);
    }
    return h;
}
// layer 1 of the staged pillar: the two half-waves take alternate slots (lane & 31 = channel).  Writes x1 rows [0, rows) (row n = e1 when
// n < T) and m1; returns the first-maximum slot (n = an empty slot) and the number of rows.
__device__ __forceinline__ int pfn2_layer1(Pfn2Wave &w, const PfnHead &h, const PfnGeo &g, const float (&w1)[PFN_F], float sc1, float sh1, int lane, int &arg1) {
    const int half = lane >> 5, c1 = lane & 31;
    float m = -1.f;
    int am = 0;
    for (int t = half; t < h.n; t += 2) {
        float f[PFN_F];
        const float x = fmaxf(fmaf(pfn2_h1(w.st, t, g.ndim, h, w1, f), sc1, sh1), 0.f);
        w.x1[t][c1] = x;
        if (x > m) { m = x; am = t; }
    }
    const float mo = __shfl_xor(m, 32);
    const int ao = __shfl_xor(am, 32);
    if (mo > m || (mo == m && ao < am)) { m = mo; am = ao; }
    const float e1 = fmaxf(sh1, 0.f);
    int rows = h.n;
    if (h.n < g.slots) {
        if (e1 > m) { m = e1; am = h.n; }
        w.x1[h.n][c1] = e1;
        rows = h.n + 1;
    }
    w.m1[c1] = m;
    arg1 = am;
    __builtin_amdgcn_wave_barrier();
    return rows;
}

__global__ __launch_bounds__(256) void pfn2_stats1_kernel(const float *__restrict__ voxels, const int32_t *__restrict__ num, const int32_t *__restrict__ coors,
                                                          const float *__restrict__ w1, int64_t pillars, PfnGeo g, float *__restrict__ partial) {
    __shared__ float lds[4 * 2 * 64];
    __shared__ PfnStage stage[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, half = lane >> 5;
    float w[PFN_F];
#pragma unroll
    for (int k = 0; k < PFN_F; ++k) w[k] = w1[(lane & 31) * PFN_F + k];
    float acc[2] = {0.f, 0.f};
    const int64_t step = (int64_t)gridDim.x * 4;
    int64_t p = (int64_t)blockIdx.x * 4 + wid;
    PfnFetch nx = pfn2_fetch(voxels, num, coors, p, pillars, g, lane);
    for (; p < pillars; p += step) {
        const PfnHead h = pfn2_stage(stage[wid], nx, g, lane);
        nx = pfn2_fetch(voxels, num, coors, p + step, pillars, g, lane);
        for (int t = half; t < h.n; t += 2) {
            float f[PFN_F];
            const float v = pfn2_h1(stage[wid], t, g.ndim, h, w, f);
            acc[0] += v;
            acc[1] = fmaf(v, v, acc[1]);
        }
    }
    acc[0] += __shfl_xor(acc[0], 32);
    acc[1] += __shfl_xor(acc[1], 32);
    pfn_block_fold<2>(acc, partial + (int64_t)blockIdx.x * 2 * PFN_C, lds);   // columns 32..63 duplicate 0..31 (ignored by the host)
}

// MODE 0: sums of h2, h2^2 over all rows -> partial[block][2][64];  MODE 1: out / arg / h2 at the maximum
template <int MODE>
__global__ __launch_bounds__(256) void pfn2_fwd_kernel(const float *__restrict__ voxels, const int32_t *__restrict__ num, const int32_t *__restrict__ coors,
                                                       const float *__restrict__ w1, const float *__restrict__ w2, const float *__restrict__ ss1,
                                                       const float *__restrict__ ss2, int64_t pillars, PfnGeo g, float *__restrict__ partial,
                                                       float *__restrict__ out, uint8_t *__restrict__ arg, float *__restrict__ h2max) {
    __shared__ float lds[4 * 2 * 64];
    __shared__ Pfn2Wave wv[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    Pfn2Wave &w = wv[wid];
    float wl1[PFN_F], wa[PFN2_C1], wb[PFN2_C1];
#pragma unroll
    for (int k = 0; k < PFN_F; ++k) wl1[k] = w1[(lane & 31) * PFN_F + k];
#pragma unroll
    for (int j = 0; j < PFN2_C1; ++j) {
        wa[j] = w2[lane * PFN_C + j];
        wb[j] = w2[lane * PFN_C + PFN2_C1 + j];
    }
    const float sc1 = ss1[lane & 31], sh1 = ss1[PFN2_C1 + (lane & 31)];
    const float sc2 = MODE ? ss2[lane] : 1.f, sh2 = MODE ? ss2[PFN_C + lane] : 0.f;
    float acc[2] = {0.f, 0.f};
    const int64_t step = (int64_t)gridDim.x * 4;
    int64_t p = (int64_t)blockIdx.x * 4 + wid;
    PfnFetch nx = pfn2_fetch(voxels, num, coors, p, pillars, g, lane);
    for (; p < pillars; p += step) {
        const PfnHead h = pfn2_stage(w.st, nx, g, lane);
        nx = pfn2_fetch(voxels, num, coors, p + step, pillars, g, lane);
        int arg1;
        const int rows = pfn2_layer1(w, h, g, wl1, sc1, sh1, lane, arg1);
        const float bm = pfn2_dot_lds(wb, w.m1, 0.f);   // (B m1)[l]
        float best = -1.f, hbest = 0.f;
        int bi = 0;
        for (int t = 0; t < rows; ++t) {   // row n (when n < T) stands for all the empty slots
            const float h2 = pfn2_dot_lds(wa, w.x1[t], bm);
            if (MODE == 0) {
                const float mult = t == h.n ? (float)(g.slots - h.n) : 1.f;
                acc[0] = fmaf(mult, h2, acc[0]);
                acc[1] = fmaf(mult * h2, h2, acc[1]);
            } else {
                const float y = fmaxf(fmaf(h2, sc2, sh2), 0.f);
                if (y > best) { best = y; bi = t; hbest = h2; }
            }
        }
        if (MODE == 1) {
            out[p * PFN_C + lane] = best;
            arg[p * PFN_C + lane] = (uint8_t)bi;
            h2max[p * PFN_C + lane] = hbest;
        }
    }
    if (MODE == 0) pfn_block_fold<2>(acc, partial + (int64_t)blockIdx.x * 2 * PFN_C, lds);
}

struct Pfn2BwdWave {
    Pfn2Wave f;
    float dh2[PFN_T][PFN_C];       // dh2 of every row (row n: summed over the empty rows)
    float sdh2[PFN_C];
};

__global__ __launch_bounds__(256) void pfn2_bwd_kernel(const float *__restrict__ voxels, const int32_t *__restrict__ num, const int32_t *__restrict__ coors,
                                                       const float *__restrict__ w1, const float *__restrict__ w2, const float *__restrict__ ss1,
                                                       const float *__restrict__ abd2, const float *__restrict__ gout, const uint8_t *__restrict__ arg2,
                                                       int64_t pillars, PfnGeo g, float *__restrict__ partial) {
    constexpr int LD = PFN_C + 1;                // padded rows: lane j reading column j AND lane l reading row l are both conflict-free
    __shared__ float w2s[PFN_C * LD];            // W2[c][j] at c * LD + j
    __shared__ Pfn2BwdWave wv[4];
    for (int i = threadIdx.x; i < PFN_C * PFN_C; i += 256) w2s[(i >> 6) * LD + (i & 63)] = w2[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, half = lane >> 5, c1 = lane & 31;
    Pfn2BwdWave &w = wv[wid];
    float wl1[PFN_F], wa[PFN2_C1];
#pragma unroll
    for (int k = 0; k < PFN_F; ++k) wl1[k] = w1[c1 * PFN_F + k];
#pragma unroll
    for (int j = 0; j < PFN2_C1; ++j) wa[j] = w2[lane * PFN_C + j];
    const float sc1 = ss1[c1], sh1 = ss1[PFN2_C1 + c1];
    const float a2 = abd2[lane], b2 = abd2[PFN_C + lane], d2 = abd2[2 * PFN_C + lane];
    const float e1 = fmaxf(sh1, 0.f);
    float dwa[PFN2_C1], dwb[PFN2_C1];      // dW2[l][0..31], dW2[l][32..63]
    float acc1[2 + 2 * PFN_F];             // layer 1, per half-wave: sum g1, sum g1 h1, M1[k], M2[k]
    float m3[PFN_F];
#pragma unroll
    for (int j = 0; j < PFN2_C1; ++j) dwa[j] = dwb[j] = 0.f;
#pragma unroll
    for (int k = 0; k < 2 + 2 * PFN_F; ++k) acc1[k] = 0.f;
#pragma unroll
    for (int k = 0; k < PFN_F; ++k) m3[k] = 0.f;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wid, step = (int64_t)gridDim.x * 4;
    int64_t p = wave;
    PfnFetch nx = pfn2_fetch(voxels, num, coors, p, pillars, g, lane);
    float go_n = p < pillars ? gout[p * PFN_C + lane] : 0.f;   // dout with relu' already applied
    int bi_n = p < pillars ? arg2[p * PFN_C + lane] : 0;
    for (; p < pillars; p += step) {
        const PfnHead h = pfn2_stage(w.f.st, nx, g, lane);
        const float go = go_n;
        const int bi = bi_n;
        nx = pfn2_fetch(voxels, num, coors, p + step, pillars, g, lane);
        if (p + step < pillars) {
            go_n = gout[(p + step) * PFN_C + lane];
            bi_n = arg2[(p + step) * PFN_C + lane];
        }
        int arg1;
        const int rows = pfn2_layer1(w.f, h, g, wl1, sc1, sh1, lane, arg1);
        // the once-per-pillar weight reads (row l of W2[:, 32:] for bm, its columns for dm1) stay in LDS: hoisted out of the pillar loop
        // they would take 96 more registers next to the 64 of the W2[:, :32] column the row loop below keeps, and the kernel could not
        // hold two waves per SIMD.  The opaque offset keeps the compiler from proving the loads loop-invariant.
        int once = 0;
        asm volatile("" : "+v"(once));
        const float *w2bt_l = w2s + lane * LD + PFN2_C1 + once, *w2s_m = w2s + PFN2_C1 + c1 + once;
        float bm = 0.f;
#pragma unroll
        for (int q = 0; q < PFN2_C1 / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(w.f.m1 + 4 * q);
            bm = fmaf(w2bt_l[4 * q + 0], v.x, bm);
            bm = fmaf(w2bt_l[4 * q + 1], v.y, bm);
            bm = fmaf(w2bt_l[4 * q + 2], v.z, bm);
            bm = fmaf(w2bt_l[4 * q + 3], v.w, bm);
        }
        // layer 2 of every row (lane = channel): dh2, dW2[:, :32], the rows' dh2 to LDS for the transposed product below
        float sdh2 = 0.f;
        for (int t = 0; t < rows; ++t) {
            float4 xv[PFN2_C1 / 4];   // the row's x1 (wave-uniform LDS reads), kept for the dW2 update: one wave per SIMD, registers are free
            float h2 = bm;
#pragma unroll
            for (int q = 0; q < PFN2_C1 / 4; ++q) {
                xv[q] = *reinterpret_cast<const float4 *>(w.f.x1[t] + 4 * q);
                h2 = fmaf(wa[4 * q + 0], xv[q].x, h2);
                h2 = fmaf(wa[4 * q + 1], xv[q].y, h2);
                h2 = fmaf(wa[4 * q + 2], xv[q].z, h2);
                h2 = fmaf(wa[4 * q + 3], xv[q].w, h2);
            }
            const float rest = fmaf(b2, h2, d2);
            // row n: ONE of the empty rows holds an arg-2 maximum, all of them carry the b / d terms
            const float dh2 = t == h.n ? fmaf(a2, bi == t ? go : 0.f, (float)(g.slots - h.n) * rest) : fmaf(a2, bi == t ? go : 0.f, rest);
            sdh2 += dh2;
#pragma unroll
            for (int q = 0; q < PFN2_C1 / 4; ++q) {
                dwa[4 * q + 0] = fmaf(dh2, xv[q].x, dwa[4 * q + 0]);
                dwa[4 * q + 1] = fmaf(dh2, xv[q].y, dwa[4 * q + 1]);
                dwa[4 * q + 2] = fmaf(dh2, xv[q].z, dwa[4 * q + 2]);
                dwa[4 * q + 3] = fmaf(dh2, xv[q].w, dwa[4 * q + 3]);
            }
            w.dh2[t][lane] = dh2;
        }
        w.sdh2[lane] = sdh2;
        __builtin_amdgcn_wave_barrier();
        // dW2[:, 32:] += sdh2 (x) m1 ;  dm1[j] = sum_c W2[c][32 + j] sdh2[c]
        float dm1 = 0.f;
#pragma unroll
        for (int q = 0; q < PFN2_C1 / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(w.f.m1 + 4 * q);
            dwb[4 * q + 0] = fmaf(sdh2, v.x, dwb[4 * q + 0]);
            dwb[4 * q + 1] = fmaf(sdh2, v.y, dwb[4 * q + 1]);
            dwb[4 * q + 2] = fmaf(sdh2, v.z, dwb[4 * q + 2]);
            dwb[4 * q + 3] = fmaf(sdh2, v.w, dwb[4 * q + 3]);
        }
#pragma unroll
        for (int q = 0; q < PFN_C / 4; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(w.sdh2 + 4 * q);
            dm1 = fmaf(w2s_m[(4 * q + 0) * LD], v.x, dm1);
            dm1 = fmaf(w2s_m[(4 * q + 1) * LD], v.y, dm1);
            dm1 = fmaf(w2s_m[(4 * q + 2) * LD], v.z, dm1);
            dm1 = fmaf(w2s_m[(4 * q + 3) * LD], v.w, dm1);
        }
        // layer 1: the half-waves take alternate rows (lane & 31 = channel): dz = W2[:, :32]^T dh2, then max-1 / relu / bn1 / W1 sums
        for (int t = half; t < rows; t += 2) {
            float dz = 0.f;
#pragma unroll
            for (int q = 0; q < PFN_C / 4; ++q) {
                const float4 v = *reinterpret_cast<const float4 *>(w.dh2[t] + 4 * q);
                dz = fmaf(w2s[(4 * q + 0) * LD + c1], v.x, dz);
                dz = fmaf(w2s[(4 * q + 1) * LD + c1], v.y, dz);
                dz = fmaf(w2s[(4 * q + 2) * LD + c1], v.z, dz);
                dz = fmaf(w2s[(4 * q + 3) * LD + c1], v.w, dz);
            }
            const float gin = dz + (t == arg1 ? dm1 : 0.f);
            if (t == h.n) {   // the empty rows: h1 = 0, f = 0 - only the plain sum
                acc1[0] += e1 > 0.f ? gin : 0.f;
            } else {
                float f[PFN_F];
                const float h1 = pfn2_h1(w.f.st, t, g.ndim, h, wl1, f);
                const float g1 = fmaf(h1, sc1, sh1) > 0.f ? gin : 0.f;
                acc1[0] += g1;
                acc1[1] = fmaf(g1, h1, acc1[1]);
#pragma unroll
                for (int k = 0; k < PFN_F; ++k) {
                    acc1[2 + k] = fmaf(g1, f[k], acc1[2 + k]);
                    acc1[2 + PFN_F + k] = fmaf(h1, f[k], acc1[2 + PFN_F + k]);
                    m3[k] += f[k];
                }
            }
        }
    }
    // one row per wave: dW2[l][0..63] | layer-1 sums (both half-waves folded, columns 0..31) | M3
    float *row = partial + wave * PFN2_BWD_COLS;
#pragma unroll
    for (int j = 0; j < PFN2_C1; ++j) {
        row[lane * PFN_C + j] = dwa[j];
        row[lane * PFN_C + PFN2_C1 + j] = dwb[j];
    }
    float *r1 = row + PFN_C * PFN_C;
#pragma unroll
    for (int k = 0; k < 2 + 2 * PFN_F; ++k) {
        const float v = acc1[k] + __shfl_xor(acc1[k], 32);
        r1[k * PFN_C + lane] = half == 0 ? v : 0.f;
    }
    float mv = 0.f;
#pragma unroll
    for (int k = 0; k < PFN_F; ++k) {
        const float v = m3[k] + __shfl_xor(m3[k], 32);
        mv = lane == k ? v : mv;
    }
    if (lane < PFN_F) r1[(2 + 2 * PFN_F) * PFN_C + lane] = mv;
}

constexpr int PFN2_BWD_BLOCKS = 256;   // one 4-wave workgroup per CU: the kernel keeps > 256 registers per lane (see the r04 notes in its body)

}  // namespace s2d

extern "C" int s2d_pfn2_supported(int ndim, int slots, int feats, int c1, int c2) {
    return ndim == 5 && slots >= 1 && slots <= PFN_T && feats == PFN_F && c1 == PFN2_C1 && c2 == PFN_C;
}
extern "C" int s2d_pfn2_bwd_rows(void) { return PFN2_BWD_BLOCKS * 4; }
extern "C" int s2d_pfn2_bwd_cols(void) { return PFN2_BWD_COLS; }

#define S2D_PFN2_COMMON(what)                                                                                                           \
    int rc = pfn_check(voxels, num_points, coors, w1, pillars, slots, ndim);                                                            \
    if (rc) return rc;                                                                                                                  \
    S2D_CHECK_ARG(w2, what ": null second-layer weight");                                                                              \
    const PfnGeo g{vx, vy, x_offset, y_offset, slots, ndim};                                                                            \
    hipStream_t st = (hipStream_t)stream

extern "C" int s2d_pfn2_stats1_f32(const float *voxels, const int32_t *num_points, const int32_t *coors, const float *w1, int64_t pillars, int slots,
                                   int ndim, float vx, float vy, float x_offset, float y_offset, float *partial, s2d_stream_t stream) {
    int rc = pfn_check(voxels, num_points, coors, w1, pillars, slots, ndim);
    if (rc) return rc;
    S2D_CHECK_ARG(partial, "pfn2_stats1: null output");
    const PfnGeo g{vx, vy, x_offset, y_offset, slots, ndim};
    hipLaunchKernelGGL(pfn2_stats1_kernel, dim3(pfn_blocks(pillars)), dim3(256), 0, (hipStream_t)stream, voxels, num_points, coors, w1, pillars, g, partial);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

/* scale_shift1 = scale[32] | shift[32] of the first batch norm */
extern "C" int s2d_pfn2_stats2_f32(const float *voxels, const int32_t *num_points, const int32_t *coors, const float *w1, const float *w2,
                                   const float *scale_shift1, int64_t pillars, int slots, int ndim, float vx, float vy, float x_offset,
                                   float y_offset, float *partial, s2d_stream_t stream) {
    S2D_PFN2_COMMON("pfn2_stats2");
    S2D_CHECK_ARG(scale_shift1 && partial, "pfn2_stats2: null argument");
    hipLaunchKernelGGL(pfn2_fwd_kernel<0>, dim3(pfn_blocks(pillars)), dim3(256), 0, st, voxels, num_points, coors, w1, w2, scale_shift1, nullptr, pillars, g,
                       partial, nullptr, nullptr, nullptr);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" int s2d_pfn2_apply_max_f32(const float *voxels, const int32_t *num_points, const int32_t *coors, const float *w1, const float *w2,
                                      const float *scale_shift1, const float *scale_shift2, int64_t pillars, int slots, int ndim, float vx, float vy,
                                      float x_offset, float y_offset, float *out, uint8_t *argmax, float *h2_at_max, s2d_stream_t stream) {
    S2D_PFN2_COMMON("pfn2_apply_max");
    S2D_CHECK_ARG(scale_shift1 && scale_shift2 && out && argmax && h2_at_max, "pfn2_apply_max: null argument");
    hipLaunchKernelGGL(pfn2_fwd_kernel<1>, dim3(pfn_blocks(pillars)), dim3(256), 0, st, voxels, num_points, coors, w1, w2, scale_shift1, scale_shift2, pillars,
                       g, nullptr, out, argmax, h2_at_max);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

/* abd2 = a[64] | b[64] | d[64] of the second batch norm's backward (dh2 = a g + b h2 + d); gout = dout with relu' applied;
 * partial: [s2d_pfn2_bwd_rows()][s2d_pfn2_bwd_cols()] = dW2[64][64] | sum g1[64] | sum g1 h1[64] | M1[10][64] | M2[10][64] | M3[10] per wave */
extern "C" int s2d_pfn2_bwd_f32(const float *voxels, const int32_t *num_points, const int32_t *coors, const float *w1, const float *w2,
                                const float *scale_shift1, const float *abd2, const float *gout, const uint8_t *argmax, int64_t pillars, int slots,
                                int ndim, float vx, float vy, float x_offset, float y_offset, float *partial, s2d_stream_t stream) {
    S2D_PFN2_COMMON("pfn2_bwd");
    S2D_CHECK_ARG(scale_shift1 && abd2 && gout && argmax && partial, "pfn2_bwd: null argument");
    hipLaunchKernelGGL(pfn2_bwd_kernel, dim3(PFN2_BWD_BLOCKS), dim3(256), 0, st, voxels, num_points, coors, w1, w2, scale_shift1, abd2, gout, argmax, pillars, g,
                       partial);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
