/*
 * s2d.h — C-ABI of libs2d_hip.so: the MI355X (gfx950) hot path of Sparse2Dense / CenterPoint.
 *
 * One `extern "C"` shared object, plain pointers and sizes only (no torch / C++ types).  It
 * replaces, for the voxel-backbone path, what the reference binds through numba and through the
 * third-party `spconv` extension (reference = /root/reference, stevewongv/Sparse2Dense):
 *
 *   s2d_voxelize_*        <- det3d/ops/point_cloud/point_cloud_ops.py:7-55,112-184
 *                            (points_to_voxel / _points_to_voxel_reverse_kernel) fused with
 *                            det3d/models/readers/voxel_encoder.py:17-24 (per-voxel mean)
 *   s2d_rulebook_subm_*   <- spconv.ops.get_indice_pairs(subm=True)  as called by SubMConv3d at
 *                            det3d/models/backbones/scn.py:18-26,105,202-238
 *   s2d_rulebook_conv_*   <- spconv.ops.get_indice_pairs(subm=False) as called by SparseConv3d at
 *                            det3d/models/backbones/scn.py:116-118,126-128,136-138,147-149
 *   s2d_spconv_fwd / _wgrad <- spconv.ops.indice_conv / indice_conv_backward (same call sites)
 *   s2d_bn1d_*            <- nn.BatchNorm1d + ReLU + residual on `.features`, scn.py:69-85,104-152
 *   s2d_densify_*         <- spconv.SparseConvTensor.dense(), scn.py:173-176, voxelnet.py:203-215
 *
 * Conventions
 *   - every entry returns int: 0 = OK, <0 = invalid argument / unsupported shape / capacity,
 *     >0 = hipError_t of a failed HIP call; text via s2d_last_error() (thread-local).
 *   - all buffers are caller-owned DEVICE pointers (PyTorch caching allocator in our host code);
 *     the library never allocates, frees or retains device memory.
 *   - every entry is asynchronous on the caller's stream (`s2d_stream_t` = hipStream_t passed as
 *     void*; NULL = the null stream) and never synchronises the device.
 *   - variable-size results use a device scalar (`out_m`, `out_n`) that the HOST reads back only
 *     where the reference API itself exposes a size.
 *   - indices are int32, features fp32 (bf16 entry points carry the _bf16 suffix).
 */
#ifndef S2D_H_
#define S2D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *s2d_stream_t;

#define S2D_OK 0
#define S2D_ERR_INVALID_ARG (-1)
#define S2D_ERR_UNSUPPORTED (-2)
#define S2D_ERR_CAPACITY (-3)
#define S2D_ERR_WORKSPACE (-4)
#define S2D_ERR_COMM (-5)

/* ---- library ------------------------------------------------------------------------------ */
int s2d_version(void);
/* copies the calling thread's last error text (NUL terminated) into buf; returns its length */
int s2d_last_error(char *buf, size_t buf_len);
/* copies "compiler; HIP runtime headers; target; build date" of this library into buf; returns the text's length (no reference
 * counterpart: the reference's kernels come prebuilt with spconv / torch, docs/INSTALL.md:12,65-72) */
int s2d_build_info(char *buf, size_t buf_len);

/* ---- voxelization (hard voxelizer + fused reader mean) ------------------------------------ */
/*
 * Batched form: `frames` point clouds concatenated frame after frame (point_offsets[frames + 1], HOST array, offsets[0] = 0),
 * voxelized in ONE launch chain into the COLLATED example (preprocess.py:316-345 per frame + collate.py:105-144): voxels / num_points /
 * mean rows frame by frame in first-seen order, coors4[row] = (b, z, y, x) with the batch index prepended, out_m[frames] = voxels
 * per frame (device), out_base[frames + 1] = their exclusive prefix (out_base[frames] = total rows).  Outputs must hold
 * sum_b min(n_b, max_voxels) rows.  Same per-frame results as s2d_voxelize_run (bit exact).  frames <= 64.
 */
size_t s2d_voxelize_batch_workspace_bytes(int frames, const int64_t *point_offsets, int max_points, int max_voxels);
int s2d_voxelize_batch_run(const float *points, int frames, const int64_t *point_offsets, int ndim,
                           const float coors_range[6], const float voxel_size[3], int max_points, int max_voxels,
                           float *voxels, int32_t *coors4, int32_t *num_points, float *mean, int32_t *out_m,
                           int32_t *out_base, void *ws, size_t ws_bytes, s2d_stream_t stream);
/* workspace bytes for one call (hash table, per-point slots, scan scratch, k-smallest lists) */
size_t s2d_voxelize_workspace_bytes(int64_t n_points, int max_points, int max_voxels);
/*
 * points[n_points][ndim] fp32 (x,y,z,...) ; coors_range = xmin,ymin,zmin,xmax,ymax,zmax ;
 * voxel_size = (vx,vy,vz).  Outputs (rows >= *out_m are left untouched):
 *   voxels[max_voxels][max_points][ndim] fp32 (zero padded), coors[max_voxels][3] int32 (z,y,x),
 *   num_points[max_voxels] int32, mean[max_voxels][ndim] fp32 (may be NULL), out_m device int32.
 * Voxel order = order of first appearance in `points`; slot order = point order; voxels whose
 * first point comes after the max_voxels-th distinct voxel are dropped (reference semantics).
 */
int s2d_voxelize_run(const float *points, int64_t n_points, int ndim, const float coors_range[6],
                     const float voxel_size[3], int max_points, int max_voxels, float *voxels,
                     int32_t *coors, int32_t *num_points, float *mean, int32_t *out_m, void *ws,
                     size_t ws_bytes, s2d_stream_t stream);

/* ---- rulebooks ---------------------------------------------------------------------------- */
/*
 * A rulebook is a pair of dense gather maps (K = kD*kH*kW offsets, row-major over kz,ky,kx):
 *   nbr_out[k][o] = input row j feeding output row o through offset k, or -1
 *   nbr_in [k][j] = output row o fed by input row j through offset k, or -1   (transposed map,
 *                   used by the data-gradient; for SubM it is nbr_out[K-1-k] and not stored)
 * pair_count[k] = number of (j -> o) pairs of offset k (what spconv calls indice_pair_num).
 * `shape` is the (D,H,W) extent the coordinates index; coors rows are (b,z,y,x).
 */
size_t s2d_rulebook_workspace_bytes(int batch, const int32_t shape[3], int64_t n_rows);

/* SubM: output rows = input rows, same order; centred (pad = k/2, stride 1). */
int s2d_rulebook_subm_build(const int32_t *coors, int64_t n, int batch, const int32_t shape[3],
                            const int32_t ksize[3], const int32_t dilation[3], int32_t *nbr_out,
                            int32_t *pair_count, void *ws, size_t ws_bytes, s2d_stream_t stream);

/*
 * Regular (strided) conv, two phases around the one host read of *out_n:
 *   _count: marks every reachable output site, numbers them in sorted linear (b,z,y,x) order and
 *           writes the count to the device scalar out_n (workspace keeps the occupancy index);
 *   _fill : with n_out = *out_n known to the host, writes out_coors[n_out][4], nbr_out[K][n_out],
 *           nbr_in[K][n] and pair_count[K].  Must reuse the same workspace, unchanged.
 */
int s2d_rulebook_conv_count(const int32_t *coors, int64_t n, int batch, const int32_t shape[3],
                            const int32_t ksize[3], const int32_t stride[3],
                            const int32_t padding[3], const int32_t dilation[3], int32_t *out_n,
                            void *ws, size_t ws_bytes, s2d_stream_t stream);
int s2d_rulebook_conv_fill(const int32_t *coors, int64_t n, int batch, const int32_t shape[3],
                           const int32_t ksize[3], const int32_t stride[3],
                           const int32_t padding[3], const int32_t dilation[3], int64_t n_out,
                           int32_t *out_coors, int32_t *nbr_out, int32_t *nbr_in,
                           int32_t *pair_count, void *ws, size_t ws_bytes, s2d_stream_t stream);

/*
 * r04 - every rulebook of one backbone pass in one chain (spconv get_indice_pairs x 8 as called by
 * det3d/models/backbones/scn.py:104-152: conv_input/conv1 .. conv4 SubM keys res0..res3 + the four strided convs).
 * Stage 0 = the input rows (any order); stage l+1 = outputs of strided conv l (ksize/stride/padding are [n_strided][3],
 * dilation 1, the first conv must be k3 s2 p1: its outputs index the 2x2x2 blocks of stage 0).
 *   _plan : marks + numbers the sites of stages 1..n (canonical sorted (b,z,y,x) order) and writes their row counts to the
 *           device vector counts[n_strided + 1] (last element: non-zero = internal error).  ONE host read per pass.
 *   _fill : with the counts known to the host (n_rows[n_strided]) writes out_coors[l] (stage l+1 rows), subm_nbr[l]
 *           (i32[27][N_l], l = 0..n_strided, NULL = not wanted), conv_nbr_out[l] (i32[K_l][N_{l+1}]), conv_nbr_in[l]
 *           (i32[K_l][N_l]) and pair_counts (i32[(2 n_strided + 1)][27]: SubM of stage l in row l, conv l in row
 *           n_strided + 1 + l).  `child` = scratch i32[n_rows[0]][8].  Same workspace as _plan, unchanged in between.
 * Maps are dense gather maps as above; every pointer array lives on the HOST and holds device pointers.
 */
int s2d_rulebook_chain_supported(int batch, const int32_t shape0[3], int n_strided, const int32_t *ksize,
                                 const int32_t *stride, const int32_t *padding);
size_t s2d_rulebook_chain_workspace_bytes(int batch, const int32_t shape0[3], int n_strided, const int32_t *ksize,
                                          const int32_t *stride, const int32_t *padding);
int s2d_rulebook_chain_plan(const int32_t *coors0, int64_t n0, int batch, const int32_t shape0[3], int n_strided,
                            const int32_t *ksize, const int32_t *stride, const int32_t *padding, int32_t *counts,
                            void *ws, size_t ws_bytes, s2d_stream_t stream);
int s2d_rulebook_chain_fill(const int32_t *coors0, int64_t n0, int batch, const int32_t shape0[3], int n_strided,
                            const int32_t *ksize, const int32_t *stride, const int32_t *padding, const int64_t *n_rows,
                            int32_t *const *out_coors, int32_t *const *subm_nbr, int32_t *const *conv_nbr_out,
                            int32_t *const *conv_nbr_in, int32_t *pair_counts, int32_t *child, void *ws, size_t ws_bytes,
                            s2d_stream_t stream);

/* ---- sparse convolution (implicit GEMM on MFMA) ------------------------------------------- */
/*
 * out[o][:] = sum_k in[nbr[k][o]][:] * weight[k][:][:] (+ bias), fp32 in / fp32 accumulate on
 * v_mfma_f32_16x16x4_f32.  weight is [K][cin][cout] (spconv v1.x layout [kD,kH,kW,Cin,Cout]).
 * The same entry computes the data gradient when given nbr_in and the transposed weights.
 * `nbr` has row stride n_out.
 */
int s2d_spconv_fwd_f32(const float *in_feat, int64_t n_in, const float *weight, const float *bias,
                       const int32_t *nbr, int64_t n_out, int kvol, int cin, int cout,
                       float *out_feat, s2d_stream_t stream);
/*
 * bf16-input / fp32-accumulate variant (v_mfma_f32_16x16x32_bf16).  Features and outputs stay
 * fp32 in HBM; gathered rows are rounded to bf16 (RNE) in registers.  The weights are packed once
 * per call into the LDS image the kernel stages linearly (kvol*cin*cout bf16 = 2 bytes each).
 * Supported: cin in {32,64,128}, cout in {16,32,64,128} (s2d_spconv_bf16_supported).
 */
int s2d_spconv_bf16_supported(int cin, int cout);
/* (cin, cout) describe the packed operand.  transpose=1: `weight` is stored [K][cout][cin] (the
 * data gradient multiplies by W^T); flip=1: offsets mirrored, k -> K-1-k (SubM transposed map). */
int s2d_spconv_pack_weights_bf16(const float *weight, int kvol, int cin, int cout, int transpose,
                                 int flip, void *packed, s2d_stream_t stream);
int s2d_spconv_fwd_bf16(const float *in_feat, int64_t n_in, const void *packed_weight,
                        const float *bias, const int32_t *nbr, int64_t n_out, int kvol, int cin,
                        int cout, float *out_feat, s2d_stream_t stream);
/* dweight[k] = sum_o in[nbr[k][o]]^T * dout[o]  ([K][cin][cout]); deterministic two-pass. */
size_t s2d_spconv_wgrad_workspace_bytes(int64_t n_out, int kvol, int cin, int cout);
int s2d_spconv_wgrad_f32(const float *in_feat, int64_t n_in, const float *dout, const int32_t *nbr,
                         int64_t n_out, int kvol, int cin, int cout, float *dweight, void *ws,
                         size_t ws_bytes, s2d_stream_t stream);

/* same contract, MFMA inputs rounded to bf16 (fp32 accumulate); channel counts that the matrix path
 * does not cover fall back to the fp32 kernels inside. */
int s2d_spconv_wgrad_bf16(const float *in_feat, int64_t n_in, const float *dout, const int32_t *nbr,
                          int64_t n_out, int kvol, int cin, int cout, float *dweight, void *ws,
                          size_t ws_bytes, s2d_stream_t stream);

/* ---- BatchNorm1d on features (+ReLU, +residual) ------------------------------------------- */
/* stats[0..C) = sum_x, stats[C..2C) = sum_x^2 over the n rows (deterministic tree reduction) */
size_t s2d_bn1d_workspace_bytes(int64_t n, int c);
int s2d_bn1d_stats_f32(const float *x, int64_t n, int c, float *stats, void *ws, size_t ws_bytes,
                       s2d_stream_t stream);
/*
 * Per-channel finalisation, one launch each (replaces ~15 elementwise kernels per BN layer):
 *  fwd: stats[2C] and the (possibly all-reduced) row count -> mean, invstd, scale = gamma*invstd,
 *       shift = beta - mean*scale; running_mean/var (may both be NULL) updated with `momentum`
 *       and the unbiased variance, exactly like nn.BatchNorm1d.
 *  bwd: sums_local (this rank) and sums_global (all ranks; same pointer on one GPU) of
 *       [sum g, sum g*x] -> dgamma, dbeta (local, DDP averages them) and the three vectors of
 *       s2d_bn1d_bwd_apply_f32 (global).
 */
int s2d_bn1d_finalize_fwd_f32(const float *stats, const float *count, const float *gamma,
                              const float *beta, float eps, float momentum, int c, float *mean,
                              float *invstd, float *scale, float *shift, float *running_mean,
                              float *running_var, int64_t *batches_tracked, s2d_stream_t stream);
int s2d_bn1d_finalize_bwd_f32(const float *sums_local, const float *sums_global, const float *count,
                              const float *gamma, const float *mean, const float *invstd, int c,
                              float *dgamma, float *dbeta, float *a, float *b, float *d,
                              s2d_stream_t stream);
/* single-GPU fusions of {stats, finalize_fwd} and {bwd_reduce, finalize_bwd}: two launches each,
 * no intermediate [2C] vector on the host side (the multi-rank path keeps the split entries so the
 * sums can be all-reduced in between). */
int s2d_bn1d_stats_finalize_f32(const float *x, int64_t n, int c, const float *gamma,
                                const float *beta, float eps, float momentum, float *mean,
                                float *invstd, float *scale, float *shift, float *running_mean,
                                float *running_var, int64_t *batches_tracked, void *ws,
                                size_t ws_bytes, s2d_stream_t stream);
int s2d_bn1d_bwd_reduce_finalize_f32(const float *dy, const float *y, const float *x, int relu,
                                     int64_t n, int c, const float *gamma, const float *mean,
                                     const float *invstd, float *g_out, float *dgamma, float *dbeta,
                                     float *a, float *b, float *d, void *ws, size_t ws_bytes,
                                     s2d_stream_t stream);
/* y = relu?( (x - mean) * invstd * gamma + beta (+ residual) ); scale/shift precomputed [C] */
int s2d_bn1d_apply_f32(const float *x, const float *scale, const float *shift,
                       const float *residual, int relu, int64_t n, int c, float *y,
                       s2d_stream_t stream);
/*
 * Backward of y = relu?(x*scale + shift (+res)):  g = dy * (y > 0 if relu);
 * sums[0..C) = sum g, sums[C..2C) = sum g * x  (for dgamma/dbeta and the batch-stat terms);
 * g is written to dres (may alias nothing; may be NULL when there is no residual and dx is
 * produced by the second call).
 */
int s2d_bn1d_bwd_reduce_f32(const float *dy, const float *y, const float *x, int relu, int64_t n,
                            int c, float *g_out, float *sums, void *ws, size_t ws_bytes,
                            s2d_stream_t stream);
/* dx = a[c]*g + b[c]*x + d[c]  (training-mode BN backward folded into three per-channel vectors) */
int s2d_bn1d_bwd_apply_f32(const float *g, const float *x, const float *a, const float *b,
                           const float *d, int64_t n, int c, float *dx, s2d_stream_t stream);

/* ---- densify (SparseConvTensor.dense()) --------------------------------------------------- */
/* out[B][C][D][H][W] (zero filled here) <- features[n][C] at coors[n][4]; bwd is the gather. */
int s2d_densify_fwd_f32(const float *feat, const int32_t *coors, int64_t n, int batch,
                        const int32_t shape[3], int c, float *out, s2d_stream_t stream);
int s2d_densify_bwd_f32(const float *dout, const int32_t *coors, int64_t n, int batch,
                        const int32_t shape[3], int c, float *dfeat, s2d_stream_t stream);
/* dense() + view(N, C*D, H, W) written directly in the layout the bf16 BEV neck consumes:
 * out / dout are [batch][H][W][c*D] bf16 (torch channels_last of [batch][c*D][H][W]), BEV channel
 * = ch*D + z (scn.py:173-176).  feat / dfeat are [n][c], fp32 or (feat_bf16=1) bf16. */
int s2d_densify_bev_fwd_bf16(const void *feat, int feat_bf16, const int32_t *coors, int64_t n,
                             int batch, const int32_t shape[3], int c, void *out,
                             s2d_stream_t stream);
int s2d_densify_bev_bwd_bf16(const void *dout, const int32_t *coors, int64_t n, int batch,
                             const int32_t shape[3], int c, void *dfeat, int feat_bf16,
                             s2d_stream_t stream);

/* ---- PCR head of the S2D neck (dense, NCDHW fp32, HBM-bound) -------------------------------- */
/*
 * 1x1x1 convolution = per-position channel mixing (nn.Conv3d(k=1) / nn.Conv2d(k=1),
 * det3d/models/necks/rpn.py:263-296): out[n][co][p] = bias[co] + sum_ci weight[co][ci]*in[n][ci][p].
 * The data gradient is the same entry with the transposed weight and bias = NULL.
 */
int s2d_pointwise_conv_f32(const float *in, const float *weight, const float *bias, int batch,
                           int cin, int cout, int64_t positions, float *out, s2d_stream_t stream);
/*
 * nn.ConvTranspose3d(kernel 4, stride 2, padding 1) (rpn.py:267,284): in [n][cin][d][h][w] ->
 * out [n][cout][2d][2h][2w]; weight [cin][cout][4][4][4].  _dgrad is its input gradient.
 */
int s2d_convt3d_k4s2p1_fwd_f32(const float *in, const float *weight, const float *bias, int batch,
                               int cin, int cout, int d, int h, int w, float *out,
                               s2d_stream_t stream);
int s2d_convt3d_k4s2p1_dgrad_f32(const float *dout, const float *weight, int batch, int cin,
                                 int cout, int d, int h, int w, float *din, s2d_stream_t stream);
/* weight gradient [cin][cout][4][4][4] (cin, cout <= 32), fp32 matrix cores, deterministic */
size_t s2d_convt3d_k4s2p1_wgrad_workspace_bytes(int batch, int cin, int cout, int d, int h, int w);
int s2d_convt3d_k4s2p1_wgrad_f32(const float *in, const float *dout, int batch, int cin, int cout,
                                 int d, int h, int w, float *dweight, void *ws, size_t ws_bytes,
                                 s2d_stream_t stream);

/*
 * Channel-major batch norm (nn.BatchNorm3d / BatchNorm2d on NC[D]HW fp32 with few channels and
 * ~1e7 positions per plane: the PCR head, rpn.py:266-289).  Same maths and the same per-channel
 * finalisation entries as the row-major s2d_bn1d_* family; positions per plane must be a
 * multiple of 4.  stats/sums are [2C]: (sum x, sum x^2) resp. (sum g, sum g*x), g = dy*(y>0 if relu).
 */
size_t s2d_bncm_workspace_bytes(int batch, int c, int64_t positions);
int s2d_bncm_stats_f32(const float *x, int batch, int c, int64_t positions, float *stats, void *ws,
                       size_t ws_bytes, s2d_stream_t stream);
int s2d_bncm_apply_f32(const float *x, const float *scale, const float *shift, int relu, int batch,
                       int c, int64_t positions, float *y, s2d_stream_t stream);
int s2d_bncm_bwd_reduce_f32(const float *dy, const float *y, const float *x, int relu, int batch,
                            int c, int64_t positions, float *sums, void *ws, size_t ws_bytes,
                            s2d_stream_t stream);
int s2d_bncm_bwd_apply_f32(const float *dy, const float *y, const float *x, const float *a,
                           const float *b, const float *d, int relu, int batch, int c,
                           int64_t positions, float *dx, s2d_stream_t stream);
/* the two backward passes with the ReLU mask re-derived from x - y > 0 <=> fma(x, scale[c], shift[c]) > 0, the expression
 * s2d_bncm_apply_f32 evaluated, hence the same mask bit for bit - instead of reading the output planes a third time */
int s2d_bncm_bwd_reduce_x_f32(const float *dy, const float *x, const float *scale, const float *shift, int batch, int c, int64_t positions,
                              float *sums, void *ws, size_t ws_bytes, s2d_stream_t stream);
int s2d_bncm_bwd_apply_x_f32(const float *dy, const float *x, const float *scale, const float *shift, const float *a, const float *b,
                             const float *d, int batch, int c, int64_t positions, float *dx, s2d_stream_t stream);

/*
 * Dense 3x3 convolution, stride 1 (or 2: forward only), padding 0 or 1, on NHWC bf16 activations (the BEV neck blocks
 * and the CenterHead towers: det3d/models/necks/rpn.py:126-145, bbox_heads/center_head.py:209-232;
 * replaces the cuDNN call behind nn.Conv2d there).  Implicit GEMM on v_mfma_f32_16x16x32_bf16,
 * fp32 accumulate, bf16 output [N][Ho][Wo][cout].  cin and cout multiples of 64.
 * The weight ([cout][cin][3][3] fp32, torch layout) is packed once per call into the kernel's
 * LDS image (9*cin*cout bf16).  transpose_flip=1 packs the data-gradient operand: then (cin,cout)
 * = (forward cout, forward cin) and the forward kernel run on dY with pad 1 yields dX (a pad-0
 * forward first pads dY by one ring of zeros).  weight_nhwc=1: `weight` memory order is
 * [cout][3][3][cin] (torch channels_last).  zero_page: >= 16 zero bytes of device memory
 * (source of the border taps).
 */
int s2d_conv2d3x3_supported(int cin, int cout);
int s2d_conv2d3x3_pack_weights_bf16(const float *weight, int cin, int cout, int transpose_flip,
                                    int weight_nhwc, void *packed, s2d_stream_t stream);
/* stats_partial (optional, fp32 [tiles][2][cout], tiles = s2d_conv2d3x3_stats_tiles(...) for the same shape - the
 * pixel-tile height is chosen per launch): per pixel tile (sum, sum of squares) of the stored
 * outputs per channel — the statistics pass of a following batch norm, produced in the conv epilogue; finish it with
 * s2d_bn_partials_finalize_f32 (or s2d_bn_partials_sum_f32 -> all-reduce -> s2d_bn1d_finalize_fwd_f32). */
int64_t s2d_conv2d3x3_stats_tiles(int n_img, int h, int w, int cin, int cout, int pad, int stride);
/* output pixels per workgroup tile of the launch plan for this shape (= rows per BN-statistics slab) */
int s2d_conv2d3x3_tile_rows(int n_img, int h, int w, int cin, int cout, int pad, int stride);
int s2d_conv2d3x3_nhwc_bf16(const void *x, const void *packed_weight, const float *bias,
                            const void *zero_page, int n_img, int h, int w, int cin, int cout,
                            int pad, int stride, void *y, float *stats_partial, s2d_stream_t stream);
/* batch-norm statistics from per-tile partial sums [nblocks][2][c] (a conv epilogue's stats_partial) over n rows:
 * fused finalize (mean, invstd, scale, shift, running stats, batches_tracked) or the plain [2c](+count) sums */
int s2d_bn_partials_finalize_f32(const float *partial, int nblocks, int64_t n, int c, const float *gamma,
                                 const float *beta, float eps, float momentum, float *mean,
                                 float *invstd, float *scale, float *shift, float *running_mean,
                                 float *running_var, int64_t *batches_tracked, s2d_stream_t stream);
int s2d_bn_partials_sum_f32(const float *partial, int nblocks, int64_t n, int c, float *stats,
                            int write_count, s2d_stream_t stream);
/* the same with a workspace (s2d_bn_partials_sum_workspace_bytes, 0 = not needed): lists of more than 1536 rows (a producer that
 * writes one row per tile, e.g. s2d_convt3d_mfma_fwd_stats) are folded in two stages, fixed order */
size_t s2d_bn_partials_sum_workspace_bytes(int nblocks, int c);
int s2d_bn_partials_sum_ws_f32(const float *partial, int nblocks, int64_t n, int c, float *stats, int write_count, void *ws,
                               size_t ws_bytes, s2d_stream_t stream);
/* s2d_bn_partials_finalize_f32 with the same optional workspace (two-stage fold of lists longer than 1536 rows) */
int s2d_bn_partials_finalize_ws_f32(const float *partial, int nblocks, int64_t n, int c, const float *gamma, const float *beta, float eps,
                                    float momentum, float *mean, float *invstd, float *scale, float *shift, float *running_mean,
                                    float *running_var, int64_t *batches_tracked, void *ws, size_t ws_bytes, s2d_stream_t stream);

/* Batch-norm backward sums out of the data-gradient conv (r06; rpn.py:126-145 conv -> BatchNorm2d -> ReLU chains, rpn.py:186-253
 * conv -> BatchNorm2d -> GELU groups; replaces the reduction half of the cuDNN batch-norm backward there).  When the conv being
 * differentiated reads the output of y = act(z * scale + shift) (act: 0 none, 1 ReLU, 2 exact GELU), its data gradient IS that layer's
 * dY: the *_bnbwd forms of the conv entries take z (bf16 [n][h][w][cout of this launch]) and the layer's fp32 scale / shift and write,
 * next to dY, the per-tile sums (sum g, sum g z), g = dY act'(z scale + shift) over the stored bf16 dY, to bn_partial[tiles][2][cout]
 * (tiles = s2d_conv2d3x3_stats_tiles / s2d_conv2d1x1_stats_tiles of the same launch shape).  s2d_bn_partials_bwd_finalize_ws_f32 folds
 * them (fixed order) into dgamma, dbeta and the coefficients a, b, d of s2d_bnrow_bwd_apply_ld_bf16 - no pass over (dY, z) in between.
 * *_supported: the launch plan of the shape takes the kernel that carries the epilogue (stride 1). */
int s2d_conv2d3x3_bnbwd_supported(int cin, int cout, int pad, int stride);
int s2d_conv2d3x3_nhwc_bf16_bnbwd(const void *x, const void *packed_weight, const void *zero_page, int n_img, int h, int w, int cin, int cout,
                                  int pad, void *y, const void *bn_z, const float *bn_scale, const float *bn_shift, int bn_act,
                                  float *bn_partial, s2d_stream_t stream);
int s2d_conv2d1x1_bnbwd_supported(int cin, int cout);
int s2d_conv2d1x1_nhwc_bf16_bnbwd(const void *x, const void *packed_weight, const void *zero_page, int n_img, int h, int w, int cin, int cout,
                                  void *y, const void *bn_z, const float *bn_scale, const float *bn_shift, int bn_act, float *bn_partial,
                                  s2d_stream_t stream);
int s2d_bn_partials_bwd_finalize_ws_f32(const float *partial, int nblocks, int64_t n, int c, const float *gamma, const float *mean,
                                        const float *invstd, float *dgamma, float *dbeta, float *a, float *b, float *d, void *ws,
                                        size_t ws_bytes, s2d_stream_t stream);

/*
 * Row-major bf16 batch norm: nn.BatchNorm2d (+ the ReLU that follows it) on NHWC bf16 activations
 * viewed as [n = N*H*W rows][c] (BEV neck and head: rpn.py:126-145, center_head.py:209-232;
 * replaces the cuDNN batch-norm calls there).  c multiple of 8, <= 1024.  Statistics and all
 * per-channel vectors are fp32 and use the same finalisation as the s2d_bn1d_* family
 * (s2d_bn1d_finalize_fwd/bwd_f32 after an all-reduce of the split `stats` / `sums` vectors, or the
 * fused *_finalize entries on one GPU).  Also the batch norm of the bf16-storage sparse stack
 * ([n sites][c], scn.py:73-83) with `residual` added before the ReLU.  Backward: with y == NULL the
 * ReLU mask is recomputed from x, scale and shift (y > 0 <=> x*scale+shift > 0; only valid without a
 * residual), otherwise taken from y; dres (optional) receives the masked dy, i.e. the gradient of
 * the residual branch.  `relu` is an activation code: 0 none, 1 ReLU, 2 exact (erf) GELU - the conv-BN-GELU groups of the
 * S2D module (det3d/models/necks/rpn.py:186-253); the GELU derivative is always recomputed from x, scale and shift.
 */
size_t s2d_bnrow_workspace_bytes(int64_t n, int c);
/* stats: [2c] sums, plus the row count at stats[2c] when write_count (the [2c+1] vector SyncBN all-reduces) */
int s2d_bnrow_stats_bf16(const void *x, int64_t n, int c, float *stats, int write_count, void *ws,
                         size_t ws_bytes, s2d_stream_t stream);
int s2d_bnrow_stats_finalize_bf16(const void *x, int64_t n, int c, const float *gamma,
                                  const float *beta, float eps, float momentum, float *mean,
                                  float *invstd, float *scale, float *shift, float *running_mean,
                                  float *running_var, int64_t *batches_tracked, void *ws,
                                  size_t ws_bytes, s2d_stream_t stream);
int s2d_bnrow_apply_bf16(const void *x, const float *scale, const float *shift, const void *residual,
                         int relu, int64_t n, int c, void *y, s2d_stream_t stream);
/* sums_copy (optional): second copy of the [2c] sums — the local one feeds dgamma/dbeta, the copy is all-reduced */
int s2d_bnrow_bwd_reduce_bf16(const void *dy, const void *x, const void *y, const float *scale,
                              const float *shift, int relu, int64_t n, int c, float *sums,
                              float *sums_copy, void *ws, size_t ws_bytes, s2d_stream_t stream);
int s2d_bnrow_bwd_reduce_finalize_bf16(const void *dy, const void *x, const void *y,
                                       const float *scale, const float *shift, int relu, int64_t n, int c,
                                       const float *gamma, const float *mean, const float *invstd,
                                       float *dgamma, float *dbeta, float *a, float *b, float *d,
                                       void *ws, size_t ws_bytes, s2d_stream_t stream);
int s2d_bnrow_bwd_apply_bf16(const void *dy, const void *x, const void *y, const float *scale,
                             const float *shift, int relu, const float *a, const float *b,
                             const float *d, int64_t n, int c, void *dx, void *dres,
                             s2d_stream_t stream);
/* Leading-dimension variants (r04): y / dy may be a channel slice of a wider row-major bf16 tensor - y_ld / dy_ld = its row stride in
 * elements (a multiple of 8, >= c; the pointer is the slice's first element).  Lets the RPN's up-sampling branches write their
 * batch-norm outputs straight into the concatenated tensor and read their gradients out of its gradient: no torch.cat copy, no
 * contiguous copies of the gradient slices (rpn.py:156-171). */
int s2d_bnrow_apply_ld_bf16(const void *x, const float *scale, const float *shift, const void *residual, int relu, int64_t n, int c, void *y,
                            int y_ld, s2d_stream_t stream);
int s2d_bnrow_bwd_reduce_ld_bf16(const void *dy, int dy_ld, const void *x, const void *y, const float *scale, const float *shift, int relu,
                                 int64_t n, int c, float *sums, float *sums_copy, void *ws, size_t ws_bytes, s2d_stream_t stream);
int s2d_bnrow_bwd_reduce_finalize_ld_bf16(const void *dy, int dy_ld, const void *x, const void *y, const float *scale, const float *shift,
                                          int relu, int64_t n, int c, const float *gamma, const float *mean, const float *invstd,
                                          float *dgamma, float *dbeta, float *a, float *b, float *d, void *ws, size_t ws_bytes,
                                          s2d_stream_t stream);
int s2d_bnrow_bwd_apply_ld_bf16(const void *dy, int dy_ld, const void *x, const void *y, const float *scale, const float *shift, int relu,
                                const float *a, const float *b, const float *d, int64_t n, int c, void *dx, void *dres,
                                s2d_stream_t stream);

/*
 * bf16-storage sparse convolution ("s16" path): features and outputs are bf16 [n][c] in HBM,
 * fp32 accumulation on v_mfma_f32_16x16x32_bf16.  Same gather-map contract as s2d_spconv_fwd_f32
 * (nbr[k][o] = input row feeding output row o through kernel offset k, or -1); the data gradient is
 * the same entry run on dout with nbr_in (or the SubM map with flip=1) and the weight image packed
 * with transpose=1.  cin, cout in {16, 32, 64, 128}.  Two kernels sit behind these entries: shapes with 128 input or output
 * channels (and >= 64 on the other side) run the register-gather kernel (csrc/spconv_rg.hip: A fragments gathered straight
 * into MFMA operand registers, weight slab through LDS), the others the LDS-staged kernel (csrc/spconv_s16.hip).  The weight
 * image depends on the kernel and its launch tiling chosen from n_out, so pack and fwd must be given the same n_out.
 * zero_page: >= 16 zero bytes of device memory (what a missing neighbour reads in the LDS-staged kernel).
 */
int s2d_spconv_s16_supported(int cin, int cout);
size_t s2d_spconv_s16_packed_elems(int kvol, int cin, int cout);
int s2d_spconv_s16_pack_weights(const float *weight, int kvol, int cin, int cout, int transpose,
                                int flip, int64_t n_out, void *packed, s2d_stream_t stream);
/* both operands of a layer in one launch (the weight changes once per optimizer step): packed_fwd as above for a forward launch over
 * n_out_fwd rows, packed_dgrad = the [cout -> cin] operand (transpose = 1, flip = flip_dgrad) for the data-gradient launch over n_out_dgrad rows */
int s2d_spconv_s16_pack_weights_pair(const float *weight, int kvol, int cin, int cout, int flip_dgrad, int64_t n_out_fwd, int64_t n_out_dgrad,
                                     void *packed_fwd, void *packed_dgrad, s2d_stream_t stream);
/* weight gradient of the s16 path (in_feat, dout bf16; dweight fp32 [kvol][cin][cout]); workspace from
 * s2d_spconv_wgrad_workspace_bytes */
int s2d_spconv_s16_wgrad(const void *in_feat, int64_t n_in, const void *dout, const int32_t *nbr,
                         int64_t n_out, int kvol, int cin, int cout, float *dweight, void *ws,
                         size_t ws_bytes, s2d_stream_t stream);
int s2d_spconv_s16_fwd(const void *in_feat, int64_t n_in, const void *packed_weight,
                       const float *bias, const int32_t *nbr, int64_t n_out, int kvol, int cin,
                       int cout, const void *zero_page, void *out_feat, s2d_stream_t stream);
/* the same with the batch-norm statistics of the layer that follows from the epilogue: stats_partial = fp32
 * [s2d_spconv_s16_stats_tiles(n_out, kvol, cin, cout)][2][cout], per-workgroup (sum, sum of squares) of the stored bf16 rows, folded by
 * s2d_bn_partials_finalize_f32 / s2d_bn_partials_sum_f32 (replaces the separate statistics pass of spconv's conv -> BatchNorm1d pairs,
 * /root/reference/det3d/models/backbones/scn.py:60-90) */
int64_t s2d_spconv_s16_stats_tiles(int64_t n_out, int kvol, int cin, int cout);
int s2d_spconv_s16_fwd_stats(const void *in_feat, int64_t n_in, const void *packed_weight, const float *bias, const int32_t *nbr,
                             int64_t n_out, int kvol, int cin, int cout, const void *zero_page, void *out_feat, float *stats_partial,
                             s2d_stream_t stream);

/*
 * r06 - rows of a submanifold rulebook grouped by neighbour mask, and the implicit GEMM over the grouped rows.
 * (spconv.ops.get_indice_pairs / indice_conv; call sites /root/reference/det3d/models/backbones/scn.py:104-152)
 * s2d_rulebook_sort_by_mask: from the dense gather map nbr [kvol][n] of a rulebook (kvol <= 27) builds
 *   pmask [n] u32   neighbour mask (bit k = offset k present) of the rows in ascending mask order per chunk (stable radix sort: deterministic),
 *   perm  [n] i32   canonical row index of each sorted position,
 *   nbr_perm [kvol][n] i32 = nbr[k][perm[j]].
 * s2d_spconv_s16_fwd_sorted: the bf16-storage sparse conv (same packed weight image as s2d_spconv_s16_fwd for the layer) over the sorted
 * rows: every workgroup multiplies only the kernel offsets present in the union of its rows' masks, every wave only those present in its
 * 16-row tile; results are stored to the canonical rows perm[j], so out_feat (and stats_partial: the same
 * [s2d_spconv_s16_stats_tiles][2][cout] layout) are what s2d_spconv_s16_fwd_stats returns, up to the summation order of the statistics.
 * Supported: kvol 27, 64 -> 64 and 128 -> 128 channels (s2d_spconv_s16_sorted_supported).
 * OPT-IN: measured on the benchmark scene's rulebooks (profiles/r06_sparse_sorted_rows_ab.txt) the sorted form is 5-10 % SLOWER than the
 * plain kernel although it issues 37 % fewer MFMAs - the kernel is bound by the gathered rows, which skipping does not reduce.
 * s2d_spconv_s16_set_sorted_rows(1) (or S2D_RG_SORTED=1 in the environment) switches it on and returns the previous setting; it also
 * moves 64 -> 64 layers to the register-gather weight image, so packed images made before the switch must be rebuilt.
 */
int s2d_spconv_s16_set_sorted_rows(int on);
size_t s2d_rulebook_sort_workspace_bytes(int64_t n);
/* the sort stays inside chunks of this many canonical rows (the rows one XCD's workgroups of the consuming kernel process): ascending masks
 * within a chunk, chunks in canonical order */
int64_t s2d_rulebook_sort_chunk_rows(int64_t n);
int s2d_rulebook_sort_by_mask(const int32_t *nbr, int kvol, int64_t n, int32_t *perm, uint32_t *pmask, int32_t *nbr_perm, void *ws,
                              size_t ws_bytes, s2d_stream_t stream);
int s2d_spconv_s16_sorted_supported(int kvol, int cin, int cout);
int s2d_spconv_s16_fwd_sorted(const void *in_feat, int64_t n_in, const void *packed_weight, const float *bias, const int32_t *nbr_perm,
                              const int32_t *perm, const uint32_t *pmask, int64_t n_out, int kvol, int cin, int cout, void *out_feat,
                              float *stats_partial, s2d_stream_t stream);

/*
 * Weight gradient of the dense 3x3 stride-1 convolution above (replaces the cuDNN backward-filter call):
 * x [n][h][w][cin] bf16 (the forward input), dy [n][ho][wo][cout] bf16, dweight fp32 in the torch layout
 * [cout][cin][3][3].  cin, cout multiples of 64.  Pixels are contracted on v_mfma_f32_16x16x32_bf16 through LDS
 * transpose reads; split partial sums are reduced in a fixed order (deterministic).
 */
int s2d_conv2d3x3_wgrad_supported(int cin, int cout);
size_t s2d_conv2d3x3_wgrad_workspace_bytes(int n_img, int h, int w, int cin, int cout, int pad);
int s2d_conv2d3x3_wgrad_nhwc_bf16(const void *x, const void *dy, const void *zero_page, int n_img,
                                  int h, int w, int cin, int cout, int pad, float *dweight, float *dbias /* [cout] or NULL: per-channel
                                  sums of dy = nn.Conv2d's bias gradient, accumulated by the same launch */, void *ws,
                                  size_t ws_bytes, s2d_stream_t stream);

/*
 * 1x1 convolutions (stride 1) of the S2D module (det3d/models/necks/rpn.py:186-253: fusion_sparse / fusion_dense, out_conv, the ConvNeXt
 * point-wise pairs) on NHWC bf16: GEMMs over the pixel rows on the same tile pipeline, weight image and epilogue (bias, per-tile
 * batch-norm statistics [tiles][2][cout], tiles = s2d_conv2d1x1_stats_tiles) as the 3x3 kernels.  pack: weight = torch
 * [Cout][Cin][1][1] of the forward conv, (cin, cout) = dimensions of the packed operand, transpose = 1 for the data gradient (run the
 * forward entry on dY).  Channels: multiples of 64.
 */
int s2d_conv2d1x1_pack_weights_bf16(const float *weight, int cin, int cout, int transpose, void *packed, s2d_stream_t stream);
/* both operands of a layer in one launch (the weight changes once per optimizer step, every training step needs both): weight = torch
 * [cout][cin][k][k] of the forward conv; packed_fwd = its [cin -> cout] image, packed_dgrad = the [cout -> cin] image with mirrored taps */
int s2d_conv2d3x3_pack_weights_pair_bf16(const float *weight, int cin, int cout, int weight_nhwc, void *packed_fwd, void *packed_dgrad,
                                         s2d_stream_t stream);
int s2d_conv2d1x1_pack_weights_pair_bf16(const float *weight, int cin, int cout, void *packed_fwd, void *packed_dgrad, s2d_stream_t stream);
int64_t s2d_conv2d1x1_stats_tiles(int n_img, int h, int w, int cin, int cout);
int s2d_conv2d1x1_nhwc_bf16(const void *x, const void *packed_weight, const float *bias, const void *zero_page, int n_img,
                            int h, int w, int cin, int cout, void *y, float *stats_partial, s2d_stream_t stream);
size_t s2d_conv2d1x1_wgrad_workspace_bytes(int n_img, int h, int w, int cin, int cout);
int s2d_conv2d1x1_wgrad_nhwc_bf16(const void *x, const void *dy, const void *zero_page, int n_img, int h, int w, int cin,
                                  int cout, float *dweight, float *dbias /* [cout] or NULL */, void *ws, size_t ws_bytes, s2d_stream_t stream);

/* 3x3 / padding 1 / stride 1 convolutions with 1..4 output channels: the last conv of every CenterHead branch
 * (/root/reference/det3d/models/bbox_heads/center_head.py:33-61 SepHead, Conv2d(64, classes, 3, padding=1); replaces the
 * torch.nn.Conv2d -> MIOpen call there).  x, dx: bf16 NHWC [n][h][w][cin] (cin in {8,16,32,64,128}); weight / dweight: fp32
 * [cout][cin][3][3] (torch layout); y, dy: fp32 planar [n][cout][h][w] - the predictions the losses and the decoder read.
 * wgrad also returns dbias (optional); its per-block partial sums are folded in a fixed order (deterministic). */
int s2d_smallconv3x3_supported(int cin, int cout);
int s2d_smallconv3x3_fwd(const void *x, const float *weight, const float *bias, int n_img, int h, int w, int cin, int cout, float *y,
                         s2d_stream_t stream);
int s2d_smallconv3x3_dgrad(const float *dy, const float *weight, int n_img, int h, int w, int cin, int cout, void *dx,
                           s2d_stream_t stream);
size_t s2d_smallconv3x3_wgrad_workspace_bytes(int cin, int cout);
int s2d_smallconv3x3_wgrad(const void *x, const float *dy, int n_img, int h, int w, int cin, int cout, float *dweight, float *dbias,
                           void *ws, size_t ws_bytes, s2d_stream_t stream);

/*
 * 2x2 stride-2 convolution forward (encoder_1[0] of the S2D module, rpn.py:188) on the same tile pipeline; weight = torch
 * [Cout][Cin][2][2] (weight_nhwc = 1: channels_last memory), output [n][h/2][w/2][cout], optional BN-statistics slabs.
 */
int s2d_conv2d2x2s2_pack_weights_bf16(const float *weight, int cin, int cout, int weight_nhwc, void *packed, s2d_stream_t stream);
int64_t s2d_conv2d2x2s2_stats_tiles(int n_img, int h, int w);
int s2d_conv2d2x2s2_nhwc_bf16(const void *x, const void *packed_weight, const float *bias, const void *zero_page, int n_img,
                              int h, int w, int cin, int cout, void *y, float *stats_partial, s2d_stream_t stream);

/*
 * Stride-2 transposed convolutions and their gradients on the same tile pipeline (replace the MIOpen calls behind
 * nn.ConvTranspose2d(256, 256, 4, 2, 1) in decoder_1 / decoder_2 of the S2D module, det3d/models/necks/rpn.py:217-231, and behind
 * the backward of the RPN's stride-2 3x3 convs, rpn.py:126-133).
 *   convup (ks = 4): ConvTranspose2d(4,2,1) forward, x [n][h][w][kc] -> y [n][2h][2w][nc], weight fp32 [kc][nc][4][4] (torch layout),
 *          optional bias and BN-statistics slabs [s2d_convup_stats_tiles][2][nc];
 *   convup (ks = 3): data gradient of Conv2d(nc -> kc, 3, stride 2, pad 1): dY [n][h][w][kc] -> dX [n][2h][2w][nc], weight [kc][nc][3][3];
 *   conv2d4x4s2: Conv2d(cin -> cout, 4, stride 2, pad 1) = data gradient of ConvTranspose2d(cout -> cin, 4,2,1); weight fp32
 *          [cout][cin][4][4] in the forward-conv sense = the transposed conv's weight as stored; x [n][h][w][cin] -> [n][h/2][w/2][cout];
 *   conv2d_s2_wgrad: dW[ca][cb][ks][ks] = sum a[n][i][j][.] * b[n][2i-1+ky][2j-1+kx][.] with a [n][h/2][w/2][ca], b [n][h][w][cb]
 *          (ConvTranspose2d: a = input, b = dY; stride-2 conv: a = dY, b = input); fixed-order split reduction in ws.
 */
int s2d_convup_supported(int kc, int nc, int ks);
int s2d_convup_pack_weights_bf16(const float *weight, int kc, int nc, int ks, void *packed, s2d_stream_t stream);
int64_t s2d_convup_stats_tiles(int n_img, int h, int w, int kc, int nc, int ks);
int s2d_convup_nhwc_bf16(const void *x, const void *packed_weight, const float *bias, const void *zero_page, int n_img, int h, int w,
                         int kc, int nc, int ks, void *y, float *stats_partial, s2d_stream_t stream);
int s2d_conv2d4x4s2_pack_weights_bf16(const float *weight, int cin, int cout, void *packed, s2d_stream_t stream);
int s2d_conv2d4x4s2_nhwc_bf16(const void *x, const void *packed_weight, const void *zero_page, int n_img, int h, int w, int cin,
                              int cout, void *y, s2d_stream_t stream);
int s2d_conv2d_s2_wgrad_supported(int ca, int cb, int ks);
size_t s2d_conv2d_s2_wgrad_workspace_bytes(int n_img, int h, int w, int ca, int cb, int ks);
int s2d_conv2d_s2_wgrad_nhwc_bf16(const void *a, const void *b, const void *zero_page, int n_img, int h, int w, int ca, int cb, int ks,
                                  float *dweight, void *ws, size_t ws_bytes, s2d_stream_t stream);

/*
 * Weight gradient of a bias-free Linear over a long row-major fp32 matrix (PFNLayer.linear of the pillar reader,
 * det3d/models/readers/pillar_encoder.py:41-56; replaces the GEMM behind torch.mm in nn.Linear's backward):
 *   dweight[co][ci] = sum_r dy[r][co] * x[r][ci];  x [rows][ci], dy [rows][co]; ci, co <= 64; exact fp32 (v_mfma_f32_16x16x4_f32),
 *   fixed-order split reduction in ws (s2d_rows_wgrad_workspace_bytes).
 */
int s2d_rows_wgrad_supported(int ci, int co);
size_t s2d_rows_wgrad_workspace_bytes(int64_t rows, int ci, int co);
int s2d_rows_wgrad_f32(const float *x, const float *dy, int64_t rows, int ci, int co, float *dweight, void *ws, size_t ws_bytes,
                       s2d_stream_t stream);

/*
 * Depth-wise 7x7 convolution, padding 3, stride 1 (nn.Conv2d(C, C, 7, padding=3, groups=C): first layer of the three
 * ConvNeXt blocks of the S2D module, det3d/models/necks/rpn.py:204-225) on NHWC bf16 maps.  x, y [n][h][w][c] bf16;
 * weight fp32 [c][49] (the torch layout [c][1][7][7]); bias fp32 [c] or NULL; fp32 accumulation.  flip=1 mirrors the taps:
 * the data gradient is the same stencil applied to dy.  c must be a multiple of 8.  The weight gradient returns dweight
 * fp32 [c][49] and (optionally) dbias [c]; its partial sums are folded in a fixed order (deterministic).
 */
int s2d_dwconv7_supported(int channels);
int s2d_dwconv7_nhwc_bf16(const void *x, const float *weight, const float *bias, int n_img, int h, int w,
                          int c, int flip, void *y, s2d_stream_t stream);
size_t s2d_dwconv7_wgrad_workspace_bytes(int n_img, int h, int w, int c);
int s2d_dwconv7_wgrad_nhwc_bf16(const void *x, const void *dy, int n_img, int h, int w, int c,
                                float *dweight, float *dbias, void *ws, size_t ws_bytes,
                                s2d_stream_t stream);

/*
 * ConvTranspose3d(kernel 4, stride 2, padding 1) of the PCR head on the bf16 matrix cores (bf16 compute mode;
 * det3d/models/necks/rpn.py:263-296: 32->32 and 16->3).  Tensors are NCDHW fp32 exactly as in s2d_convt3d_k4s2p1_*_f32;
 * activations and weights are rounded to bf16 while staged, accumulation is fp32.  cin in {16, 32}, cout <= 32.
 * `packed` = s2d_convt3d_mfma_packed_elems bf16 elements holding the forward and data-gradient weight images (rebuild it
 * whenever the weight changes).
 */
int s2d_convt3d_mfma_supported(int cin, int cout);
size_t s2d_convt3d_mfma_packed_elems(int cin, int cout);
int s2d_convt3d_mfma_pack_weights(const float *weight, int cin, int cout, void *packed, s2d_stream_t stream);
int s2d_convt3d_mfma_fwd(const float *in, const void *packed, const float *bias, int batch, int cin, int cout,
                         int d, int h, int w, float *out, s2d_stream_t stream);
/* ... with the statistics pass of the BatchNorm3d that follows folded into the epilogue: stats_partial (optional)
 * [s2d_convt3d_mfma_stats_tiles(batch,cin,d,h,w)][2][cout] = per-block (sum, sum of squares) per output channel of the written
 * output; s2d_bn_partials_sum_f32 folds them into the [2*cout] statistics vector */
int64_t s2d_convt3d_mfma_stats_tiles(int batch, int cin, int d, int h, int w);
int s2d_convt3d_mfma_fwd_stats(const float *in, const void *packed, const float *bias, int batch, int cin, int cout,
                               int d, int h, int w, float *out, float *stats_partial, s2d_stream_t stream);
int s2d_convt3d_mfma_dgrad(const float *dout, const void *packed, int batch, int cin, int cout, int d, int h,
                           int w, float *din, s2d_stream_t stream);
size_t s2d_convt3d_mfma_wgrad_workspace_bytes(int batch, int cin, int cout, int d, int h, int w);
int s2d_convt3d_mfma_wgrad(const float *in, const float *dout, int batch, int cin, int cout, int d, int h,
                           int w, float *dweight, void *ws, size_t ws_bytes, s2d_stream_t stream);
/* bf16-stored output gradient (r04): the same data / weight gradients from a dout the producer wrote in bf16 - the kernels above round
 * dout to bf16 on load, so the results are identical and the 0.5-0.7 GB tensor is half as large.  _supported: the layer shapes the
 * bf16-reading kernels cover (d, h, w = INPUT extents). */
int s2d_convt3d_mfma_d16_supported(int cin, int cout, int d, int h, int w);
int s2d_convt3d_mfma_dgrad_d16(const void *dout_bf16, const void *packed, int batch, int cin, int cout, int d, int h, int w, float *din,
                               s2d_stream_t stream);
int s2d_convt3d_mfma_wgrad_d16(const float *in, const void *dout_bf16, int batch, int cin, int cout, int d, int h, int w, float *dweight,
                               void *ws, size_t ws_bytes, s2d_stream_t stream);
/* The BatchNorm3d + ReLU in FRONT of the layer folded into its kernels (r04): `in` is the raw tensor, in_scale_shift = scale[cin] |
 * shift[cin] (device), applied as relu(fma(x, scale, shift)) - the expression of s2d_bncm_apply_f32 - while the forward stages its input
 * and while the weight gradient loads it; the normalised tensor is never written or read. */
int s2d_convt3d_mfma_norm_supported(int cin, int cout, int d, int h, int w);   /* the layers it pays on: narrow outputs (cout <= 4, w % 8 == 0) */
int s2d_convt3d_mfma_fwd_stats_y16_norm(const float *in, const float *in_scale_shift, const void *packed, const float *bias, int batch,
                                        int cin, int cout, int d, int h, int w, void *out_bf16, float *stats_partial, s2d_stream_t stream);
int s2d_convt3d_mfma_wgrad_d16_norm(const float *in, const float *in_scale_shift, const void *dout_bf16, int batch, int cin, int cout, int d,
                                    int h, int w, float *dweight, void *ws, size_t ws_bytes, s2d_stream_t stream);

/* weight (+ bias) gradient of the 1x1x1 Conv3d layers of the PCR head: dweight[cout][cin] = sum_{n,p} dout[n][co][p] in[n][ci][p],
 * dbias[cout] = sum dout (NCDHW fp32 tensors, positions % 4 == 0); deterministic two-stage reduction */
size_t s2d_pointwise_conv_wgrad_workspace_bytes(int cin, int cout);
int s2d_pointwise_conv_wgrad_f32(const float *in, const float *dout, int batch, int cin, int cout,
                                 int64_t positions, float *dweight, float *dbias, void *ws, size_t ws_bytes,
                                 s2d_stream_t stream);
/* same contract (and workspace) with both operands rounded to bf16 and contracted on the matrix cores in ONE pass over the two
 * tensors; supported: (cin 128, cout <= 32), (cin 32, cout <= 16), positions % 4 == 0, one sample's planes < 2 GB */
int s2d_pointwise_conv_wgrad_bf16_supported(int cin, int cout, int64_t positions);
int s2d_pointwise_conv_wgrad_bf16(const float *in, const float *dout, int batch, int cin, int cout,
                                 int64_t positions, float *dweight, float *dbias, void *ws, size_t ws_bytes,
                                 s2d_stream_t stream);
/* ... with x = relu(in*scale + shift) applied on the fly (the conv's input was a batch norm + ReLU of `in`):
 * in_scale_shift (device, 2*cin) = scale[cin] | shift[cin], NULL = plain */
int s2d_pointwise_conv_wgrad_norm_bf16(const float *in, const float *in_scale_shift, const float *dout, int batch,
                                       int cin, int cout, int64_t positions, float *dweight, float *dbias,
                                       void *ws, size_t ws_bytes, s2d_stream_t stream);

/*
 * PCR (point-cloud reconstruction) losses of the S2D student, det3d/models/detectors/voxelnet.py:171-185,203-249
 * (`mask_offset_loss` on the dense reconstruction target), computed from the SPARSE recon voxels instead of the dense
 * [B,5,D,H,W] target: gen_offset fp32 [B][3][D][H][W], gen_mask fp32 [B][1][D][H][W] (logits), coors int32 [m][4] (b,z,y,x)
 * and feats fp32 [m][5] = the reader output of the recon voxels at this scale.  out8 (device, 8 floats):
 * [0] mask_loss (BCE-with-logits, pos_weight = #neg/#pos, mean over all cells)  [1] offset_loss (L1 at the non-zero target
 * entries)  [2] beta  [3] n_sel  [4] N  [5] n_pos.  The backward takes out8 and the two upstream scalar gradients (device)
 * and writes d/d gen_mask (every cell) and d/d gen_offset (only the selected entries: the caller passes a zeroed buffer).
 */
size_t s2d_pcr_loss_workspace_bytes(void);
int s2d_pcr_loss_fwd_f32(const float *gen_offset, const float *gen_mask, const int32_t *coors,
                         const float *feats, int64_t m, int batch, int d, int h, int w, float *out8,
                         void *ws, size_t ws_bytes, s2d_stream_t stream);
int s2d_pcr_loss_bwd_f32(const float *gen_offset, const float *gen_mask, const int32_t *coors,
                         const float *feats, int64_t m, int batch, int d, int h, int w,
                         const float *fwd_out8, const float *go_mask, const float *go_offset,
                         float *g_gen_mask, float *g_gen_offset_zeroed, s2d_stream_t stream);

/*
 * Layout + precision hand-over between the bf16 NHWC neck and the fp32 planar PCR head (rpn.py:283-285): tiled transposes,
 * x bf16 [batch][hw][c] <-> y fp32 [batch][c][hw].  c % 8 == 0, hw % 4 == 0.
 */
int s2d_nhwc_bf16_to_nchw_f32(const void *x, int batch, int c, int64_t hw, float *y, s2d_stream_t stream);
int s2d_nchw_f32_to_nhwc_bf16(const float *x, int batch, int c, int64_t hw, void *y, s2d_stream_t stream);
/* r05: the same hand-overs against NHWC rows of `ld` >= c channels (the first c are used; the writer zero-fills the rest): the PCR head's first
 * 1x1x1 conv (necks/rpn.py:263-266 `generator_1[0]` on `out_conv(F_S_b).view(n, 128, 5, h, w)`) runs as ONE 1x1 conv on the NHWC map with a
 * block-diagonal weight whose output channel count is padded to the tile kernels' multiple of 64 */
int s2d_nhwc_bf16_to_nchw_f32_ld(const void *x, int batch, int c, int ld, int64_t hw, float *y, s2d_stream_t stream);
int s2d_nchw_f32_to_nhwc_bf16_ld(const float *x, int batch, int c, int ld, int64_t hw, void *y, s2d_stream_t stream);

/* 2 x 2 resampling of NHWC bf16 maps (r04, the pillar S2D module): nn.Upsample(scale_factor=2, mode="nearest") and nn.MaxPool2d(2, 2) with
 * their backward passes; c % 8 == 0.  The max-pool backward re-derives the selected window element from x (torch's scan order, NaN
 * propagating) - no index tensor. */
int s2d_upsample2x_nhwc_bf16(const void *x, int n, int h, int w, int c, void *y, s2d_stream_t stream);
int s2d_upsample2x_bwd_nhwc_bf16(const void *dy, int n, int h, int w, int c, void *dx, s2d_stream_t stream);
/* 2 x 2 space-to-depth on NHWC bf16 (r06): x[n][2h][2w][c] -> y[n][h][w][(py, px, c)] (to_depth != 0), or its inverse (to_depth == 0: x is the
 * depth image, y the space image); h, w = the extent of the depth image, c % 8 == 0.  The rearrangements around nn.Conv2d(kernel 2, stride 2)
 * and nn.ConvTranspose2d(kernel 2, stride 2) when those run as 1x1 tile kernels (/root/reference/det3d/models/necks/rpn.py:188 and :92-104;
 * replaces torch's permute + contiguous copies there). */
int s2d_space_depth2_nhwc_bf16(const void *x, int n, int h, int w, int c, int to_depth, void *y, s2d_stream_t stream);
int s2d_maxpool2x2_nhwc_bf16(const void *x, int n, int h, int w, int c, void *y, s2d_stream_t stream);
int s2d_maxpool2x2_bwd_nhwc_bf16(const void *x, const void *dy, int n, int h, int w, int c, void *dx, s2d_stream_t stream);

/*
 * LayerNorm over a whole [C,H,W] map per sample (nn.LayerNorm([256,47,47]) of the S2D ConvNeXt blocks, det3d/models/necks/rpn.py:
 * 210-247): each row is split over many workgroups (two-level fixed-order reduction).  x / y / dy / dx: bf16 [batch][row] in
 * MEMORY order; weight / bias and their gradients: fp32 [row] in that same order (the caller permutes); row % 8 == 0;
 * stats [batch][2] = (mean, rstd).  bwd: dx, dweight, dbias are each optional.
 */
size_t s2d_lnwide_workspace_bytes(int batch);
int s2d_lnwide_fwd_bf16(const void *x, const float *weight, const float *bias, int batch, int64_t row, float eps,
                        void *y, float *stats, void *ws, size_t ws_bytes, s2d_stream_t stream);
int s2d_lnwide_bwd_bf16(const void *dy, const void *x, const float *weight, const float *stats, int batch,
                        int64_t row, void *dx, float *dweight, float *dbias, void *ws, size_t ws_bytes,
                        s2d_stream_t stream);

/* CenterHead losses on the device maps (csrc/center_loss.hip), one pass per direction; they replace the torch-op chains of
 * FastFocalLoss / RegLoss (/root/reference/det3d/models/losses/centernet_loss.py:33-54 and :9-31; call site
 * /root/reference/det3d/models/bbox_heads/center_head.py:236-283).  out / target / feat: fp32 [batch][c][hw] contiguous; ind, cat: int64
 * [batch][max_objs]; mask: uint8 [batch][max_objs]; reg target: fp32 [batch][max_objs][channels].
 * focal res (device, 4 floats): loss = -(pos + neg) / max(num_pos, 1), pos, neg, num_pos.  regloss res (device, channels + 1 floats):
 * loss per channel, then sum(mask) + 1e-4.  The backward entries scale by the device scalar(s) `go`. */
size_t s2d_focal_workspace_bytes(void);
int s2d_focal_fwd(const float *out, const float *target, const int64_t *ind, const uint8_t *mask, const int64_t *cat, int batch, int classes,
                  int64_t hw, int max_objs, float *res, void *ws, size_t ws_bytes, s2d_stream_t stream);
int s2d_focal_bwd(const float *out, const float *target, const int64_t *ind, const uint8_t *mask, const int64_t *cat, int batch, int classes,
                  int64_t hw, int max_objs, const float *res, const float *go, float *dout, s2d_stream_t stream);
int s2d_regloss_fwd(const float *feat, const int64_t *ind, const uint8_t *mask, const float *target, int batch, int channels, int64_t hw,
                    int max_objs, float *res, s2d_stream_t stream);
int s2d_regloss_bwd(const float *feat, const int64_t *ind, const uint8_t *mask, const float *target, int batch, int channels, int64_t hw,
                    int max_objs, const float *res, const float *go, float *dfeat, s2d_stream_t stream);

/*
 * Feature-distillation loss of the S2D step (det3d/torchie/trainer/trainer.py:783-789): w_pos * MSE over teacher > 0 + w_neg * MSE
 * over the rest, between two dense tensors of n elements (n % 8 == 0) in the SAME memory order, bf16 (flag 1) or fp32 (0) each.
 * out4 (device) = loss, 2*w_pos/n_pos, 2*w_neg/n_neg, n_pos.  bwd: dstudent (student's element type) = go * dloss/dstudent.
 */
size_t s2d_masked_mse_workspace_bytes(void);
int s2d_masked_mse_fwd(const void *student, int student_bf16, const void *teacher, int teacher_bf16, int64_t n, float w_pos,
                       float w_neg, float *out4, void *ws, size_t ws_bytes, s2d_stream_t stream);
int s2d_masked_mse_bwd(const void *student, int student_bf16, const void *teacher, int teacher_bf16, int64_t n,
                       const float *fwd_out4, const float *go, void *dstudent, s2d_stream_t stream);

/*
 * Fused PCR level heads + losses: gen_mask_k / gen_out_k (1x1x1 Conv3d C->1 / C->3, det3d/models/necks/rpn.py:273-275,292-294)
 * and mask_offset_loss (voxelnet.py:171-185) evaluated straight from the level's feature volume g[B][C][D*H*W] - the occupancy
 * logits, the offset volume and its zero-filled gradient are never written; the offset conv runs at the m recon voxels only.
 * head_params (device, 4C+4 floats) = w_mask[C] | w_off[3][C] | b_mask | b_off[3].  out8 as s2d_pcr_loss_fwd_f32.
 * bwd writes dg[B][C][cells] = w_mask*dL/dlogit (+ w2^T.dz when co > 0: the data gradient of the level's next 1x1x1 conv
 * g -> z[B][co][cells], w2[co][C]) plus the sparse corrections, and dw_mask[C], db_mask[1], dw_off[3][C], db_off[3].
 * Supported: (C, co) in {(32,0), (32,16), (3,0)}, cells % 4 == 0.
 */
int s2d_pcr_heads_supported(int c, int co, int64_t cells);
size_t s2d_pcr_heads_workspace_bytes(int c);
int s2d_pcr_heads_fwd_f32(const float *g, const float *head_params, const int32_t *coors, const float *feats,
                          int64_t m, int batch, int c, int d, int h, int w, float *out8, void *ws,
                          size_t ws_bytes, s2d_stream_t stream);
int s2d_pcr_heads_bwd_f32(const float *g, const float *head_params, const int32_t *coors, const float *feats,
                          int64_t m, int batch, int c, int d, int h, int w, const float *fwd_out8,
                          const float *go_mask, const float *go_offset, const float *dz, const float *w2, int co,
                          float *dg, float *dw_mask, float *db_mask, float *dw_off, float *db_off, void *ws,
                          size_t ws_bytes, s2d_stream_t stream);

/*
 * The same level with the preceding BatchNorm3d + ReLU folded in (rpn.py:265-272,287-291): y is the RAW ConvTranspose3d output,
 * g = relu(y*scale + shift) is applied on the fly (bn_scale_shift, device, 2C = scale[C] | shift[C]); neither g, its gradient nor
 * the masked batch-norm gradient is written.  fwd also writes z[B][co][cells] = w2.g + b2 when co > 0.  The backward is two
 * passes around the batch norm's own finalisation: bwd_sums -> grads[4C+4] = dw_mask | dw_off | db_mask | db_off and bn_sums[2C] =
 * (sum dG*m, sum dG*m*y) (what s2d_bncm_bwd_reduce_f32 produces); bwd_apply -> dy = a*dG*m + b*y + d with abd (device, 3C).
 */
size_t s2d_pcr_level_workspace_bytes(int c);
int s2d_pcr_level_fwd_f32(const float *y, const float *bn_scale_shift, const float *head_params, const float *w2,
                          const float *b2, const int32_t *coors, const float *feats, int64_t m, int batch, int c,
                          int co, int d, int h, int w, float *z, float *z_stats /* [2 co] sum | sumsq of z per channel (the
                          statistics of the BatchNorm3d behind the 1x1x1 conv; c = 32, co = 16 only), or NULL */, float *out8,
                          void *ws, size_t ws_bytes, s2d_stream_t stream);
int s2d_pcr_level_bwd_sums_f32(const float *y, const float *bn_scale_shift, const float *head_params,
                               const int32_t *coors, const float *feats, int64_t m, int batch, int c, int d, int h,
                               int w, const float *fwd_out8, const float *go_mask, const float *go_offset,
                               const float *dz, const float *w2, int co, float *grads, float *bn_sums, void *ws,
                               size_t ws_bytes, s2d_stream_t stream);
int s2d_pcr_level_bwd_apply_f32(const float *y, const float *bn_scale_shift, const float *head_params,
                                const int32_t *coors, const float *feats, int64_t m, int batch, int c, int d, int h,
                                int w, const float *fwd_out8, const float *go_mask, const float *go_offset,
                                const float *dz, const float *w2, int co, const float *abd, float *dy,
                                s2d_stream_t stream);

/*
 * Rotated bird's-eye-view IoU and greedy NMS of CenterHead.predict (det3d/core/bbox/box_torch_ops.py:449-464 rotate_nms_pcdet,
 * det3d/ops/iou3d_nms/src/iou3d_nms_kernel.cu:236-326, src/iou3d_nms.cpp:92-130).  Boxes: 7 floats (x,y,z,dx,dy,dz,heading).
 * s2d_nms_rotated_bev takes the boxes SORTED by descending score and writes the kept indices (into that order) and their count
 * to device memory - the suppression matrix is walked on the device, no host round trip.
 */
int s2d_bev_iou_f32(const float *boxes_a, int na, const float *boxes_b, int nb, float *iou, s2d_stream_t stream);
size_t s2d_nms_workspace_bytes(int n);
int s2d_nms_rotated_bev(const float *boxes_sorted, int n, float iou_threshold, int max_keep, int64_t *keep,
                        int32_t *n_keep, void *ws, size_t ws_bytes, s2d_stream_t stream);
/* CenterPoint's circle NMS (det3d/core/utils/circle_nms_jit.py:4-31, called from bbox_heads/center_head.py:476-479,499-507): greedy
 * suppression by centre distance - a later box is removed when (xi - xj)^2 + (yi - yj)^2 <= thresh (the reference compares the squared
 * distance with test_cfg.min_radius[task] as is).  xy_sorted: [n][2] centres sorted by descending score; same outputs and workspace
 * (s2d_nms_workspace_bytes) as s2d_nms_rotated_bev. */
int s2d_nms_circle(const float *xy_sorted, int n, float thresh, int max_keep, int64_t *keep, int32_t *n_keep, void *ws, size_t ws_bytes,
                   s2d_stream_t stream);

/*
 * CenterPoint training targets on the device = AssignLabel.__call__ (det3d/datasets/pipelines/preprocess.py:489-653, one-task
 * Waymo head) with gaussian_radius / draw_umich_gaussian (det3d/core/utils/center_utils.py:18-64).  gt_boxes fp32
 * [frames][max_boxes][9] (x,y,z,w,l,h,vx,vy,yaw), gt_classes int32 [frames][max_boxes] (1-based, <= 0 = padding).  Outputs per
 * frame: hm fp32 [num_classes][fmap_h][fmap_w] (ZEROED by the caller), anno_box fp32 [max_objs][10], ind int64 [max_objs],
 * mask uint8 [max_objs], cat int64 [max_objs], gt_boxes_and_cls fp32 [max_objs][10] (optional, two-stage code).
 */
int s2d_assign_label(const float *gt_boxes, const int32_t *gt_classes, int frames, int max_boxes,
                     const float pc_range_xy[2], const float voxel_size_xy[2], int out_size_factor, int fmap_w,
                     int fmap_h, int num_classes, int max_objs, double gaussian_overlap, int min_radius,
                     float *hm_zeroed, float *anno_box, int64_t *ind, uint8_t *mask, int64_t *cat,
                     float *gt_boxes_and_cls, s2d_stream_t stream);

/*
 * Optimizer step of the reference's training loop (det3d/torchie/apis/train.py:168-186, det3d/solver/fastai_optim.py:158-171,
 * hooks/optimizer.py:15-21): gradient L2 norm -> clip coefficient (device scalar, clip_grad_norm_ semantics) -> fused
 * multi-tensor Adam with decoupled weight decay: p *= 1 - lr*wd; Adam(betas, eps) on grad*clip_coef with bias correction of
 * step `step` (1-based).  Up to s2d_adam_max_tensors() tensors per call (host pointer tables of DEVICE pointers).
 */
int s2d_adam_max_tensors(void);
int s2d_adam_step_f32(int count, float *const *params, const float *const *grads, float *const *exp_avg,
                      float *const *exp_avg_sq, const int64_t *numel, float lr, float beta1, float beta2,
                      float eps, float weight_decay, int step, const float *clip_coef, s2d_stream_t stream);
size_t s2d_grad_norm_workspace_floats(int count, const int64_t *numel);
int s2d_grad_sumsq_f32(int count, const float *const *grads, const int64_t *numel, float *partial,
                       int *written, s2d_stream_t stream);
int s2d_grad_norm_finalize_f32(const float *partial, int n, float max_norm, float *out2, s2d_stream_t stream);

/*
 * SyncBN statistics all-reduce on the compute stream (det3d/torchie/apis/train.py:281-300: apex SyncBatchNorm + DDP when
 * training distributed).  The RCCL already loaded in the process is resolved at run time; s2d_comm_available() == 0 means
 * the host keeps using its own collective.  Bootstrap: rank 0 calls s2d_comm_unique_id (128 bytes), the host broadcasts
 * them, every rank calls s2d_comm_init (collective).  s2d_comm_allreduce_sum_f32: in-place sum of `count` floats over the
 * ranks, enqueued on `stream` (between a batch-norm reduction kernel and its finalize kernel).
 */
int s2d_comm_load_library(const char *path); /* optional: the RCCL shared object the host framework has loaded */
int s2d_comm_available(void);
int s2d_comm_unique_id(void *id128);
int s2d_comm_init(const void *id128, int nranks, int rank);
int s2d_comm_ranks(void);
int s2d_comm_shutdown(void);
int s2d_comm_allreduce_sum_f32(float *buf, int64_t count, s2d_stream_t stream);

/* r04: the weight images of up to 64 dense conv layers in ONE launch (the re-pack after an optimizer step; the reference has no such
 * step - its cuDNN/MIOpen convs read the fp32 parameters directly).  Per layer the arguments of s2d_conv2d{3x3,1x1}_pack_weights[_pair]_bf16;
 * packed_dgrad[i] = NULL packs a single image (with transpose_flip[i]).  Host arrays of device pointers / ints. */
int s2d_conv2d_pack_batch_bf16(int n, const float *const *weights, const int32_t *cin, const int32_t *cout, const int32_t *taps,
                               const int32_t *weight_nhwc, const int32_t *transpose_flip, void *const *packed_fwd, void *const *packed_dgrad,
                               s2d_stream_t stream);

/* ---- r04: bf16 storage of the raw PCR up-sampler outputs ---------------------------------------------------------------------
 * (det3d/models/necks/rpn.py:263-296: the outputs of the two ConvTranspose3d layers, 724 MB and 543 MB in fp32 at batch 4, are read by
 * four passes of the fused level each).  Same contracts as the fp32 entries named alike; only the element type of that tensor changes:
 * `out_bf16` / `y` / `in_bf16` are bf16 [B][C][cells]; every other tensor, the statistics and all accumulation stay fp32. */
int s2d_convt3d_mfma_fwd_stats_y16(const float *in, const void *packed, const float *bias, int batch, int cin, int cout, int d, int h, int w,
                                   void *out_bf16, float *stats_partial, s2d_stream_t stream);
int s2d_pointwise_conv_wgrad_norm_x16(const void *in_bf16, const float *in_scale_shift, const float *dout, int batch, int cin, int cout,
                                      int64_t positions, float *dweight, float *dbias, void *ws, size_t ws_bytes, s2d_stream_t stream);
/* r06: the 16-channel volume z between the two PCR levels ([B,16,10,376,376] at the benchmark: 362 MB in fp32), its gradient dz and the second
 * up-sampler's input gradient dx' STORED IN BF16 (fp32 arithmetic, sums and statistics everywhere): the twelve passes per step that cross one
 * of the three move half the bytes.  Same contracts as the entries they extend (rpn.py:263-296; replaces the same torch calls):
 *   s2d_pcr_level_fwd_y16_z16            = s2d_pcr_level_fwd_y16 writing z as bf16 (c = 32, co = 16; z_stats of the stored values)
 *   s2d_pcr_level_bwd_sums_y16_z16, s2d_pcr_level_bwd_apply_y16_d16_z16 = the level's backward passes reading a bf16 dz
 *   s2d_pointwise_conv_wgrad_norm_x16_d16 = s2d_pointwise_conv_wgrad_norm_x16 with a bf16 dout (cin = 32)
 *   s2d_convt3d_mfma_fwd_stats_y16_norm_x16, _wgrad_d16_norm_x16 = the up-sampler's forward / weight gradient reading a bf16 raw input
 *   s2d_convt3d_mfma_dgrad_d16_x16       = its data gradient written as bf16        (all three: s2d_convt3d_mfma_x16_supported shapes)
 *   s2d_bncm_bwd_reduce_x_typed / s2d_bncm_bwd_apply_x_typed = s2d_bncm_bwd_reduce_x_f32 / _apply_x_f32 with a storage flag per tensor
 *     (0 fp32, 1 bf16); combinations (x, dy, dx): all fp32 | (fp32, bf16, fp32) | all bf16 */
/* r06: per-voxel cache for the NEXT s2d_pcr_level_* call on the calling thread (any storage variant): the forward call fills row i with the c raw
 * values of y at recon voxel i's cell ([m][c] elements of y's type, [m][4] for c = 3), the level's two backward calls read their voxels from it
 * instead of c scattered loads per voxel.  buf: >= m * (c == 3 ? 4 : c) * sizeof(y element) bytes of device memory (smaller: ignored); consumed by
 * the next call, repeat per call; NULL / 0 clears.  (voxelnet.py:171-185: the per-voxel terms of the mask / offset losses.) */
int s2d_pcr_level_site_cache(void *buf, size_t bytes);
int s2d_pcr_level_fwd_y16_z16(const void *y, const float *bn_scale_shift, const float *head_params, const float *w2, const float *b2,
                              const int32_t *coors, const float *feats, int64_t m, int batch, int c, int co, int d, int h, int w, void *z_bf16,
                              float *z_stats, float *out8, void *ws, size_t ws_bytes, s2d_stream_t stream);
int s2d_pcr_level_bwd_sums_y16_z16(const void *y, const float *bn_scale_shift, const float *head_params, const int32_t *coors, const float *feats,
                                   int64_t m, int batch, int c, int d, int h, int w, const float *fwd_out8, const float *go_mask,
                                   const float *go_offset, const void *dz_bf16, const float *w2, int co, float *grads, float *bn_sums, void *ws,
                                   size_t ws_bytes, s2d_stream_t stream);
int s2d_pcr_level_bwd_apply_y16_d16_z16(const void *y, const float *bn_scale_shift, const float *head_params, const int32_t *coors,
                                        const float *feats, int64_t m, int batch, int c, int d, int h, int w, const float *fwd_out8,
                                        const float *go_mask, const float *go_offset, const void *dz_bf16, const float *w2, int co,
                                        const float *abd, void *dy_bf16, s2d_stream_t stream);
int s2d_pointwise_conv_wgrad_norm_x16_d16(const void *in_bf16, const float *in_scale_shift, const void *dout_bf16, int batch, int cin, int cout,
                                          int64_t positions, float *dweight, float *dbias, void *ws, size_t ws_bytes, s2d_stream_t stream);
int s2d_convt3d_mfma_x16_supported(int cin, int cout, int d, int h, int w);
int s2d_convt3d_mfma_fwd_stats_y16_norm_x16(const void *in_bf16, const float *in_scale_shift, const void *packed, const float *bias, int batch,
                                            int cin, int cout, int d, int h, int w, void *out_bf16, float *stats_partial, s2d_stream_t stream);
int s2d_convt3d_mfma_wgrad_d16_norm_x16(const void *in_bf16, const float *in_scale_shift, const void *dout_bf16, int batch, int cin, int cout,
                                        int d, int h, int w, float *dweight, void *ws, size_t ws_bytes, s2d_stream_t stream);
int s2d_convt3d_mfma_dgrad_d16_x16(const void *dout_bf16, const void *packed, int batch, int cin, int cout, int d, int h, int w, void *din_bf16,
                                   s2d_stream_t stream);
int s2d_bncm_bwd_reduce_x_typed(const void *dy, int dy_bf16, const void *x, int x_bf16, const float *scale, const float *shift, int batch, int c,
                                int64_t positions, float *sums, void *ws, size_t ws_bytes, s2d_stream_t stream);
int s2d_bncm_bwd_apply_x_typed(const void *dy, int dy_bf16, const void *x, int x_bf16, const float *scale, const float *shift, const float *a,
                               const float *b, const float *d, int batch, int c, int64_t positions, void *dx, int dx_bf16, s2d_stream_t stream);
int s2d_pcr_level_fwd_y16(const void *y, const float *bn_scale_shift, const float *head_params, const float *w2, const float *b2,
                          const int32_t *coors, const float *feats, int64_t m, int batch, int c, int co, int d, int h, int w, float *z,
                          float *z_stats, float *out8, void *ws, size_t ws_bytes, s2d_stream_t stream);
int s2d_pcr_level_bwd_sums_y16(const void *y, const float *bn_scale_shift, const float *head_params, const int32_t *coors, const float *feats,
                               int64_t m, int batch, int c, int d, int h, int w, const float *fwd_out8, const float *go_mask,
                               const float *go_offset, const float *dz, const float *w2, int co, float *grads, float *bn_sums, void *ws,
                               size_t ws_bytes, s2d_stream_t stream);
int s2d_pcr_level_bwd_apply_y16(const void *y, const float *bn_scale_shift, const float *head_params, const int32_t *coors, const float *feats,
                                int64_t m, int batch, int c, int d, int h, int w, const float *fwd_out8, const float *go_mask,
                                const float *go_offset, const float *dz, const float *w2, int co, const float *abd, float *dy,
                                s2d_stream_t stream);
/* bf16 y AND bf16 dy (the gradient's only readers, s2d_convt3d_mfma_{dgrad,wgrad}_d16, round it to bf16 anyway) */
int s2d_pcr_level_bwd_apply_y16_d16(const void *y, const float *bn_scale_shift, const float *head_params, const int32_t *coors, const float *feats,
                                    int64_t m, int batch, int c, int d, int h, int w, const float *fwd_out8, const float *go_mask,
                                    const float *go_offset, const float *dz, const float *w2, int co, const float *abd, void *dy_bf16,
                                    s2d_stream_t stream);

/* ---- fused PointPillars feature net (r04) ----------------------------------------------------------------------------------
 * One PFN layer of det3d/models/readers/pillar_encoder.py:41-56,114-154 (decorate -> Linear(10 -> 64) -> BatchNorm1d -> ReLU -> max over
 * the slots) from the raw pillars voxels[P][slots][5], num_points[P], coors[P][4] (b,z,y,x), recomputing the per-row products in every
 * pass instead of storing [P,slots,64] tensors.  _stats: per-workgroup partial slabs [s2d_pfn_blocks(P)][2][64] (fold with
 * s2d_bn_partials_sum_f32, n = P * slots rows).  _apply_max: out[P][64] and the slot of the maximum (uint8, first maximum).  _bwd: per-
 * workgroup rows of s2d_pfn_bwd_cols() floats: sum g[64] | sum g*h[64] | M1[10][64] | M2[10][64] | M3[10] (dout must already carry relu').
 */
int s2d_pfn_supported(int ndim, int slots, int feats, int cout);
int s2d_pfn_blocks(int64_t pillars);
int s2d_pfn_bwd_cols(void);
int s2d_pfn_stats_f32(const float *voxels, const int32_t *num_points, const int32_t *coors, const float *weight, int64_t pillars,
                      int slots, int ndim, float vx, float vy, float x_offset, float y_offset, float *partial, s2d_stream_t stream);
int s2d_pfn_apply_max_f32(const float *voxels, const int32_t *num_points, const int32_t *coors, const float *weight, const float *scale,
                          const float *shift, int64_t pillars, int slots, int ndim, float vx, float vy, float x_offset, float y_offset,
                          float *out, uint8_t *argmax, s2d_stream_t stream);
int s2d_pfn_bwd_f32(const float *voxels, const int32_t *num_points, const int32_t *coors, const float *weight, const float *dout,
                    const uint8_t *argmax, int64_t pillars, int slots, int ndim, float vx, float vy, float x_offset, float y_offset,
                    float *partial, s2d_stream_t stream);

/* Two PFN layers (the configs under configs/waymo/pp: num_filters = [64, 64]; layer 1 Linear(10 -> 32) -> BN -> ReLU -> [x | max over slots], layer 2
 * Linear(64 -> 64) -> BN -> ReLU -> max over slots; pillar_encoder.py:41-56).  scale_shift1 = scale[32] | shift[32] of the first batch
 * norm, scale_shift2 = scale[64] | shift[64] of the second.  _stats1 / _stats2: per-workgroup partial slabs [s2d_pfn_blocks(P)][2][64]
 * of (sum h, sum h^2) over the valid rows of layer 1 (columns 32..63 duplicate 0..31) / over all P*slots rows of layer 2.  _apply_max:
 * out[P][64], the slot of the maximum (first maximum; slot index num_points = an empty slot) and h2 at that slot.  _bwd: abd2 = a | b | d
 * of the second batch norm's backward (dh2 = a g + b h2 + d), gout = dout with relu' applied; writes s2d_pfn2_bwd_rows() rows of
 * s2d_pfn2_bwd_cols() floats: dW2[64][64] | sum g1[64] | sum g1 h1[64] | M1[10][64] | M2[10][64] | M3[10] (layer-1 entries in columns
 * 0..31), to be summed over the rows. */
int s2d_pfn2_supported(int ndim, int slots, int feats, int c1, int c2);
int s2d_pfn2_bwd_rows(void);
int s2d_pfn2_bwd_cols(void);
int s2d_pfn2_stats1_f32(const float *voxels, const int32_t *num_points, const int32_t *coors, const float *w1, int64_t pillars, int slots,
                        int ndim, float vx, float vy, float x_offset, float y_offset, float *partial, s2d_stream_t stream);
int s2d_pfn2_stats2_f32(const float *voxels, const int32_t *num_points, const int32_t *coors, const float *w1, const float *w2,
                        const float *scale_shift1, int64_t pillars, int slots, int ndim, float vx, float vy, float x_offset,
                        float y_offset, float *partial, s2d_stream_t stream);
int s2d_pfn2_apply_max_f32(const float *voxels, const int32_t *num_points, const int32_t *coors, const float *w1, const float *w2,
                           const float *scale_shift1, const float *scale_shift2, int64_t pillars, int slots, int ndim, float vx, float vy,
                           float x_offset, float y_offset, float *out, uint8_t *argmax, float *h2_at_max, s2d_stream_t stream);
int s2d_pfn2_bwd_f32(const float *voxels, const int32_t *num_points, const int32_t *coors, const float *w1, const float *w2,
                     const float *scale_shift1, const float *abd2, const float *gout, const uint8_t *argmax, int64_t pillars, int slots,
                     int ndim, float vx, float vy, float x_offset, float y_offset, float *partial, s2d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* S2D_H_ */
