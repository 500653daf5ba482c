/*
 * ORACLE (test infrastructure, never the product path).
 *
 * Plain-C restatement of the reference's hard voxelizer:
 *   /root/reference/det3d/ops/point_cloud/point_cloud_ops.py:7-55
 *     (_points_to_voxel_reverse_kernel: the sequential first-come-first-served loop)
 *   /root/reference/det3d/ops/point_cloud/point_cloud_ops.py:112-184
 *     (points_to_voxel: scratch allocation, trimming to voxel_num)
 * and of the mean reader
 *   /root/reference/det3d/models/readers/voxel_encoder.py:17-24
 *     (VoxelFeatureExtractorV3.forward: sum over the point slots / num_points)
 *
 * Pinned by tests/golden/voxelize_*.npz, which were produced by importing the reference
 * file itself (tests/golden/make_golden.py).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may call into this file.
 *
 * Build: see oracle/Makefile (gcc -O2 -shared -fPIC; NO -ffast-math: the coordinate
 * arithmetic must stay IEEE fp32 sub + div + floor exactly like numpy/numba).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/*
 * points      : [n_points, ndim] fp32 (x, y, z, extra...)
 * voxel_size  : [3] fp32 (x, y, z)
 * coors_range : [6] fp32 (xmin, ymin, zmin, xmax, ymax, zmax)
 * voxels      : [max_voxels, max_points, ndim] fp32, caller-zeroed
 * coors       : [max_voxels, 3] int32 (z, y, x)         ("reverse_index=True")
 * num_points  : [max_voxels] int32, caller-zeroed
 * returns voxel_num, or -1 on allocation failure.
 */
int s2d_oracle_points_to_voxel(const float *points, int64_t n_points, int ndim,
                               const float *voxel_size, const float *coors_range,
                               int max_points, int max_voxels,
                               float *voxels, int32_t *coors, int32_t *num_points)
{
    int32_t grid[3];
    for (int j = 0; j < 3; ++j) {
        /* point_cloud_ops.py:24-27: fp32 (hi-lo)/vs, np.round (half-to-even), cast */
        float g = (coors_range[3 + j] - coors_range[j]) / voxel_size[j];
        grid[j] = (int32_t)nearbyintf(g);
    }
    /* dense lookup grid indexed [z][y][x] (point_cloud_ops.py:144-150) */
    const int64_t cells = (int64_t)grid[0] * grid[1] * grid[2];
    int32_t *lut = (int32_t *)malloc(sizeof(int32_t) * (size_t)cells);
    if (!lut) return -1;
    memset(lut, 0xff, sizeof(int32_t) * (size_t)cells); /* -1 */

    int32_t voxel_num = 0;
    for (int64_t i = 0; i < n_points; ++i) {
        const float *p = points + i * ndim;
        int32_t c[3]; /* c[0]=z c[1]=y c[2]=x */
        int failed = 0;
        for (int j = 0; j < 3; ++j) {
            /* point_cloud_ops.py:36: np.floor((p - lo) / vs), all fp32 */
            volatile float d = p[j] - coors_range[j];
            volatile float q = d / voxel_size[j];
            float f = floorf(q);
            /* :37 comparisons are done on the float value (NaN compares false -> kept by the
             * reference and then crashes on the int cast; we reject NaN explicitly) */
            if (!(f >= 0.0f) || f >= (float)grid[j]) { failed = 1; break; }
            c[2 - j] = (int32_t)f;
        }
        if (failed) continue;
        int64_t cell = ((int64_t)c[0] * grid[1] + c[1]) * grid[0] + c[2];
        int32_t v = lut[cell];
        if (v == -1) {
            v = voxel_num;
            if (voxel_num >= max_voxels) continue; /* :46-47 new voxels dropped once full */
            voxel_num += 1;
            lut[cell] = v;
            coors[3 * v + 0] = c[0];
            coors[3 * v + 1] = c[1];
            coors[3 * v + 2] = c[2];
        }
        int32_t n = num_points[v];
        if (n < max_points) { /* :51-54 */
            memcpy(voxels + ((int64_t)v * max_points + n) * ndim, p, sizeof(float) * (size_t)ndim);
            num_points[v] = n + 1;
        }
    }
    free(lut);
    return voxel_num;
}

/*
 * voxel_encoder.py:20-22: features[:, :, :C].sum(dim=1) / num_points  (fp32).
 * The slot sum is accumulated in slot order 0..max_points-1 (zero padded slots add +0).
 */
void s2d_oracle_voxel_mean(const float *voxels, const int32_t *num_points, int64_t n_voxels,
                           int max_points, int ndim, int n_feat, float *mean_out)
{
    for (int64_t v = 0; v < n_voxels; ++v) {
        for (int c = 0; c < n_feat; ++c) {
            float s = 0.0f;
            for (int t = 0; t < max_points; ++t) {
                volatile float a = s + voxels[((int64_t)v * max_points + t) * ndim + c];
                s = a;
            }
            mean_out[v * n_feat + c] = s / (float)num_points[v];
        }
    }
}
