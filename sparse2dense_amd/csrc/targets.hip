// CenterPoint training targets on the device: AssignLabel.__call__ for the one-task Waymo head
// (/root/reference/det3d/datasets/pipelines/preprocess.py:489-653) with gaussian_radius / gaussian2D / draw_umich_gaussian of
// /root/reference/det3d/core/utils/center_utils.py:18-64 and box_np_ops.limit_period (det3d/core/bbox/box_np_ops.py:360-361).
// In the reference this is per-frame numpy in the DataLoader workers; at hundreds of frames/s per GPU the targets have to be
// produced where the frames already are.
//
// One thread per (frame, object slot k < max_objs):
//   yaw <- yaw - floor(yaw / 2pi + 0.5) * 2pi                                              (fp32, as numpy on the fp32 box array)
//   w, l in feature-map cells (fp32); radius = max(min_radius, int(gaussian_radius((l, w), overlap)))   (float64 roots)
//   ct = ((x - x0) / vx / f, (y - y0) / vy / f) fp32; ct_int = trunc(ct); skipped when outside the map
//   hm[cls] = max(hm[cls], gaussian)  on the (2r+1)^2 window clipped to the map        (float atomicMax: order independent)
//   ind = y*W + x, mask = 1, cat = cls, anno_box = (ct - ct_int, z, log(w,l,h), vx, vy, sin yaw, cos yaw)
//   gt_boxes_and_cls[k] = (x, y, z, w, l, h, yaw, vx, vy, class)                          (two-stage code, preprocess.py:626-649)
// Boxes are [frames][max_boxes][9] = (x,y,z,w,l,h,vx,vy,yaw) fp32, classes int32 (1-based; <= 0 = padding).  hm must be zeroed
// by the caller (the other outputs are fully written).
#include "s2d_common.h"

namespace s2d {

struct TgtGeo {
    float x0, y0, vx, vy;
    int factor, fw, fh, num_classes, max_objs, min_radius;
    double overlap;
};

__device__ __forceinline__ double tgt_gaussian_radius(double height, double width, double mo) {
    const double b1 = height + width, c1 = width * height * (1 - mo) / (1 + mo);
    const double r1 = (b1 + sqrt(b1 * b1 - 4 * c1)) / 2;
    const double b2 = 2 * (height + width), c2 = (1 - mo) * width * height;
    const double r2 = (b2 + sqrt(b2 * b2 - 16 * c2)) / 2;
    const double a3 = 4 * mo, b3 = -2 * mo * (height + width), c3 = (mo - 1) * width * height;
    const double r3 = (b3 + sqrt(b3 * b3 - 4 * a3 * c3)) / 2;
    return fmin(r1, fmin(r2, r3));
}

__global__ __launch_bounds__(256) void assign_label_kernel(const float *__restrict__ boxes, const int32_t *__restrict__ classes, int frames, int max_boxes,
                                                           TgtGeo g, float *__restrict__ hm, float *__restrict__ anno_box, int64_t *__restrict__ ind,
                                                           uint8_t *__restrict__ mask, int64_t *__restrict__ cat, float *__restrict__ boxes_cls) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= frames * g.max_objs) return;
    const int b = i / g.max_objs, k = i - b * g.max_objs;
    float ab[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, bc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int64_t o_ind = 0, o_cat = 0;
    uint8_t o_mask = 0;
    if (k < max_boxes) {
        const float *bx = boxes + ((int64_t)b * max_boxes + k) * 9;
        const int cls = classes[(int64_t)b * max_boxes + k];
        if (cls > 0 && cls <= g.num_classes) {
            const float two_pi = (float)(3.141592653589793 * 2);
            const float yaw = bx[8] - floorf(bx[8] / two_pi + 0.5f) * two_pi;
            bc[0] = bx[0]; bc[1] = bx[1]; bc[2] = bx[2]; bc[3] = bx[3]; bc[4] = bx[4]; bc[5] = bx[5];
            bc[6] = yaw; bc[7] = bx[6]; bc[8] = bx[7]; bc[9] = (float)cls;
            const float w = bx[3] / g.vx / (float)g.factor, l = bx[4] / g.vy / (float)g.factor;
            if (w > 0.f && l > 0.f) {
                int radius = (int)tgt_gaussian_radius((double)l, (double)w, g.overlap);
                radius = radius > g.min_radius ? radius : g.min_radius;
                const float cx = (bx[0] - g.x0) / g.vx / (float)g.factor, cy = (bx[1] - g.y0) / g.vy / (float)g.factor;
                const int xi = (int)cx, yi = (int)cy;   // truncation, as ndarray.astype(int32)
                if (xi >= 0 && xi < g.fw && yi >= 0 && yi < g.fh) {
                    const double sigma = (2 * radius + 1) / 6.0;
                    const int left = min(xi, radius), right = min(g.fw - xi, radius + 1);
                    const int top = min(yi, radius), bottom = min(g.fh - yi, radius + 1);
                    float *plane = hm + ((int64_t)b * g.num_classes + (cls - 1)) * g.fh * g.fw;
                    for (int dy = -top; dy < bottom; ++dy)
                        for (int dx = -left; dx < right; ++dx) {
                            const double gv = exp(-(double)(dx * dx + dy * dy) / (2 * sigma * sigma));
                            const float gf = gv < 2.220446049250313e-16 ? 0.f : (float)gv;
                            // non-negative floats order like their bit patterns
                            atomicMax(reinterpret_cast<int *>(plane + (int64_t)(yi + dy) * g.fw + xi + dx), __float_as_int(gf));
                        }
                    o_cat = cls - 1;
                    o_ind = (int64_t)yi * g.fw + xi;
                    o_mask = 1;
                    ab[0] = cx - (float)xi; ab[1] = cy - (float)yi; ab[2] = bx[2];
                    ab[3] = logf(bx[3]); ab[4] = logf(bx[4]); ab[5] = logf(bx[5]);
                    ab[6] = bx[6]; ab[7] = bx[7]; ab[8] = sinf(yaw); ab[9] = cosf(yaw);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 10; ++e) {
        anno_box[(int64_t)i * 10 + e] = ab[e];
        if (boxes_cls) boxes_cls[(int64_t)i * 10 + e] = bc[e];
    }
    ind[i] = o_ind;
    mask[i] = o_mask;
    cat[i] = o_cat;
}

}  // namespace s2d

using namespace s2d;

extern "C" int s2d_assign_label(const float *gt_boxes, const int32_t *gt_classes, int frames, int max_boxes, const float pc_range_xy[2],
                                const float voxel_size_xy[2], int out_size_factor, int fmap_w, int fmap_h, int num_classes, int max_objs,
                                double gaussian_overlap, int min_radius, float *hm_zeroed, float *anno_box, int64_t *ind, uint8_t *mask,
                                int64_t *cat, float *gt_boxes_and_cls, s2d_stream_t stream) {
    S2D_CHECK_ARG(frames > 0 && max_boxes >= 0 && max_objs > 0 && fmap_w > 0 && fmap_h > 0 && num_classes > 0 && out_size_factor > 0,
                  "assign_label: bad sizes");
    S2D_CHECK_ARG(pc_range_xy && voxel_size_xy && hm_zeroed && anno_box && ind && mask && cat && (max_boxes == 0 || (gt_boxes && gt_classes)),
                  "assign_label: null argument");
    TgtGeo g{pc_range_xy[0], pc_range_xy[1], voxel_size_xy[0], voxel_size_xy[1], out_size_factor, fmap_w, fmap_h, num_classes, max_objs,
             min_radius, gaussian_overlap};
    const int total = frames * max_objs;
    hipLaunchKernelGGL(assign_label_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, gt_boxes, gt_classes, frames, max_boxes,
                       g, hm_zeroed, anno_box, ind, mask, cat, gt_boxes_and_cls);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
