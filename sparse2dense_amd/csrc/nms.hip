// Rotated bird's-eye-view IoU and greedy NMS of CenterHead.predict:
//   box_torch_ops.rotate_nms_pcdet            /root/reference/det3d/core/bbox/box_torch_ops.py:449-464
//   iou3d_nms_cuda.nms_gpu / nms_kernel       /root/reference/det3d/ops/iou3d_nms/src/iou3d_nms_kernel.cu:236-326, src/iou3d_nms.cpp:92-130
// Boxes are 7 floats (x, y, z, dx, dy, dz, heading), already sorted by descending score.  Same geometry as the reference: the
// overlap polygon of two rotated rectangles is collected from (i) the proper intersections of the 4 x 4 edge pairs and (ii) the
// corners of either box lying inside the other (with its 1e-2 margin), ordered by angle around their centroid, and measured by
// the shoelace sum; IoU = overlap / max(area_a + area_b - overlap, 1e-8); box j is suppressed by an earlier kept box i when
// IoU(i, j) > threshold.
// Design: one kernel fills the upper-triangular suppression bit matrix (64 x 64 tiles, column boxes staged in LDS), a second,
// single-workgroup kernel walks it on the DEVICE (the reference copies the matrix to the host and walks it there) and writes
// the kept indices + their count: no host round trip inside predict().
#include "s2d_common.h"

namespace s2d {

struct P2 {
    float x, y;
};

__host__ __device__ inline float cross2(const P2 &a, const P2 &b, const P2 &o) { return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y); }

__host__ __device__ inline void rect_corners(const float *bx, P2 (&c)[5]) {
    const float hx = bx[3] / 2, hy = bx[4] / 2;
    const float ca = cosf(bx[6]), sa = sinf(bx[6]);
    const float lx[4] = {-hx, hx, hx, -hx}, ly[4] = {-hy, -hy, hy, hy};
    for (int k = 0; k < 4; ++k) {
        // rotate the axis-aligned corner about the centre
        const float px = bx[0] + lx[k], py = bx[1] + ly[k];
        c[k].x = (px - bx[0]) * ca + (py - bx[1]) * (-sa) + bx[0];
        c[k].y = (px - bx[0]) * sa + (py - bx[1]) * ca + bx[1];
    }
    c[4] = c[0];
}

__host__ __device__ inline bool inside_rect(const float *bx, const P2 &p) {
    const float margin = 1e-2f;
    const float ca = cosf(-bx[6]), sa = sinf(-bx[6]);
    const float rx = (p.x - bx[0]) * ca + (p.y - bx[1]) * (-sa);
    const float ry = (p.x - bx[0]) * sa + (p.y - bx[1]) * ca;
    return fabsf(rx) < bx[3] / 2 + margin && fabsf(ry) < bx[4] / 2 + margin;
}

// proper intersection of segment p0-p1 with q0-q1 (bounding-box reject, strict opposite-side test)
__host__ __device__ inline bool seg_intersect(const P2 &p1, const P2 &p0, const P2 &q1, const P2 &q0, P2 &out) {
    const bool boxes_touch = fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
                             fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y);
    if (!boxes_touch) return false;
    const float s1 = cross2(q0, p1, p0), s2 = cross2(p1, q1, p0), s3 = cross2(p0, q1, q0), s4 = cross2(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
    const float s5 = cross2(q1, p1, p0);
    if (fabsf(s5 - s1) > 1e-8f) {
        out.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        out.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {   // nearly parallel: solve the two line equations
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float d = a0 * b1 - a1 * b0;
        out.x = (b0 * c1 - b1 * c0) / d;
        out.y = (a1 * c0 - a0 * c1) / d;
    }
    return true;
}

__host__ __device__ inline float bev_overlap(const float *a, const float *b) {
    P2 ca[5], cb[5], pts[16];
    rect_corners(a, ca);
    rect_corners(b, cb);
    int cnt = 0;
    float sx = 0.f, sy = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (seg_intersect(ca[i + 1], ca[i], cb[j + 1], cb[j], pts[cnt])) {
                sx += pts[cnt].x; sy += pts[cnt].y;
                ++cnt;
            }
    for (int k = 0; k < 4; ++k) {
        if (inside_rect(a, cb[k])) { sx += cb[k].x; sy += cb[k].y; pts[cnt++] = cb[k]; }
        if (inside_rect(b, ca[k])) { sx += ca[k].x; sy += ca[k].y; pts[cnt++] = ca[k]; }
    }
    const P2 ctr{sx / cnt, sy / cnt};
    // order by angle around the centroid (exchange sort, as the reference: descending-angle pairs are swapped)
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i)
            if (atan2f(pts[i].y - ctr.y, pts[i].x - ctr.x) > atan2f(pts[i + 1].y - ctr.y, pts[i + 1].x - ctr.x)) {
                const P2 tmp = pts[i];
                pts[i] = pts[i + 1];
                pts[i + 1] = tmp;
            }
    float area = 0.f;
    for (int k = 0; k < cnt - 1; ++k)
        area += (pts[k].x - pts[0].x) * (pts[k + 1].y - pts[0].y) - (pts[k].y - pts[0].y) * (pts[k + 1].x - pts[0].x);
    return fabsf(area) / 2.0f;
}

__host__ __device__ inline float bev_iou(const float *a, const float *b) {
    const float sa = a[3] * a[4], sb = b[3] * b[4];
    const float ov = bev_overlap(a, b);
    return ov / fmaxf(sa + sb - ov, 1e-8f);
}

__global__ __launch_bounds__(256) void bev_iou_matrix_kernel(const float *__restrict__ a, int na, const float *__restrict__ b, int nb, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)na * nb) return;
    out[i] = bev_iou(a + (i / nb) * 7, b + (i % nb) * 7);
}

// suppression bits: mask[i][cb] bit j = IoU(box i, box cb*64+j) > thresh, for j > i only (earlier boxes cannot be suppressed by later)
__global__ __launch_bounds__(64) void nms_mask_kernel(const float *__restrict__ boxes, int n, float thresh, unsigned long long *__restrict__ mask) {
    __shared__ float col[64 * 7];
    const int rb = blockIdx.y, cb = blockIdx.x;
    const int col_blocks = (n + 63) / 64;
    const int t = threadIdx.x;
    if (cb < rb) {   // strictly lower tiles carry no bits
        if (rb * 64 + t < n) mask[(int64_t)(rb * 64 + t) * col_blocks + cb] = 0ull;
        return;
    }
    const int cj = cb * 64 + t;
    if (cj < n)
        for (int e = 0; e < 7; ++e) col[t * 7 + e] = boxes[(int64_t)cj * 7 + e];
    __syncthreads();
    const int ri = rb * 64 + t;
    if (ri >= n) return;
    const float *mine = boxes + (int64_t)ri * 7;
    const int ncol = min(64, n - cb * 64);
    unsigned long long bits = 0ull;
    for (int j = (rb == cb ? t + 1 : 0); j < ncol; ++j)
        if (bev_iou(mine, col + j * 7) > thresh) bits |= 1ull << j;
    mask[(int64_t)ri * col_blocks + cb] = bits;
}

// centre-distance suppression bits (CenterPoint's circle NMS, det3d/core/utils/circle_nms_jit.py:4-31): bit j = (xi - xj)^2 + (yi - yj)^2 <= thresh
// (the reference compares the SQUARED distance with `min_radius` as is), j > i only
__global__ __launch_bounds__(64) void circle_mask_kernel(const float *__restrict__ xy, int n, float thresh, unsigned long long *__restrict__ mask) {
    __shared__ float col[64 * 2];
    const int rb = blockIdx.y, cb = blockIdx.x;
    const int col_blocks = (n + 63) / 64;
    const int t = threadIdx.x;
    if (cb < rb) {
        if (rb * 64 + t < n) mask[(int64_t)(rb * 64 + t) * col_blocks + cb] = 0ull;
        return;
    }
    const int cj = cb * 64 + t;
    if (cj < n) {
        col[t * 2] = xy[(int64_t)cj * 2];
        col[t * 2 + 1] = xy[(int64_t)cj * 2 + 1];
    }
    __syncthreads();
    const int ri = rb * 64 + t;
    if (ri >= n) return;
    const float x = xy[(int64_t)ri * 2], y = xy[(int64_t)ri * 2 + 1];
    const int ncol = min(64, n - cb * 64);
    unsigned long long bits = 0ull;
    for (int j = (rb == cb ? t + 1 : 0); j < ncol; ++j) {
        const float dx = x - col[j * 2], dy = y - col[j * 2 + 1];
        // the reference's expression, evaluated without FMA contraction (the library is built with -ffp-contract=off): (dx)**2 + (dy)**2
        if (dx * dx + dy * dy <= thresh) bits |= 1ull << j;
    }
    mask[(int64_t)ri * col_blocks + cb] = bits;
}

// greedy walk over the bit matrix in one workgroup: `removed` lives in LDS; box i is kept iff its bit is clear when reached
__global__ __launch_bounds__(256) void nms_select_kernel(const unsigned long long *__restrict__ mask, int n, int max_keep, int64_t *__restrict__ keep,
                                                         int32_t *__restrict__ n_keep) {
    extern __shared__ unsigned long long removed[];
    const int col_blocks = (n + 63) / 64;
    for (int e = threadIdx.x; e < col_blocks; e += 256) removed[e] = 0ull;
    __shared__ int count;
    __shared__ int alive;
    if (threadIdx.x == 0) count = 0;
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        if (threadIdx.x == 0) alive = !((removed[i >> 6] >> (i & 63)) & 1ull);
        __syncthreads();
        if (alive) {
            if (threadIdx.x == 0) {
                if (count < max_keep) keep[count] = i;
                ++count;
            }
            for (int e = (i >> 6) + threadIdx.x; e < col_blocks; e += 256) removed[e] |= mask[(int64_t)i * col_blocks + e];
        }
        __syncthreads();
        if (count >= max_keep) break;   // uniform: count is shared and settled by the barrier
    }
    if (threadIdx.x == 0) n_keep[0] = count < max_keep ? count : max_keep;
}

}  // namespace s2d

using namespace s2d;

extern "C" int s2d_bev_iou_f32(const float *boxes_a, int na, const float *boxes_b, int nb, float *iou, s2d_stream_t stream) {
    S2D_CHECK_ARG(na >= 0 && nb >= 0, "bev_iou: bad sizes");
    if (na == 0 || nb == 0) return S2D_OK;
    S2D_CHECK_ARG(boxes_a && boxes_b && iou, "bev_iou: null argument");
    hipLaunchKernelGGL(bev_iou_matrix_kernel, dim3((unsigned)ceil_div((int64_t)na * nb, 256)), dim3(256), 0, (hipStream_t)stream, boxes_a, na, boxes_b,
                       nb, iou);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

extern "C" size_t s2d_nms_workspace_bytes(int n) { return n <= 0 ? 256 : align_up((size_t)n * ((n + 63) / 64) * sizeof(unsigned long long), 256); }

extern "C" int s2d_nms_rotated_bev(const float *boxes_sorted, int n, float iou_threshold, int max_keep, int64_t *keep, int32_t *n_keep, void *ws,
                                   size_t ws_bytes, s2d_stream_t stream) {
    S2D_CHECK_ARG(n >= 0 && n <= 65536 && max_keep >= 0 && keep && n_keep, "nms: bad argument (n <= 65536)");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0 || max_keep == 0) {
        S2D_HIP(hipMemsetAsync(n_keep, 0, sizeof(int32_t), st));
        return S2D_OK;
    }
    S2D_CHECK_ARG(boxes_sorted, "nms: null boxes");
    if (!ws || ws_bytes < s2d_nms_workspace_bytes(n)) {
        set_error("nms: workspace too small");
        return S2D_ERR_WORKSPACE;
    }
    const int cb = (n + 63) / 64;
    unsigned long long *mask = (unsigned long long *)ws;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(cb, cb), dim3(64), 0, st, boxes_sorted, n, iou_threshold, mask);
    hipLaunchKernelGGL(nms_select_kernel, dim3(1), dim3(256), (size_t)cb * sizeof(unsigned long long), st, mask, n, max_keep, keep, n_keep);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}

/* CenterPoint's circle NMS (see include/s2d.h) */
extern "C" int s2d_nms_circle(const float *xy_sorted, int n, float thresh, int max_keep, int64_t *keep, int32_t *n_keep, void *ws, size_t ws_bytes,
                              s2d_stream_t stream) {
    S2D_CHECK_ARG(n >= 0 && n <= 65536 && max_keep >= 0 && keep && n_keep, "nms_circle: bad argument (n <= 65536)");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0 || max_keep == 0) {
        S2D_HIP(hipMemsetAsync(n_keep, 0, sizeof(int32_t), st));
        return S2D_OK;
    }
    S2D_CHECK_ARG(xy_sorted, "nms_circle: null centres");
    if (!ws || ws_bytes < s2d_nms_workspace_bytes(n)) {
        set_error("nms_circle: workspace too small");
        return S2D_ERR_WORKSPACE;
    }
    const int cb = (n + 63) / 64;
    unsigned long long *mask = (unsigned long long *)ws;
    hipLaunchKernelGGL(circle_mask_kernel, dim3(cb, cb), dim3(64), 0, st, xy_sorted, n, thresh, mask);
    hipLaunchKernelGGL(nms_select_kernel, dim3(1), dim3(256), (size_t)cb * sizeof(unsigned long long), st, mask, n, max_keep, keep, n_keep);
    S2D_LAUNCH_CHECK();
    return S2D_OK;
}
