/* ORACLE (test infrastructure only) - CPU restatement of the rotated BEV IoU + greedy NMS used by CenterHead.predict:
 *   /root/reference/det3d/ops/iou3d_nms/src/iou3d_nms_kernel.cu:30-234 (box_overlap / iou_bev), :267-326 (nms_kernel),
 *   /root/reference/det3d/ops/iou3d_nms/src/iou3d_nms.cpp:92-130 (host walk of the suppression mask),
 *   /root/reference/det3d/core/bbox/box_torch_ops.py:449-464 (rotate_nms_pcdet: sort by score, pre/post max sizes).
 * PARITY UNPINNED BY REFERENCE-RUN VECTORS: the reference implementation is CUDA (its CPU twin iou3d_cpu.cpp includes <cuda.h>),
 * neither builds in this image.  The restatement is cross-checked against an independent float64 polygon-clipping IoU
 * (tests/test_nms.py).  Plain C, fp32 arithmetic in the reference's operation order. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y; } pt;

static float crs(pt a, pt b, pt o) { return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y); }

static void corners(const float *bx, pt *c) {
    const float hx = bx[3] / 2, hy = bx[4] / 2, ca = cosf(bx[6]), sa = sinf(bx[6]);
    const float lx[4] = {-hx, hx, hx, -hx}, ly[4] = {-hy, -hy, hy, hy};
    for (int k = 0; k < 4; ++k) {
        const float px = bx[0] + lx[k], py = bx[1] + ly[k];
        c[k].x = (px - bx[0]) * ca + (py - bx[1]) * (-sa) + bx[0];
        c[k].y = (px - bx[0]) * sa + (py - bx[1]) * ca + bx[1];
    }
    c[4] = c[0];
}

static int inside(const float *bx, pt p) {
    const float ca = cosf(-bx[6]), sa = sinf(-bx[6]);
    const float rx = (p.x - bx[0]) * ca + (p.y - bx[1]) * (-sa), ry = (p.x - bx[0]) * sa + (p.y - bx[1]) * ca;
    return fabsf(rx) < bx[3] / 2 + 1e-2f && fabsf(ry) < bx[4] / 2 + 1e-2f;
}

static int isect(pt p1, pt p0, pt q1, pt q0, pt *out) {
    if (!(fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) && fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) &&
          fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y)))
        return 0;
    const float s1 = crs(q0, p1, p0), s2 = crs(p1, q1, p0), s3 = crs(p0, q1, q0), s4 = crs(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    const float s5 = crs(q1, p1, p0);
    if (fabsf(s5 - s1) > 1e-8f) {
        out->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        out->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y, d = a0 * b1 - a1 * b0;
        out->x = (b0 * c1 - b1 * c0) / d;
        out->y = (a1 * c0 - a0 * c1) / d;
    }
    return 1;
}

float s2d_oracle_bev_iou(const float *a, const float *b) {
    pt ca[5], cb[5], p[16];
    corners(a, ca);
    corners(b, cb);
    int n = 0;
    float sx = 0, sy = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (isect(ca[i + 1], ca[i], cb[j + 1], cb[j], &p[n])) { sx += p[n].x; sy += p[n].y; ++n; }
    for (int k = 0; k < 4; ++k) {
        if (inside(a, cb[k])) { sx += cb[k].x; sy += cb[k].y; p[n++] = cb[k]; }
        if (inside(b, ca[k])) { sx += ca[k].x; sy += ca[k].y; p[n++] = ca[k]; }
    }
    const float cx = sx / n, cy = sy / n;
    for (int j = 0; j < n - 1; ++j)
        for (int i = 0; i < n - j - 1; ++i)
            if (atan2f(p[i].y - cy, p[i].x - cx) > atan2f(p[i + 1].y - cy, p[i + 1].x - cx)) { pt t = p[i]; p[i] = p[i + 1]; p[i + 1] = t; }
    float area = 0;
    for (int k = 0; k < n - 1; ++k) area += (p[k].x - p[0].x) * (p[k + 1].y - p[0].y) - (p[k].y - p[0].y) * (p[k + 1].x - p[0].x);
    const float ov = fabsf(area) / 2.0f, sa = a[3] * a[4], sb = b[3] * b[4];
    return ov / fmaxf(sa + sb - ov, 1e-8f);
}

void s2d_oracle_bev_iou_matrix(const float *a, int na, const float *b, int nb, float *out) {
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) out[(int64_t)i * nb + j] = s2d_oracle_bev_iou(a + i * 7, b + j * 7);
}

/* boxes already sorted by descending score; returns the number kept (<= max_keep), indices into the sorted list */
int s2d_oracle_nms(const float *boxes, int n, float thresh, int max_keep, int64_t *keep) {
    unsigned char *dead = (unsigned char *)calloc(n > 0 ? n : 1, 1);
    int cnt = 0;
    for (int i = 0; i < n && cnt < max_keep; ++i) {
        if (dead[i]) continue;
        keep[cnt++] = i;
        for (int j = i + 1; j < n; ++j)
            if (!dead[j] && s2d_oracle_bev_iou(boxes + i * 7, boxes + j * 7) > thresh) dead[j] = 1;
    }
    free(dead);
    return cnt;
}

/* CenterPoint's circle NMS (/root/reference/det3d/core/utils/circle_nms_jit.py:4-31): centres already sorted by descending score; a later
 * centre j is suppressed by a kept centre i when (xi - xj)^2 + (yi - yj)^2 <= thresh (squared distance against the threshold as is).
 * Returns the number kept (the caller truncates to post_max_size, center_head.py:503). */
int s2d_oracle_circle_nms(const float *xy, int n, float thresh, int64_t *keep) {
    unsigned char *dead = (unsigned char *)calloc(n > 0 ? n : 1, 1);
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        if (dead[i]) continue;
        keep[cnt++] = i;
        for (int j = i + 1; j < n; ++j) {
            if (dead[j]) continue;
            const float dx = xy[2 * i] - xy[2 * j], dy = xy[2 * i + 1] - xy[2 * j + 1];
            const float d0 = dx * dx, d1 = dy * dy;   /* two roundings, then the sum: no fused multiply-add */
            if (d0 + d1 <= thresh) dead[j] = 1;
        }
    }
    free(dead);
    return cnt;
}
