"""Synthetic Waymo-shaped LiDAR scenes (scenes A/B/C) and CenterPoint targets.

The reference trains on real Waymo frames (`det3d/datasets/waymo/waymo.py:66-96`); there is no
dataset on the GPU box, so the bench and the tests use a seeded ring-scan generator whose point
layout follows the reference loader (`det3d/datasets/pipelines/loading.py:61-70,140-145`:
`points f32[N,5] = x, y, z, tanh(intensity), elongation`) and whose training targets follow
`AssignLabel` (`det3d/datasets/pipelines/preprocess.py:553-624`) with the gaussian helpers of
`det3d/core/utils/center_utils.py:18-64`.
"""
import math

import numpy as np

WAYMO_RANGE = (-75.2, -75.2, -2.0, 75.2, 75.2, 4.0)
WAYMO_VOXEL = (0.1, 0.1, 0.15)
PILLAR_RANGE = (-74.88, -74.88, -2.0, 74.88, 74.88, 4.0)
PILLAR_VOXEL = (0.32, 0.32, 6.0)
WAYMO_BEAM_JITTER = 2.5e-3   # see make_scene(beam_jitter=...)


def _ray_box(origin, dirs, center, size, yaw):
    """Distance along each ray to a yawed box (slab test), inf when missed."""
    c, s = math.cos(yaw), math.sin(yaw)
    rot = np.array([[c, s, 0.0], [-s, c, 0.0], [0.0, 0.0, 1.0]])
    o = rot @ (origin - center)
    d = dirs @ rot.T
    half = np.asarray(size) * 0.5
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        t0 = (-half - o) * inv
        t1 = (half - o) * inv
    tmin = np.minimum(t0, t1).max(axis=1)
    tmax = np.maximum(t0, t1).min(axis=1)
    hit = (tmax >= np.maximum(tmin, 0.0)) & np.isfinite(tmin)
    return np.where(hit, np.maximum(tmin, 0.0), np.inf)


def _ground(x, y):
    return 0.5 * np.sin(x / 25.0) * np.cos(y / 31.0)


def make_scene(n_points=150000, seed=20240928, n_cars=100, n_walls=25, n_peds=40,
               pc_range=WAYMO_RANGE, return_objects=True, beam_jitter=2e-4):
    """One LiDAR sweep: 64 beams x 2650 azimuths ray-cast against an undulating ground, walls,
    car-sized boxes and pedestrian-sized posts; cropped to `pc_range`, resampled to exactly
    `n_points` rows and shuffled (the reference shuffles too, preprocess.py:257-260).

    Returns dict(points f32[n,5], gt_boxes f32[K,9] (x,y,z,w,l,h,vx,vy,yaw), gt_classes i32[K]
    (1=VEHICLE 2=PEDESTRIAN 3=CYCLIST), object_points f32[P,5] (points that hit objects)).
    """
    rs = np.random.RandomState(seed)
    origin = np.array([0.0, 0.0, 2.1])
    elev = np.deg2rad(-17.6 + 20.0 * np.linspace(0.0, 1.0, 64) ** 0.7)  # denser near the horizon
    azim = np.linspace(-math.pi, math.pi, 2650, endpoint=False)
    ee, aa = np.meshgrid(elev, azim, indexing="ij")
    # beam_jitter (rad): per-return pointing noise.  2e-4 keeps the 64 rings razor thin (M ~ 65-70 k voxels per 150 k
    # points: the sparse end of Waymo); WAYMO_BEAM_JITTER spreads them over neighbouring cells the way ego-motion
    # compensation and the four short-range lidars do in real sweeps (M ~ 100 k, SURVEY 8(d)'s 90-130 k bracket).
    ee = ee.ravel() + rs.normal(0, beam_jitter, ee.size)
    aa = aa.ravel() + rs.normal(0, beam_jitter, aa.size)
    dirs = np.stack([np.cos(ee) * np.cos(aa), np.cos(ee) * np.sin(aa), np.sin(ee)], axis=1)

    # ground: flat-plane hit refined twice against the height field
    with np.errstate(divide="ignore", invalid="ignore"):
        t_g = np.where(dirs[:, 2] < -1e-3, -origin[2] / dirs[:, 2], np.inf)
    for _ in range(2):
        gx = origin[0] + t_g * dirs[:, 0]
        gy = origin[1] + t_g * dirs[:, 1]
        h = _ground(np.nan_to_num(gx, posinf=0.0), np.nan_to_num(gy, posinf=0.0))
        with np.errstate(divide="ignore", invalid="ignore"):
            t_g = np.where(dirs[:, 2] < -1e-3, (h - origin[2]) / dirs[:, 2], np.inf)
    t_best = t_g.copy()
    is_obj = np.zeros(t_best.shape, bool)

    boxes, classes = [], []

    def place(kind_count, size_fn, cls, rmin, rmax, is_gt=True):
        for _ in range(kind_count):
            r = rs.uniform(rmin, rmax)
            th = rs.uniform(-math.pi, math.pi)
            x, y = r * math.cos(th), r * math.sin(th)
            w, l, hgt = size_fn()
            yaw = rs.uniform(-math.pi, math.pi)
            z = float(_ground(x, y)) + hgt / 2
            t = _ray_box(origin, dirs, np.array([x, y, z]), (l, w, hgt), yaw)
            closer = t < t_best
            t_best[closer] = t[closer]
            is_obj[closer] = is_gt
            if is_gt:
                boxes.append([x, y, z, w, l, hgt, rs.normal(0, 2.0), rs.normal(0, 2.0), yaw])
                classes.append(cls)

    place(n_walls, lambda: (0.4, rs.uniform(5, 30), 6.0), 0, 25.0, 72.0, is_gt=False)
    place(n_cars, lambda: (rs.uniform(1.8, 2.2), rs.uniform(4.2, 5.2), rs.uniform(1.5, 1.9)), 1, 8.0, 70.0)
    place(n_peds, lambda: (rs.uniform(0.5, 0.9), rs.uniform(0.5, 0.9), rs.uniform(1.5, 1.9)), 2, 4.0, 50.0)
    place(max(n_peds // 4, 1), lambda: (rs.uniform(0.6, 0.9), rs.uniform(1.5, 2.0), rs.uniform(1.4, 1.8)), 3, 4.0, 50.0)

    ok = np.isfinite(t_best) & (t_best < 120.0)
    t = t_best[ok] + rs.normal(0, 0.01, ok.sum())
    pts = origin[None, :] + t[:, None] * dirs[ok]
    obj = is_obj[ok]
    lo, hi = np.asarray(pc_range[:3]), np.asarray(pc_range[3:])
    inside = np.all((pts >= lo + 1e-3) & (pts < hi - 1e-3), axis=1)
    pts, obj = pts[inside], obj[inside]

    if pts.shape[0] > n_points:
        keep = rs.choice(pts.shape[0], n_points, replace=False)
        pts, obj = pts[keep], obj[keep]
    elif pts.shape[0] < n_points:
        extra = n_points - pts.shape[0]
        r = np.sqrt(rs.uniform(20.0 ** 2, 74.0 ** 2, extra))
        th = rs.uniform(-math.pi, math.pi, extra)
        ex, ey = r * np.cos(th), r * np.sin(th)
        ez = _ground(ex, ey) + rs.normal(0, 0.03, extra)
        epts = np.stack([ex, ey, ez], axis=1)
        epts = np.clip(epts, lo + 1e-3, hi - 1e-3)
        pts = np.concatenate([pts, epts], 0)
        obj = np.concatenate([obj, np.zeros(extra, bool)], 0)

    feats = np.stack([np.tanh(rs.uniform(0, 1.5, pts.shape[0])), rs.uniform(0, 1.0, pts.shape[0])], 1)
    points = np.concatenate([pts, feats], axis=1).astype(np.float32)
    perm = rs.permutation(points.shape[0])
    points, obj = points[perm], obj[perm]
    out = dict(points=np.ascontiguousarray(points))
    if return_objects:
        out["gt_boxes"] = np.asarray(boxes, np.float32).reshape(-1, 9)
        out["gt_classes"] = np.asarray(classes, np.int32)
        out["object_points"] = np.ascontiguousarray(points[obj])
    return out


def make_distill_points(scene, seed=1, n_extra=20000):
    """Teacher-side inputs of the distillation config (preprocess.py:59-272 builds them from the
    GT object database): `dense_points` = sweep + extra object-surface points,
    `reconstruction_points` = object points only."""
    rs = np.random.RandomState(seed)
    boxes = scene["gt_boxes"]
    k = rs.randint(0, boxes.shape[0], n_extra)
    b = boxes[k]
    u = rs.uniform(-0.5, 0.5, (n_extra, 3))
    face = rs.randint(0, 3, n_extra)
    u[np.arange(n_extra), face] = np.sign(u[np.arange(n_extra), face]) * 0.5
    local = u * b[:, [4, 3, 5]]
    c, s = np.cos(b[:, 8]), np.sin(b[:, 8])
    x = b[:, 0] + c * local[:, 0] - s * local[:, 1]
    y = b[:, 1] + s * local[:, 0] + c * local[:, 1]
    z = b[:, 2] + local[:, 2]
    extra = np.stack([x, y, z, np.tanh(rs.uniform(0, 1.5, n_extra)), rs.uniform(0, 1, n_extra)], 1)
    extra = extra.astype(np.float32)
    lo, hi = np.asarray(WAYMO_RANGE[:3], np.float32), np.asarray(WAYMO_RANGE[3:], np.float32)
    extra = extra[np.all((extra[:, :3] >= lo + 1e-3) & (extra[:, :3] < hi - 1e-3), axis=1)]
    dense = np.concatenate([scene["points"], extra], 0)
    dense = dense[rs.permutation(dense.shape[0])]
    recon = np.concatenate([scene["object_points"], extra], 0)
    recon = recon[rs.permutation(recon.shape[0])]
    return np.ascontiguousarray(dense), np.ascontiguousarray(recon)


# ----------------------------------------------------------------------------------------------
# CenterPoint targets (AssignLabel, preprocess.py:553-624; center_utils.py:18-64)
# ----------------------------------------------------------------------------------------------
def gaussian_radius(det_size, min_overlap=0.5):
    """center_utils.py:18-39 (three quadratic roots, take the smallest)."""
    height, width = det_size
    b1 = height + width
    c1 = width * height * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 + np.sqrt(b1 ** 2 - 4 * c1)) / 2
    b2 = 2 * (height + width)
    c2 = (1 - min_overlap) * width * height
    r2 = (b2 + np.sqrt(b2 ** 2 - 16 * c2)) / 2
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (height + width)
    c3 = (min_overlap - 1) * width * height
    r3 = (b3 + np.sqrt(b3 ** 2 - 4 * a3 * c3)) / 2
    return min(r1, r2, r3)


def draw_gaussian(heatmap, center, radius):
    """center_utils.py:41-64 (draw_umich_gaussian with k=1)."""
    diameter = 2 * radius + 1
    sigma = diameter / 6
    yy, xx = np.ogrid[-radius:radius + 1, -radius:radius + 1]
    g = np.exp(-(xx * xx + yy * yy) / (2 * sigma * sigma))
    g[g < np.finfo(g.dtype).eps * g.max()] = 0
    x, y = int(center[0]), int(center[1])
    h, w = heatmap.shape
    left, right = min(x, radius), min(w - x, radius + 1)
    top, bottom = min(y, radius), min(h - y, radius + 1)
    mh = heatmap[y - top:y + bottom, x - left:x + right]
    mg = g[radius - top:radius + bottom, radius - left:radius + right]
    if min(mg.shape) > 0 and min(mh.shape) > 0:
        np.maximum(mh, mg, out=mh)
    return heatmap


def assign_targets(gt_boxes, gt_classes, pc_range=WAYMO_RANGE, voxel_size=WAYMO_VOXEL,
                   out_size_factor=8, num_classes=3, max_objs=500, gaussian_overlap=0.1, min_radius=2,
                   grid_xy=(1504, 1504)):
    """One-task Waymo targets: hm f32[3,H,W], anno_box f32[500,10], ind i64[500], mask u8[500],
    cat i64[500] (preprocess.py:553-624)."""
    fw, fh = grid_xy[0] // out_size_factor, grid_xy[1] // out_size_factor
    hm = np.zeros((num_classes, fh, fw), np.float32)
    anno_box = np.zeros((max_objs, 10), np.float32)
    ind = np.zeros((max_objs,), np.int64)
    mask = np.zeros((max_objs,), np.uint8)
    cat = np.zeros((max_objs,), np.int64)
    n = min(gt_boxes.shape[0], max_objs)
    gt_boxes = np.array(gt_boxes, np.float32, copy=True)
    if n:   # rotation limited to [-pi, pi) first (preprocess.py:540-543, box_np_ops.limit_period), on the fp32 box array
        gt_boxes[:, -1] = gt_boxes[:, -1] - np.floor(gt_boxes[:, -1] / (np.pi * 2) + 0.5) * (np.pi * 2)
    voxel_size = np.asarray(voxel_size, np.float32)
    pc_range = np.asarray(pc_range, np.float32)
    for k in range(n):
        box = gt_boxes[k]
        cls_id = int(gt_classes[k]) - 1
        if cls_id < 0:
            continue
        w = box[3] / voxel_size[0] / out_size_factor
        l = box[4] / voxel_size[1] / out_size_factor
        if not (w > 0 and l > 0):
            continue
        # the three roots in float64 from the fp32 sizes (the reference's numpy promoted fp32-scalar x python-float to float64)
        radius = max(min_radius, int(gaussian_radius((float(l), float(w)), min_overlap=gaussian_overlap)))
        cx = (box[0] - pc_range[0]) / voxel_size[0] / out_size_factor
        cy = (box[1] - pc_range[1]) / voxel_size[1] / out_size_factor
        ct = np.array([cx, cy], np.float32)
        ct_int = ct.astype(np.int32)
        if not (0 <= ct_int[0] < fw and 0 <= ct_int[1] < fh):
            continue
        draw_gaussian(hm[cls_id], ct, radius)
        x, y = int(ct_int[0]), int(ct_int[1])
        cat[k] = cls_id
        ind[k] = y * fw + x
        mask[k] = 1
        yaw = box[8]
        anno_box[k] = np.concatenate((ct - (x, y), box[2], np.log(box[3:6]), box[6], box[7],
                                      np.sin(yaw), np.cos(yaw)), axis=None)
    return dict(hm=hm, anno_box=anno_box, ind=ind, mask=mask, cat=cat)
