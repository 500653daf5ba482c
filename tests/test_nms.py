"""CenterHead.predict post-processing: rotated BEV IoU + greedy NMS.
 * oracle/iou_nms.c (the restatement of the reference's CUDA geometry) vs an INDEPENDENT float64 convex-polygon clipping IoU
   (Sutherland-Hodgman) and closed-form cases: the oracle is unpinned by reference-run vectors (the reference is CUDA only);
 * device kernels (csrc/nms.hip) vs the oracle: IoU within 1e-5, kept indices exact;
 * CenterHead.predict end to end vs a numpy restatement of center_head.py:293-495 built on the oracle NMS."""
import numpy as np
import pytest
import torch

from oracle import iou_nms as O


def _poly(b):
    x, y, dx, dy, a = b[0], b[1], b[3] / 2, b[4] / 2, b[6]
    c, s = np.cos(a), np.sin(a)
    pts = np.array([[-dx, -dy], [dx, -dy], [dx, dy], [-dx, dy]], np.float64)
    return pts @ np.array([[c, s], [-s, c]]) + [x, y]


def _clip_area(p, q):
    """area of the intersection of two convex CCW polygons, float64 Sutherland-Hodgman"""
    out = [tuple(v) for v in p]
    for i in range(len(q)):
        a, b = q[i], q[(i + 1) % len(q)]
        inp, out = out, []
        if not inp:
            break
        side = lambda v: (b[0] - a[0]) * (v[1] - a[1]) - (b[1] - a[1]) * (v[0] - a[0])
        for j in range(len(inp)):
            cur, nxt = inp[j], inp[(j + 1) % len(inp)]
            sc, sn = side(cur), side(nxt)
            if sc >= 0:
                out.append(cur)
            if (sc >= 0) != (sn >= 0):
                t = sc / (sc - sn)
                out.append((cur[0] + t * (nxt[0] - cur[0]), cur[1] + t * (nxt[1] - cur[1])))
    if len(out) < 3:
        return 0.0
    o = np.array(out)
    return 0.5 * abs(np.dot(o[:, 0], np.roll(o[:, 1], -1)) - np.dot(o[:, 1], np.roll(o[:, 0], -1)))


def _rand_boxes(n, seed, spread=6.0):
    rs = np.random.RandomState(seed)
    b = np.zeros((n, 7), np.float32)
    b[:, :2] = rs.uniform(-spread, spread, (n, 2))
    b[:, 3] = rs.uniform(1.5, 5.0, n); b[:, 4] = rs.uniform(0.8, 2.2, n); b[:, 5] = 1.6
    b[:, 6] = rs.uniform(-np.pi, np.pi, n)
    return b


def test_oracle_iou_matches_independent_polygon_clipping():
    a, b = _rand_boxes(60, 1), _rand_boxes(50, 2)
    got = O.bev_iou(a, b)
    for i in range(0, 60, 3):
        for j in range(0, 50, 3):
            inter = _clip_area(_poly(a[i]), _poly(b[j]))
            want = inter / max(a[i, 3] * a[i, 4] + b[j, 3] * b[j, 4] - inter, 1e-8)
            # the reference counts corners within 1e-2 of a box as inside: a few 1e-3 of slack on grazing contacts
            assert abs(got[i, j] - want) <= 5e-3 + 1e-4 * want, (i, j, got[i, j], want)
    # closed forms: identical boxes, disjoint boxes, unit squares rotated by 45 degrees (octagon area 2(sqrt 2 - 1))
    sq = np.array([[0, 0, 0, 1, 1, 1, 0], [0, 0, 0, 1, 1, 1, np.pi / 4], [5, 5, 0, 1, 1, 1, 0]], np.float32)
    m = O.bev_iou(sq, sq)
    assert abs(m[0, 0] - 1) < 1e-5 and m[0, 2] == 0
    oct_ = 2 * (np.sqrt(2) - 1)
    assert abs(m[0, 1] - oct_ / (2 - oct_)) < 1e-4


def test_oracle_nms_is_greedy_on_the_iou_matrix():
    b = _rand_boxes(300, 3, spread=10.0)
    s = np.random.RandomState(4).rand(300).astype(np.float32)
    keep = O.rotate_nms(b, s, 0.3, pre_maxsize=256, post_max_size=64)
    order = np.argsort(-s, kind="stable")[:256]
    iou = O.bev_iou(b[order], b[order])
    alive, want = np.ones(len(order), bool), []
    for i in range(len(order)):
        if alive[i]:
            want.append(order[i])
            alive[i + 1:] &= ~(iou[i, i + 1:] > 0.3)
    assert list(keep) == want[:64]


@pytest.mark.gpu
def test_device_iou_and_nms_match_oracle():
    from sparse2dense_amd import nms
    a, b = _rand_boxes(130, 5), _rand_boxes(70, 6)
    got = nms.boxes_iou_bev(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    np.testing.assert_allclose(got, O.bev_iou(a, b), rtol=0, atol=1e-5)
    for n, seed, thr in [(1000, 7, 0.7), (4096, 8, 0.25), (65, 9, 0.1), (1, 10, 0.5)]:
        bx = _rand_boxes(n, seed, spread=25.0)
        sc = np.random.RandomState(seed + 100).rand(n).astype(np.float32)
        sel = nms.rotate_nms(torch.from_numpy(bx).cuda(), torch.from_numpy(sc).cuda(), thr, pre_maxsize=4096, post_max_size=500)
        assert sel.cpu().numpy().tolist() == O.rotate_nms(bx, sc, thr, 4096, 500).tolist(), n
    empty = nms.rotate_nms(torch.zeros(0, 7).cuda(), torch.zeros(0).cuda(), 0.5)
    assert empty.numel() == 0


@pytest.mark.gpu
def test_center_head_predict_matches_restatement():
    from sparse2dense_amd import waymo_configs
    from sparse2dense_amd.registry import build_head
    cfg = waymo_configs.centerpoint_voxelnet()
    head = build_head(cfg["bbox_head"]).cuda().eval()
    g = torch.Generator().manual_seed(0)
    h = w = 188
    preds = [dict(reg=torch.rand(2, 2, h, w, generator=g), height=torch.randn(2, 1, h, w, generator=g), dim=torch.randn(2, 3, h, w, generator=g) * 0.3 + 0.8,
                  rot=torch.randn(2, 2, h, w, generator=g), hm=torch.randn(2, 3, h, w, generator=g) * 1.5 - 3.0)]
    test_cfg = dict(post_center_limit_range=[-80, -80, -10.0, 80, 80, 10.0], nms=dict(nms_pre_max_size=4096, nms_post_max_size=500, nms_iou_threshold=0.7),
                    score_threshold=0.1, pc_range=[-75.2, -75.2], out_size_factor=8, voxel_size=[0.1, 0.1])
    out = head.predict({}, [{k: v.cuda() for k, v in preds[0].items()}], test_cfg)
    assert len(out) == 2
    for i in range(2):
        p = {k: v[i].permute(1, 2, 0).reshape(h * w, -1).numpy().astype(np.float32) for k, v in preds[0].items()}
        hm = 1 / (1 + np.exp(-p["hm"].astype(np.float64)))
        ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        x = (xs.reshape(-1) + p["reg"][:, 0]) * 8 * 0.1 - 75.2
        y = (ys.reshape(-1) + p["reg"][:, 1]) * 8 * 0.1 - 75.2
        boxes = np.stack([x, y, p["height"][:, 0], *np.exp(p["dim"]).T, np.arctan2(p["rot"][:, 0], p["rot"][:, 1])], 1).astype(np.float32)
        scores, labels = hm.max(1).astype(np.float32), hm.argmax(1)
        m = (scores > 0.1) & np.all(boxes[:, :3] >= [-80, -80, -10], 1) & np.all(boxes[:, :3] <= [80, 80, 10], 1)
        sel = O.rotate_nms(boxes[m], scores[m], 0.7, 4096, 500)
        got = out[i]
        assert got["box3d_lidar"].shape == (len(sel), 7)
        # the device scores differ from the float64 sigmoid in the last ulp: compare as sets of boxes, ordered by score
        np.testing.assert_allclose(got["scores"].cpu().numpy(), scores[m][sel], rtol=1e-5)
        np.testing.assert_allclose(got["box3d_lidar"].cpu().numpy(), boxes[m][sel], rtol=1e-4, atol=1e-4)
        assert np.array_equal(got["label_preds"].cpu().numpy(), labels[m][sel])


# ---- r06: CenterPoint's circle NMS and the double-flip decode of CenterHead.predict (center_head.py:301-381,476-479,499-507) --------------
def _circle_brute(xy, scores, thresh, post_max):
    """the reference's loop (circle_nms_jit.py:4-31) written out again in plain Python, float32 arithmetic"""
    order = np.argsort(-scores, kind="stable")
    dead, keep = np.zeros(len(order), bool), []
    for a in range(len(order)):
        if dead[a]:
            continue
        i = order[a]
        keep.append(i)
        for b in range(a + 1, len(order)):
            j = order[b]
            d = np.float32(np.float32(xy[i, 0] - xy[j, 0]) ** 2) + np.float32(np.float32(xy[i, 1] - xy[j, 1]) ** 2)
            if d <= np.float32(thresh):
                dead[b] = True
    return keep[:post_max]


def test_oracle_circle_nms_is_the_greedy_centre_distance_loop():
    rs = np.random.RandomState(12)
    for n, thr, post in [(400, 4.0, 83), (50, 0.5, 500), (1, 1.0, 83), (0, 1.0, 83)]:
        xy = (rs.rand(n, 2) * 30).astype(np.float32)
        sc = rs.rand(n).astype(np.float32)
        assert list(O.circle_nms(xy, sc, thr, post)) == _circle_brute(xy, sc, thr, post), n


@pytest.mark.gpu
def test_device_circle_nms_matches_oracle():
    from sparse2dense_amd import nms
    rs = np.random.RandomState(13)
    for n, thr, post in [(3000, 2.0, 83), (4097, 0.25, 500), (65, 9.0, 83), (1, 1.0, 5)]:
        xy = (rs.rand(n, 2) * 60 - 30).astype(np.float32)
        sc = rs.rand(n).astype(np.float32)
        sel = nms.circle_nms(torch.from_numpy(xy).cuda(), torch.from_numpy(sc).cuda(), thr, post)
        assert sel.cpu().numpy().tolist() == O.circle_nms(xy, sc, thr, post).tolist(), n
    assert nms.circle_nms(torch.zeros(0, 2).cuda(), torch.zeros(0).cuda(), 1.0).numel() == 0


@pytest.mark.gpu
def test_center_head_predict_with_double_flip_and_circle_nms_matches_the_reference_golden(golden_dir):
    """the REFERENCE's CenterHead.predict(double_flip=True, circular_nms=True) on the same seeded maps (tests/golden/make_golden_r06.py)"""
    import os
    from golden_util import PREDICT_FLIP_CIRCLE_CFG as CFG, predict_flip_circle_inputs
    from sparse2dense_amd import waymo_configs
    from sparse2dense_amd.registry import build_head
    g = np.load(os.path.join(golden_dir, "predict_flip_circle.npz"))
    head = build_head(waymo_configs.centerpoint_voxelnet()["bbox_head"]).cuda().eval()
    preds = {k: v.cuda() for k, v in predict_flip_circle_inputs().items()}
    out = head.predict({}, [preds], CFG)
    assert len(out) == int(g["samples"]) == 2
    for i, r in enumerate(out):
        want_b, want_s, want_l = g[f"boxes{i}"], g[f"scores{i}"], g[f"labels{i}"]
        assert r["box3d_lidar"].shape == want_b.shape == (83, 7)
        np.testing.assert_allclose(r["scores"].cpu().numpy(), want_s, rtol=1e-5)
        np.testing.assert_allclose(r["box3d_lidar"].cpu().numpy(), want_b, rtol=1e-4, atol=2e-4)
        assert np.array_equal(r["label_preds"].cpu().numpy(), want_l)
    with pytest.raises(NotImplementedError):
        head.predict({}, [preds], dict(CFG, per_class_nms=True))
