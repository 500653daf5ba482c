"""GPU parity tests of the HIP kernels (through the C-ABI) against the CPU oracle and the golden
fixtures generated from the reference.  Run on the MI355X box:  pytest tests -m gpu

Bars: voxel indices / counts / rulebooks bit exact (rulebook pairs compared as sets, strided-conv
output rows in canonical sorted order); fp32 features within rtol 1e-4 / atol 1e-5 per layer
(different but fixed summation order: MFMA k-chains vs per-offset mm + index_add)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import spconv_ref as R
from oracle import voxelize as OV
from sparse2dense_amd import hip_ops as H
from sparse2dense_amd import scene

DEV = "cuda:0"


def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


# ------------------------------------------------------------------------------------------------
# voxelization
# ------------------------------------------------------------------------------------------------
GOLD = ["voxelize_small", "voxelize_maxvox", "voxelize_second8k", "voxelize_pillar", "voxelize_empty"]


def _check_vox(out, voxels, coors, num, mean=None, max_points=5):
    v, c, n, m = [x.cpu().numpy() if x is not None else None for x in out]
    assert np.array_equal(c, coors), "voxel coordinates / first-seen order differ"
    assert np.array_equal(n, num)
    assert np.array_equal(v.view(np.uint32), voxels.view(np.uint32))
    if mean is not None and mean.size:
        if max_points <= 5:
            ulp = np.abs(m.view(np.int32).astype(np.int64) - mean.view(np.int32).astype(np.int64))
            assert ulp.max() <= 2
        else:
            np.testing.assert_allclose(m, mean, rtol=3e-6, atol=1e-6)


@pytest.mark.parametrize("name", GOLD)
def test_voxelize_matches_reference_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    out = H.voxelize(_dev(g["points"]), g["voxel_size"], g["pc_range"], int(g["max_points"]), int(g["max_voxels"]))
    _check_vox(out, g["voxels"], g["coors"], g["num_points"], g["mean"], int(g["max_points"]))


def test_voxelize_150k_scene_matches_oracle_bit_exact():
    s = scene.make_scene(150000)
    pts = s["points"]
    voxels, coors, num = OV.points_to_voxel(pts, scene.WAYMO_VOXEL, scene.WAYMO_RANGE, 5, 150000)
    out = H.voxelize(_dev(pts), scene.WAYMO_VOXEL, scene.WAYMO_RANGE, 5, 150000)
    _check_vox(out, voxels, coors, num, OV.voxel_mean(voxels, num), 5)
    # the cap: same scene, max_voxels far below the distinct-voxel count
    voxels, coors, num = OV.points_to_voxel(pts, scene.WAYMO_VOXEL, scene.WAYMO_RANGE, 5, 20000)
    out = H.voxelize(_dev(pts), scene.WAYMO_VOXEL, scene.WAYMO_RANGE, 5, 20000)
    assert out[1].shape[0] == 20000
    _check_vox(out, voxels, coors, num, None, 5)
    # pillars (max_points 20)
    voxels, coors, num = OV.points_to_voxel(pts, scene.PILLAR_VOXEL, scene.PILLAR_RANGE, 20, 32000)
    out = H.voxelize(_dev(pts), scene.PILLAR_VOXEL, scene.PILLAR_RANGE, 20, 32000)
    _check_vox(out, voxels, coors, num, OV.voxel_mean(voxels, num), 20)


def test_voxelize_150k_bench_scene_matches_the_reference_golden(golden_dir):
    """the HIP voxelizer against the REFERENCE's own output on the 150 000-point bench scene (tests/golden/voxelize_150k.npz)"""
    g = np.load(os.path.join(golden_dir, "voxelize_150k.npz"))
    s = scene.make_scene(int(g["n_points"]), seed=int(g["seed"]), beam_jitter=float(g["beam_jitter"]))
    v, c, n, _ = [t.cpu().numpy() if t is not None else None for t in H.voxelize(_dev(s["points"]), scene.WAYMO_VOXEL, scene.WAYMO_RANGE, 5, 150000)]
    assert np.array_equal(c, g["coors"].astype(c.dtype)) and np.array_equal(n, g["num_points"].astype(n.dtype))
    v64 = v.astype(np.float64)
    assert np.array_equal(v64.sum((1, 2)).astype(np.float32), g["voxel_sums"]) and float((v64 * v64).sum()) == float(g["voxel_sq"])
    assert np.array_equal(v[:64], g["first_voxels"]) and np.array_equal(v[-64:], g["last_voxels"])


def test_voxelize_is_deterministic_and_idempotent():
    s = scene.make_scene(30000, seed=3)
    p = _dev(s["points"])
    a = H.voxelize(p, scene.WAYMO_VOXEL, scene.WAYMO_RANGE, 5, 150000)
    b = H.voxelize(p, scene.WAYMO_VOXEL, scene.WAYMO_RANGE, 5, 150000)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # re-voxelizing the voxel means (one point per voxel) reproduces the same coordinate set
    mean = a[3]
    c2 = H.voxelize(mean, scene.WAYMO_VOXEL, scene.WAYMO_RANGE, 5, 150000)[1]
    assert torch.equal(c2, a[1])


@pytest.mark.parametrize("max_voxels,max_points,voxel,rng", [(150000, 5, "WAYMO", None), (6000, 5, "WAYMO", None),
                                                              (32000, 20, "PILLAR", None)])
def test_batched_voxelizer_matches_per_frame_oracle(max_voxels, max_points, voxel, rng):
    """one launch chain for B frames == per-frame oracle + collate (batch index prepended), incl. an empty frame, a frame whose
    points all fall outside the range, and the max_voxels cut applied per frame"""
    vs, pr = (scene.WAYMO_VOXEL, scene.WAYMO_RANGE) if voxel == "WAYMO" else (scene.PILLAR_VOXEL, scene.PILLAR_RANGE)
    frames = [scene.make_scene(20000, seed=1)["points"], np.zeros((0, 5), np.float32), scene.make_scene(35000, seed=2)["points"],
              np.full((100, 5), 1e4, np.float32), scene.make_scene(9000, seed=3)["points"]]
    offs = np.concatenate([[0], np.cumsum([len(f) for f in frames])])
    out = H.voxelize_batch(_dev(np.concatenate(frames, 0)), offs.tolist(), vs, pr, max_points, max_voxels)
    ev, ec, en, em, cnt = [], [], [], [], []
    for b, f in enumerate(frames):
        v, c, n = OV.points_to_voxel(f, vs, pr, max_points, max_voxels)
        ev.append(v); en.append(n); cnt.append(len(n))
        ec.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], 1))
        em.append(OV.voxel_mean(v, n).reshape(len(n), 5))
    assert out[4].cpu().tolist() == cnt
    _check_vox(out[:4], np.concatenate(ev), np.concatenate(ec), np.concatenate(en), np.concatenate(em), max_points)


def test_batched_voxelizer_all_frames_empty():
    out = H.voxelize_batch(torch.zeros((0, 5), device=DEV), [0, 0, 0], scene.WAYMO_VOXEL, scene.WAYMO_RANGE, 5, 1000)
    assert out[0].shape[0] == 0 and out[1].shape == (0, 4) and out[4].cpu().tolist() == [0, 0]


# ------------------------------------------------------------------------------------------------
# rulebooks
# ------------------------------------------------------------------------------------------------
def _random_coors(rs, batch, shape, occupancy):
    cells = batch * shape[0] * shape[1] * shape[2]
    n = max(1, int(cells * occupancy))
    lin = rs.choice(cells, n, replace=False)
    c = np.zeros((n, 4), np.int32)
    c[:, 3] = lin % shape[2]; lin = lin // shape[2]
    c[:, 2] = lin % shape[1]; lin = lin // shape[1]
    c[:, 1] = lin % shape[0]; lin = lin // shape[0]
    c[:, 0] = lin
    return c


def _pairs_equal(rb_pairs, ref_pairs):
    assert len(rb_pairs) == len(ref_pairs)
    for k, ((ai, ao), (bi, bo)) in enumerate(zip(rb_pairs, ref_pairs)):
        a = np.stack([ai, ao], 1); b = np.stack([np.asarray(bi), np.asarray(bo)], 1)
        a = a[np.lexsort((a[:, 0], a[:, 1]))]; b = b[np.lexsort((b[:, 0], b[:, 1]))]
        assert a.shape == b.shape and np.array_equal(a, b), f"offset {k}: pair sets differ"


RB_CASES = [((9, 12, 11), 3, 1, 1, True), ((9, 12, 11), 3, 2, 1, False), ((11, 12, 12), 3, 2, (0, 1, 1), False),
            ((5, 8, 8), (3, 1, 1), (2, 1, 1), 0, False), ((41, 64, 48), 3, 1, 1, True), ((41, 64, 48), 3, 2, 1, False)]


@pytest.mark.parametrize("shape,ksize,stride,padding,subm", RB_CASES)
@pytest.mark.parametrize("occ", [0.03, 0.4])
def test_rulebook_matches_oracle(shape, ksize, stride, padding, subm, occ):
    rs = np.random.RandomState(abs(hash((shape, occ))) % 2 ** 31)
    coors = _random_coors(rs, 3, shape, occ)
    k3, s3, p3 = R._triple(ksize), R._triple(stride), R._triple(padding)
    if subm:
        ref = R.rulebook_subm(coors, shape, k3)
        rb = H.build_subm_rulebook(_dev(coors), 3, shape, k3)
        assert rb.n_out == coors.shape[0]
    else:
        oc, oshape, ref = R.rulebook_conv(coors, shape, k3, s3, p3)
        rb = H.build_conv_rulebook(_dev(coors), 3, shape, k3, s3, p3)
        assert rb.out_shape == oshape
        assert np.array_equal(rb.out_coors.cpu().numpy(), oc), "output rows not in canonical sorted order"
        # transposed map consistent with the gather map
        no, ni = rb.nbr_out.cpu().numpy(), rb.nbr_in.cpu().numpy()
        for k in range(rb.kvol):
            o = np.nonzero(no[k] >= 0)[0]
            assert np.array_equal(ni[k][no[k][o]], o)
            assert (ni[k] >= 0).sum() == o.size
    _pairs_equal(rb.pairs(), ref)
    assert np.array_equal(rb.pair_count.cpu().numpy(), np.array([len(p[0]) for p in ref], np.int32))


def test_rulebook_waymo_grid_full_scene():
    """Full-size property checks on the 150k-pt scene: extents of scn.py:118-149, pair counts vs
    the oracle, SubM symmetry k <-> K-1-k, centre offset = identity."""
    s = scene.make_scene(150000)
    _, coors3, _ = OV.points_to_voxel(s["points"], scene.WAYMO_VOXEL, scene.WAYMO_RANGE, 5, 150000)
    coors = np.concatenate([np.zeros((coors3.shape[0], 1), np.int32), coors3], 1)
    shape = (41, 1504, 1504)
    rb = H.build_subm_rulebook(_dev(coors), 1, shape, (3, 3, 3))
    ref = R.rulebook_subm(coors, shape, 3)
    _pairs_equal(rb.pairs(), ref)
    nb = rb.nbr_out.cpu().numpy()
    assert np.array_equal(nb[13], np.arange(coors.shape[0]))
    cnt = rb.pair_count.cpu().numpy()
    assert np.array_equal(cnt, cnt[::-1])
    cur, cshape = coors, shape
    for (k, st, p), expect in zip([((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)),
                                   ((3, 3, 3), (2, 2, 2), (0, 1, 1)), ((3, 1, 1), (2, 1, 1), (0, 0, 0))],
                                  [(21, 752, 752), (11, 376, 376), (5, 188, 188), (2, 188, 188)]):
        rbc = H.build_conv_rulebook(_dev(cur), 1, cshape, k, st, p)
        oc, oshape, refc = R.rulebook_conv(cur, cshape, k, st, p)
        assert rbc.out_shape == expect == oshape
        assert np.array_equal(rbc.out_coors.cpu().numpy(), oc)
        _pairs_equal(rbc.pairs(), refc)
        cur, cshape = oc, oshape


def test_rulebook_empty_and_out_of_range_rows():
    shape = (9, 12, 11)
    empty = torch.zeros((0, 4), dtype=torch.int32, device=DEV)
    rb = H.build_subm_rulebook(empty, 2, shape, (3, 3, 3))
    assert rb.n_out == 0 and int(rb.pair_count.sum()) == 0
    rbc = H.build_conv_rulebook(empty, 2, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1))
    assert rbc.n_out == 0
    # a row outside the grid is ignored (never indexes out of bounds)
    coors = np.array([[0, 1, 1, 1], [0, 1, 1, 2], [5, 1, 1, 1], [0, 100, 0, 0]], np.int32)
    rb = H.build_subm_rulebook(_dev(coors), 2, shape, (3, 3, 3))
    nb = rb.nbr_out.cpu().numpy()
    assert (nb[:, 2] == -1).all() and (nb[:, 3] == -1).all()
    assert nb[13, 0] == 0 and nb[14, 0] == 1 and nb[12, 1] == 0


# ------------------------------------------------------------------------------------------------
# sparse conv arithmetic
# ------------------------------------------------------------------------------------------------
CH = [(5, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (32, 16), (128, 64), (7, 9)]


@pytest.mark.parametrize("cin,cout", CH)
@pytest.mark.parametrize("subm", [True, False])
def test_spconv_forward_dgrad_wgrad_vs_oracle(cin, cout, subm):
    rs = np.random.RandomState(cin * 131 + cout)
    torch.manual_seed(cin * 7 + cout)
    shape = (11, 40, 36)
    coors = _random_coors(rs, 2, shape, 0.12)
    n = coors.shape[0]
    feats = torch.randn(n, cin)
    w = torch.randn(3, 3, 3, cin, cout) * (1.0 / (27 * cin) ** 0.5)
    b = torch.randn(cout) * 0.1 if subm else None
    if subm:
        pairs = R.rulebook_subm(coors, shape, 3)
        rb = H.build_subm_rulebook(_dev(coors), 2, shape, (3, 3, 3))
        n_out = n
    else:
        oc, _, pairs = R.rulebook_conv(coors, shape, 3, 2, 1)
        rb = H.build_conv_rulebook(_dev(coors), 2, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1))
        n_out = oc.shape[0]
    fr = feats.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    ref = R.sparse_conv(fr, wr, b, pairs, n_out)
    wk = w.reshape(27, cin, cout).to(DEV)
    out = H.spconv_gather_gemm(feats.to(DEV), wk, b.to(DEV) if b is not None else None, rb.nbr_out, n_out)
    torch.testing.assert_close(out.cpu(), ref.detach(), rtol=1e-4, atol=1e-5)

    g = torch.randn(n_out, cout)
    gin_ref, gw_ref = torch.autograd.grad(ref, [fr, wr], g)
    # data gradient = same kernel on the transposed map with transposed (and, for SubM, flipped) weights
    if subm:
        wt = wk.flip(0).transpose(1, 2).contiguous()
        gin = H.spconv_gather_gemm(g.to(DEV), wt, None, rb.nbr_out, n)
    else:
        wt = wk.transpose(1, 2).contiguous()
        gin = H.spconv_gather_gemm(g.to(DEV), wt, None, rb.nbr_in, n)
    torch.testing.assert_close(gin.cpu(), gin_ref, rtol=1e-4, atol=1e-5)
    gw = H.spconv_wgrad(feats.to(DEV), g.to(DEV), rb.nbr_out, 27)
    torch.testing.assert_close(gw.cpu().reshape(3, 3, 3, cin, cout), gw_ref, rtol=1e-4, atol=2e-4)  # K-dim = ~1e3 rows, order differs


def test_spconv_transpose_detecting_identity():
    """A = identity-like weights with an ASYMMETRIC pattern: catches row/col or k-order swaps."""
    shape = (5, 8, 8)
    coors = _random_coors(np.random.RandomState(0), 1, shape, 0.5)
    n = coors.shape[0]
    rb = H.build_subm_rulebook(_dev(coors), 1, shape, (3, 3, 3))
    cin = cout = 16
    w = torch.zeros(27, cin, cout)
    w[14] = torch.diag(torch.arange(1, 17).float()).roll(1, dims=1)  # offset (1,1,2): x+1 neighbour, shifted channels
    feats = torch.randn(n, cin)
    out = H.spconv_gather_gemm(feats.to(DEV), w.to(DEV), None, rb.nbr_out, n).cpu()
    lut = {tuple(c): i for i, c in enumerate(coors.tolist())}
    exp = torch.zeros(n, cout)
    for i, (b, z, y, x) in enumerate(coors.tolist()):
        j = lut.get((b, z, y, x + 1))
        if j is not None:
            exp[i] = (feats[j] * torch.arange(1, 17).float()).roll(1)
    torch.testing.assert_close(out, exp, rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------------------------------------
# BN1d / densify
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,c", [(1, 16), (777, 16), (5000, 32), (12345, 64), (3000, 128)])
def test_bn1d_kernels_vs_torch(n, c):
    torch.manual_seed(n + c)
    x = torch.randn(n, c) * 2 + 0.5
    res = torch.randn(n, c)
    gamma, beta = torch.rand(c) + 0.5, torch.randn(c) * 0.1
    eps = 1e-3
    xd = x.to(DEV)
    stats = H.bn1d_stats(xd).cpu().double()
    torch.testing.assert_close(stats[:c], x.double().sum(0), rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(stats[c:], (x.double() ** 2).sum(0), rtol=1e-5, atol=1e-3)
    mean = x.mean(0); var = x.var(0, unbiased=False)
    invstd = torch.rsqrt(var + eps)
    scale = gamma * invstd; shift = beta - mean * scale
    for relu in (False, True):
        for r in (None, res):
            y = H.bn1d_apply(xd, scale.to(DEV), shift.to(DEV), r.to(DEV) if r is not None else None, relu).cpu()
            e = x * scale + shift
            if r is not None:
                e = e + r
            if relu:
                e = e.relu()
            torch.testing.assert_close(y, e, rtol=1e-5, atol=1e-5)
    # backward of y = relu(bn(x) + res) in training mode vs autograd
    if n > 1:
        xr = x.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
        yr = (torch.nn.functional.batch_norm(xr, None, None, gr, br, True, 0.0, eps) + res).relu()
        dy = torch.randn(n, c)
        dx_ref, dg_ref, db_ref = torch.autograd.grad(yr, [xr, gr, br], dy)
        g, sums = H.bn1d_bwd_reduce(dy.to(DEV), yr.detach().to(DEV), xd, True)
        sg, sgx = sums[:c].cpu(), sums[c:].cpu()
        dbeta = sg; dgamma = invstd * (sgx - mean * sg)
        torch.testing.assert_close(dbeta, db_ref, rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(dgamma, dg_ref, rtol=1e-4, atol=2e-3)
        a = gamma * invstd
        bb = -gamma * invstd * invstd * dgamma / n
        d = -gamma * invstd * dbeta / n + gamma * invstd * invstd * dgamma * mean / n
        dx = H.bn1d_bwd_apply(g, xd, a.to(DEV), bb.to(DEV), d.to(DEV)).cpu()
        torch.testing.assert_close(dx, dx_ref, rtol=1e-4, atol=1e-5)


def test_densify_roundtrip():
    rs = np.random.RandomState(5)
    shape = (2, 20, 24)
    coors = _random_coors(rs, 3, shape, 0.3)
    feats = torch.randn(coors.shape[0], 128)
    dense = H.densify(feats.to(DEV), _dev(coors), 3, shape)
    ref = R.densify(feats, coors, shape, 3)
    assert torch.equal(dense.cpu(), ref)
    back = H.densify_bwd(dense, _dev(coors), 3, shape, 128)
    assert torch.equal(back.cpu(), feats)


# ------------------------------------------------------------------------------------------------
# bf16-input MFMA variant
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cin,cout", [(32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (64, 32), (128, 64), (32, 16)])
@pytest.mark.parametrize("subm", [True, False])
def test_spconv_bf16_vs_oracle(cin, cout, subm):
    """Exactness of the kernel: against the oracle fed with bf16-ROUNDED features and weights the
    only difference is fp32 summation order (rtol 1e-4); against the fp32 oracle the stated bf16
    tolerance (rtol 2e-2 of the output scale) holds."""
    rs = np.random.RandomState(cin * 17 + cout)
    torch.manual_seed(cin * 3 + cout)
    shape = (11, 40, 36)
    coors = _random_coors(rs, 2, shape, 0.12)
    n = coors.shape[0]
    feats = torch.randn(n, cin)
    w = torch.randn(3, 3, 3, cin, cout) * (1.0 / (27 * cin) ** 0.5)
    b = torch.randn(cout) * 0.1
    if subm:
        pairs = R.rulebook_subm(coors, shape, 3)
        rb = H.build_subm_rulebook(_dev(coors), 2, shape, (3, 3, 3)); n_out = n
    else:
        oc, _, pairs = R.rulebook_conv(coors, shape, 3, 2, 1)
        rb = H.build_conv_rulebook(_dev(coors), 2, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1)); n_out = oc.shape[0]
    H.set_sparse_compute_dtype("bf16")
    try:
        out = H.spconv_gather_gemm(feats.to(DEV), w.reshape(27, cin, cout).to(DEV), b.to(DEV), rb.nbr_out, n_out).cpu()
    finally:
        H.set_sparse_compute_dtype("f32")
    ref_r = R.sparse_conv(feats.bfloat16().float(), w.bfloat16().float(), b, pairs, n_out)
    torch.testing.assert_close(out, ref_r, rtol=1e-4, atol=1e-5)
    ref = R.sparse_conv(feats, w, b, pairs, n_out)
    assert (out - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()


@pytest.mark.parametrize("cin,cout", [(16, 16), (16, 32), (32, 32), (64, 64), (64, 128), (128, 128), (128, 64), (5, 16)])
def test_spconv_wgrad_bf16_vs_oracle(cin, cout):
    rs = np.random.RandomState(cin * 5 + cout)
    torch.manual_seed(cin + cout)
    shape = (11, 40, 36)
    coors = _random_coors(rs, 2, shape, 0.12)
    n = coors.shape[0]
    feats = torch.randn(n, cin)
    pairs = R.rulebook_subm(coors, shape, 3)
    rb = H.build_subm_rulebook(_dev(coors), 2, shape, (3, 3, 3))
    g = torch.randn(n, cout)
    H.set_sparse_compute_dtype("bf16")
    try:
        gw = H.spconv_wgrad(feats.to(DEV), g.to(DEV), rb.nbr_out, 27).cpu()
    finally:
        H.set_sparse_compute_dtype("f32")

    def ref_wgrad(f, d):
        out = torch.zeros(27, cin, cout, dtype=torch.float64)
        for k, (i_in, i_out) in enumerate(pairs):
            if len(i_in):
                out[k] = f[i_in].double().t() @ d[i_out].double()
        return out
    exact_rounded = ref_wgrad(feats.bfloat16().float(), g.bfloat16().float())
    torch.testing.assert_close(gw.double(), exact_rounded, rtol=1e-4, atol=2e-4)
    full = ref_wgrad(feats, g)
    assert (gw.double() - full).abs().max().item() <= 2e-2 * full.abs().max().item()


def test_distillation_example_fields_match_the_oracle_at_every_scale():
    """Voxelization.__call__ of the S2D example (preprocess.py:316-397): the input, `dense_*` and `reconstruction_*` clouds and the
    reconstruction clouds on the 2x / 4x voxel grids (`*_2`, `*_4`), each bit-exact vs the oracle per frame + collate (one host read
    serves all five voxelizations)"""
    from sparse2dense_amd.data import SyntheticFrames
    fr = SyntheticFrames(2, n_points=12000, seed=5, distill=True, device=DEV)
    ex = fr.example()
    clouds = {"": fr.points, "dense_": fr.dense_points, "reconstruction_": fr.recon_points}
    for prefix, suf in (("", ""), ("dense_", ""), ("reconstruction_", ""), ("reconstruction_", "_2"), ("reconstruction_", "_4")):
        gen = fr.gens[suf]
        ev, ec, en = [], [], []
        for b, pts in enumerate(clouds[prefix]):
            v, c, n = OV.points_to_voxel(pts.cpu().numpy(), gen.voxel_size, gen.point_cloud_range, gen.max_num_points_per_voxel, gen._max_voxels)
            ev.append(v); en.append(n)
            ec.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], 1))
        key = lambda f: prefix + f + suf
        assert np.array_equal(ex[key("coordinates")].cpu().numpy(), np.concatenate(ec)), (prefix, suf)
        assert np.array_equal(ex[key("num_points")].cpu().numpy(), np.concatenate(en))
        assert np.array_equal(ex[key("voxels")].cpu().numpy().view(np.uint32), np.concatenate(ev).view(np.uint32))
        assert ex[key("num_voxels")].cpu().tolist() == [len(n) for n in en]
