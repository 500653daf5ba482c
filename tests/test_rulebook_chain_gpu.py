"""r04 rulebook chain (csrc/rulebook_chain.hip: every rulebook of a backbone pass from the stage-0 coordinates, one host read)
against the oracle restatement of spconv's get_indice_pairs (oracle/spconv_ref.py; PARITY UNPINNED by the reference: spconv is
not in /root/reference) and against the per-layer builds of csrc/rulebook.hip.  Bit-exact: coordinates, both gather maps, pair counts."""
import numpy as np
import pytest
import torch

from oracle import spconv_ref as R
from oracle import voxelize as OV
from sparse2dense_amd import hip_ops as H
from sparse2dense_amd import scene

pytestmark = pytest.mark.gpu
DEV = "cuda"

WAYMO_CHAIN = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1)),
               ((3, 1, 1), (2, 1, 1), (0, 0, 0))]


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


def _check_against_oracle(coors, batch, shape, specs, want=None):
    want = [True] * (len(specs) + 1) if want is None else want
    subm, conv = H.build_rulebook_chain(torch.from_numpy(coors).to(DEV), batch, shape, specs, want)
    cur, cshape = coors, tuple(shape)
    for l in range(len(specs) + 1):
        if want[l]:
            ref = R.rulebook_subm(cur, cshape, (3, 3, 3))
            rb = subm[l]
            assert rb.n_out == cur.shape[0] and rb.out_shape == cshape
            _pairs_equal(rb.pairs(), ref)
            assert np.array_equal(rb.pair_count.cpu().numpy(), np.array([len(p[0]) for p in ref], np.int32)), f"SubM {l} pair counts"
        else:
            assert subm[l] is None
        if l == len(specs):
            break
        k, s, p = specs[l]
        oc, oshape, refc = R.rulebook_conv(cur, cshape, k, s, p)
        rb = conv[l]
        assert rb.out_shape == oshape and rb.n_in == cur.shape[0] and rb.n_out == oc.shape[0]
        assert np.array_equal(rb.out_coors.cpu().numpy().reshape(-1, 4), oc), f"conv {l}: output rows not in canonical sorted order"
        _pairs_equal(rb.pairs(), refc)
        no, ni = rb.nbr_out.cpu().numpy(), rb.nbr_in.cpu().numpy()
        for kk in range(rb.kvol):
            o = np.nonzero(no[kk] >= 0)[0]
            assert np.array_equal(ni[kk][no[kk][o]], o)
            assert (ni[kk] >= 0).sum() == o.size
        assert np.array_equal(rb.pair_count.cpu().numpy(), np.array([len(q[0]) for q in refc], np.int32)), f"conv {l} pair counts"
        cur, cshape = oc, oshape


@pytest.mark.parametrize("shape", [(9, 12, 11), (41, 64, 48), (8, 33, 65), (2, 2, 2)])
@pytest.mark.parametrize("occ", [0.03, 0.4])
@pytest.mark.parametrize("n_strided", [1, 2, 4])
def test_chain_matches_oracle(shape, occ, n_strided):
    rs = np.random.RandomState(abs(hash((shape, occ, n_strided))) % 2 ** 31)
    specs = WAYMO_CHAIN[:n_strided]
    if not H.rulebook_chain_supported(3, shape, specs):
        # grids too small for the later convs of the chain (kernel larger than the padded input): the host falls back per layer
        assert n_strided == 4 and shape != (41, 64, 48)
        return
    _check_against_oracle(_random_coors(rs, 3, shape, occ), 3, shape, specs)


def test_chain_without_some_subm_maps_and_with_bad_rows():
    shape = (17, 24, 31)
    rs = np.random.RandomState(5)
    coors = _random_coors(rs, 2, shape, 0.2)
    _check_against_oracle(coors, 2, shape, WAYMO_CHAIN[:3], want=[True, False, True, False])
    # rows outside the grid / batch are ignored everywhere and get -1 maps
    bad = np.array([[0, 1, 1, 1], [0, 1, 1, 2], [5, 1, 1, 1], [0, 100, 0, 0], [-1, 0, 0, 0], [1, 16, 23, 30]], np.int32)
    subm, conv = H.build_rulebook_chain(torch.from_numpy(bad).to(DEV), 2, shape, WAYMO_CHAIN[:2], [True, True, True])
    nb = subm[0].nbr_out.cpu().numpy()
    assert (nb[:, 2:5] == -1).all()
    assert nb[13, 0] == 0 and nb[14, 0] == 1 and nb[12, 1] == 0 and nb[13, 5] == 5
    ni = conv[0].nbr_in.cpu().numpy()
    assert (ni[:, 2:5] == -1).all()
    good = bad[[0, 1, 5]]
    oc, _, ref = R.rulebook_conv(good, shape, *WAYMO_CHAIN[0])
    assert np.array_equal(conv[0].out_coors.cpu().numpy(), oc)
    got = conv[0].pairs()
    remap = {0: 0, 1: 1, 5: 2}
    _pairs_equal([(np.array([remap[int(i)] for i in a]), b) for a, b in got], ref)


def test_chain_waymo_grid_full_scene_equals_the_per_layer_builds():
    """150 k-point scene at batch 2 (two seeds): the chain's maps equal the per-layer builds element for element (both number the
    outputs canonically), and batch 1 equals the oracle's pair sets on the full 41 x 1504 x 1504 grid."""
    frames = []
    for b, seed in enumerate((0, 7)):
        s = scene.make_scene(150000, seed=seed)
        _, c3, _ = OV.points_to_voxel(s["points"], scene.WAYMO_VOXEL, scene.WAYMO_RANGE, 5, 150000)
        frames.append(np.concatenate([np.full((c3.shape[0], 1), b, np.int32), c3], 1))
    shape = (41, 1504, 1504)
    coors = np.concatenate(frames)
    dc = torch.from_numpy(coors).to(DEV)
    subm, conv = H.build_rulebook_chain(dc, 2, shape, WAYMO_CHAIN, [True, True, True, True, False])
    cur, cshape = dc, shape
    for l in range(5):
        if l < 4:
            ref = H.build_subm_rulebook(cur, 2, cshape, (3, 3, 3))
            assert torch.equal(subm[l].nbr_out, ref.nbr_out), f"SubM {l}"
            assert torch.equal(subm[l].pair_count, ref.pair_count)
        if l == 4:
            break
        ref = H.build_conv_rulebook(cur, 2, cshape, *WAYMO_CHAIN[l])
        assert conv[l].out_shape == ref.out_shape
        assert torch.equal(conv[l].out_coors, ref.out_coors), f"conv {l} coordinates"
        assert torch.equal(conv[l].nbr_out, ref.nbr_out), f"conv {l} nbr_out"
        assert torch.equal(conv[l].nbr_in, ref.nbr_in), f"conv {l} nbr_in"
        assert torch.equal(conv[l].pair_count, ref.pair_count)
        cur, cshape = ref.out_coors, ref.out_shape
    assert [tuple(c.out_shape) for c in conv] == [(21, 752, 752), (11, 376, 376), (5, 188, 188), (2, 188, 188)]
    # batch 1 against the oracle (first two stages: the oracle takes seconds per stage at this size)
    _check_against_oracle(frames[0], 1, shape, WAYMO_CHAIN[:2], want=[True, True, False])


def test_backbone_plan_uses_the_chain_and_equals_the_per_layer_plan(monkeypatch):
    from sparse2dense_amd.backbones import SpMiddleResNetFHD, build_geometry
    bb = SpMiddleResNetFHD(num_input_features=5)
    strided, subm_keys = bb._specs()
    rs = np.random.RandomState(11)
    shape = (41, 96, 80)
    coors = torch.from_numpy(_random_coors(rs, 2, shape, 0.05)).to(DEV)
    calls = []
    orig = H.build_rulebook_chain
    monkeypatch.setattr(H, "build_rulebook_chain", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    plan = build_geometry(coors, 2, shape, strided, subm_keys)
    assert calls, "the backbone plan did not go through the chain"
    monkeypatch.setenv("S2D_RULEBOOK", "layers")
    ref = build_geometry(coors, 2, shape, strided, subm_keys)
    assert set(plan) == set(ref)
    for k in ref:
        assert torch.equal(plan[k].nbr_out, ref[k].nbr_out) and torch.equal(plan[k].pair_count, ref[k].pair_count), k
        if not ref[k].subm:
            assert torch.equal(plan[k].nbr_in, ref[k].nbr_in) and torch.equal(plan[k].out_coors, ref[k].out_coors), k
