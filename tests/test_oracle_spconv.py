"""Pins the sparse-conv oracle (oracle/spconv_ref.py) against two independent formulations:
brute-force loops for the rulebooks, and F.conv3d on the densified tensor for the arithmetic
(the reference has no spconv source/tests -> self-generated known answers)."""
import numpy as np
import pytest
import torch

from oracle import spconv_ref as R


def random_coors(rs, batch, shape, occupancy):
    cells = batch * shape[0] * shape[1] * shape[2]
    n = max(1, int(cells * occupancy))
    lin = rs.choice(cells, n, replace=False)
    rs.shuffle(lin)
    c = np.zeros((n, 4), np.int32)
    c[:, 3] = lin % shape[2]; lin //= shape[2]
    c[:, 2] = lin % shape[1]; lin //= shape[1]
    c[:, 1] = lin % shape[0]; lin //= shape[0]
    c[:, 0] = lin
    return c


CASES = [
    # (shape, ksize, stride, padding, subm)
    ((9, 12, 11), 3, 1, 1, True),
    ((9, 12, 11), 3, 2, 1, False),
    ((11, 12, 12), 3, 2, (0, 1, 1), False),
    ((5, 8, 8), (3, 1, 1), (2, 1, 1), 0, False),
    ((6, 7, 9), 3, 1, 0, True),   # SubM ignores padding= (scn.py:105 passes none)
]


@pytest.mark.parametrize("shape,ksize,stride,padding,subm", CASES)
@pytest.mark.parametrize("occ", [0.05, 0.4])
def test_rulebook_vs_bruteforce(shape, ksize, stride, padding, subm, occ):
    rs = np.random.RandomState(hash((shape, occ)) % 2 ** 31)
    coors = random_coors(rs, 2, shape, occ)
    oc_b, pairs_b = R.rulebook_bruteforce(coors, shape, ksize, stride, padding, 1, subm)
    if subm:
        pairs = R.rulebook_subm(coors, shape, ksize)
        oc = coors
    else:
        oc, oshape, pairs = R.rulebook_conv(coors, shape, ksize, stride, padding)
        assert oshape == R.conv_out_shape(shape, ksize, stride, padding)
    assert np.array_equal(oc, oc_b)
    assert R.pairs_to_set(pairs) == pairs_b
    if subm:  # symmetry invariant k <-> K-1-k
        K = len(pairs)
        for k in range(K):
            a = set(zip(pairs[k][0].tolist(), pairs[k][1].tolist()))
            b = set(zip(pairs[K - 1 - k][1].tolist(), pairs[K - 1 - k][0].tolist()))
            assert a == b


@pytest.mark.parametrize("shape,ksize,stride,padding,subm", CASES)
def test_features_and_grads_vs_dense_conv(shape, ksize, stride, padding, subm):
    rs = np.random.RandomState(3)
    torch.manual_seed(0)
    coors = random_coors(rs, 2, shape, 0.2)
    cin, cout = 5, 7
    feats = torch.randn(coors.shape[0], cin, dtype=torch.float64, requires_grad=True)
    kt = R._triple(ksize)
    w = torch.randn(*kt, cin, cout, dtype=torch.float64, requires_grad=True)
    b = torch.randn(cout, dtype=torch.float64, requires_grad=True)
    if subm:
        pairs = R.rulebook_subm(coors, shape, ksize)
        oc, n_out = coors, coors.shape[0]
    else:
        oc, _, pairs = R.rulebook_conv(coors, shape, ksize, stride, padding)
        n_out = oc.shape[0]
    out = R.sparse_conv(feats, w, b, pairs, n_out)
    oc_d, out_d = R.dense_conv_reference(feats, coors, shape, 2, w, b, ksize, stride, padding, subm)
    assert np.array_equal(oc, oc_d)
    assert torch.allclose(out, out_d, rtol=1e-10, atol=1e-10)
    g = torch.randn_like(out)
    ga = torch.autograd.grad(out, [feats, w, b], g, retain_graph=True)
    gb = torch.autograd.grad(out_d, [feats, w, b], g)
    for x, y in zip(ga, gb):
        assert torch.allclose(x, y, rtol=1e-9, atol=1e-9)


def test_backbone_output_extents():
    """scn.py:118,128,138,149: [41,1504,1504]->[21,752,752]->[11,376,376]->[5,188,188]->[2,188,188]"""
    s = (41, 1504, 1504)
    s = R.conv_out_shape(s, 3, 2, 1); assert s == (21, 752, 752)
    s = R.conv_out_shape(s, 3, 2, 1); assert s == (11, 376, 376)
    s = R.conv_out_shape(s, 3, 2, (0, 1, 1)); assert s == (5, 188, 188)
    s = R.conv_out_shape(s, (3, 1, 1), (2, 1, 1), 0); assert s == (2, 188, 188)


def test_ref_backbone_runs_and_dense_view():
    torch.manual_seed(1)
    rs = np.random.RandomState(1)
    grid_xyz = (32, 32, 40)  # input_shape is (x,y,z); sparse shape = (z+1, y, x)
    coors = random_coors(rs, 2, (41, 32, 32), 0.02)
    net = R.RefSpMiddleResNetFHD(5)
    feats = torch.randn(coors.shape[0], 5)
    bev, ms = net(feats, coors, 2, np.array(grid_xyz))
    assert bev.shape == (2, 256, 4, 4)
    assert set(ms) == {"conv1", "conv2", "conv3", "conv4"}
    assert ms["conv4"].features.shape[1] == 128
    bev.sum().backward()
    assert net.conv_input[0].weight.grad is not None
    net2 = R.RefSpMiddleFHD(5)
    bev2, c4 = net2(feats, coors, 2, np.array(grid_xyz))
    assert bev2.shape == (2, 128, 4, 4)
