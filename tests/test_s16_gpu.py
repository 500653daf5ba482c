"""bf16-storage sparse stack ("s16" path): gather/implicit-GEMM forward, data gradient, weight gradient and the row-major
bf16 batch norm with residual, against fp32/fp64 torch restatements evaluated on the same bf16-rounded operands.

Tolerances: operands are rounded to bf16 once and seen identically by both sides; products accumulate in fp32; the
kernels round their bf16 outputs once (2^-8 relative) -> 8e-3 of max|ref| for bf16 outputs, 2e-3 for fp32 outputs.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ref_conv(feat, w, bias, nbr, n_out):
    out = torch.zeros(n_out, w.shape[2], device=feat.device, dtype=torch.float64)
    fr, wr = feat.double(), w.to(torch.bfloat16).double()
    for k in range(w.shape[0]):
        o = torch.nonzero(nbr[k] >= 0).squeeze(1)
        if o.numel():
            out[o] += fr[nbr[k][o].long()] @ wr[k]
    return out + (bias.double() if bias is not None else 0)


def _random_map(kvol, n_in, n_out, p_empty=0.6, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    nbr = torch.randint(0, n_in, (kvol, n_out), device=DEV, dtype=torch.int32, generator=g)
    nbr[torch.rand(kvol, n_out, device=DEV, generator=g) < p_empty] = -1
    if n_out > 64:
        nbr[:, 16:64] = -1   # whole 16-row tiles without a neighbour (the skipped-MFMA path)
    return nbr


@pytest.mark.parametrize("cin", [16, 32, 64, 128])
@pytest.mark.parametrize("cout", [16, 32, 64, 128])
@pytest.mark.parametrize("n_in,n_out,kvol", [(500, 333, 27), (257, 130, 3), (3000, 100000, 27), (40, 1, 27)])
def test_s16_forward_and_transposed(cin, cout, n_in, n_out, kvol):
    from sparse2dense_amd import hip_ops as H
    if n_out > 50000 and cin * cout > 32 * 64:
        n_out = 9100   # keep the float64 reference quick; the 64-/128-row tile templates are still both exercised
    torch.manual_seed(cin * 131 + cout)
    feat = torch.randn(n_in, cin, device=DEV).to(torch.bfloat16)
    w = torch.randn(kvol, cin, cout, device=DEV) * 0.1
    b = torch.randn(cout, device=DEV)
    nbr = _random_map(kvol, n_in, n_out)
    out = H.spconv_s16(feat, w, b, nbr, n_out)
    ref = _ref_conv(feat, w, b, nbr, n_out)
    assert out.dtype == torch.bfloat16 and out.shape == (n_out, cout)
    assert (out.double() - ref).abs().max() <= 8e-3 * ref.abs().max()
    # the data-gradient operand: weight stored [K, cout_fwd, cin_fwd] -> W^T, offsets mirrored
    out2 = H.spconv_s16(feat, w.transpose(1, 2).contiguous(), None, nbr, n_out, transpose=True, flip=True)
    ref2 = _ref_conv(feat, w.flip(0), None, nbr, n_out)
    assert (out2.double() - ref2).abs().max() <= 8e-3 * ref2.abs().max()


def test_s16_empty_and_unsupported():
    from sparse2dense_amd import _lib, hip_ops as H
    feat = torch.randn(10, 16, device=DEV).to(torch.bfloat16)
    w = torch.randn(27, 16, 16, device=DEV)
    out = H.spconv_s16(feat, w, None, torch.empty((27, 0), dtype=torch.int32, device=DEV), 0)
    assert out.shape == (0, 16)
    all_empty = torch.full((27, 70), -1, dtype=torch.int32, device=DEV)
    out = H.spconv_s16(feat, w, None, all_empty, 70)
    assert torch.count_nonzero(out) == 0
    with pytest.raises(_lib.S2DError):
        H.spconv_s16(torch.randn(10, 24, device=DEV).to(torch.bfloat16), torch.randn(27, 24, 16, device=DEV), None, all_empty, 70)


@pytest.mark.parametrize("cin,cout", [(16, 16), (16, 32), (32, 32), (64, 64), (64, 128), (128, 128), (128, 64)])
def test_s16_wgrad(cin, cout):
    from sparse2dense_amd import hip_ops as H
    torch.manual_seed(7)
    n_in, n_out, kvol = 2000, 3100, 27
    feat = torch.randn(n_in, cin, device=DEV).to(torch.bfloat16)
    dout = torch.randn(n_out, cout, device=DEV).to(torch.bfloat16)
    nbr = _random_map(kvol, n_in, n_out, p_empty=0.7, seed=3)
    dw = H.spconv_s16_wgrad(feat, dout, nbr, kvol)
    ref = torch.zeros(kvol, cin, cout, device=DEV, dtype=torch.float64)
    for k in range(kvol):
        o = torch.nonzero(nbr[k] >= 0).squeeze(1)
        ref[k] = feat[nbr[k][o].long()].double().t() @ dout[o].double()
    assert dw.dtype == torch.float32
    assert (dw.double() - ref).abs().max() <= 2e-3 * ref.abs().max()


@pytest.mark.parametrize("cin,cout", [(16, 16), (64, 64), (128, 128)])
@pytest.mark.parametrize("n_out,p_empty", [(1, 0.0), (63, 0.5), (4097, 0.0), (4097, 0.97), (20000, 1.0), (20000, 0.6)])
def test_s16_wgrad_pair_ring_edges(cin, cout, n_out, p_empty):
    """The weight-gradient kernels cut chunks of 32 (output row, neighbour) pairs from a ring that is filled 64 rows at a
    time: dense maps (64 pairs per window), nearly empty and empty maps, a single row, row counts off the window size."""
    from sparse2dense_amd import hip_ops as H
    torch.manual_seed(11)
    n_in, kvol = 1500, 27
    feat = torch.randn(n_in, cin, device=DEV).to(torch.bfloat16)
    dout = torch.randn(n_out, cout, device=DEV).to(torch.bfloat16)
    nbr = _random_map(kvol, n_in, n_out, p_empty=p_empty, seed=5)
    dw = H.spconv_s16_wgrad(feat, dout, nbr, kvol)
    ref = torch.zeros(kvol, cin, cout, device=DEV, dtype=torch.float64)
    for k in range(kvol):
        o = torch.nonzero(nbr[k] >= 0).squeeze(1)
        if o.numel():
            ref[k] = feat[nbr[k][o].long()].double().t() @ dout[o].double()
    assert torch.isfinite(dw).all()
    assert (dw.double() - ref).abs().max() <= 2e-3 * max(ref.abs().max().item(), 1e-6)


@pytest.mark.parametrize("relu,with_res", [(True, True), (False, True), (True, False)])
@pytest.mark.parametrize("n,c", [(5000, 16), (777, 128), (33, 32)])
def test_feature_bn_bf16_rows_with_residual(n, c, relu, with_res):
    """FeatureBatchNorm1d on bf16 [n, c] features (+residual, +ReLU) vs a float64 CPU batch norm."""
    from sparse2dense_amd.spconv import FeatureBatchNorm1d
    torch.manual_seed(11)
    m = FeatureBatchNorm1d(c, eps=1e-3, momentum=0.01).to(DEV)
    with torch.no_grad():
        m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.5, 0.5)
    ref = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).double()
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    x = (torch.randn(n, c, device=DEV) * 2 + 0.3).to(torch.bfloat16)
    res = torch.randn(n, c, device=DEV).to(torch.bfloat16) if with_res else None
    dy = torch.randn(n, c, device=DEV).to(torch.bfloat16)
    xa = x.clone().requires_grad_(True)
    ra = res.clone().requires_grad_(True) if with_res else None
    y = m(xa, residual=ra, relu=relu)
    assert y.dtype == torch.bfloat16
    y.backward(dy)
    xr = x.double().cpu().requires_grad_(True)
    rr = res.double().cpu().requires_grad_(True) if with_res else None
    yr = ref(xr)
    if with_res:
        yr = yr + rr
    if relu:
        yr = torch.relu(yr)
    yr.backward(dy.double().cpu())

    def close(what, a, r, tol):
        err = (a.double().cpu() - r).abs().max() / r.abs().max().clamp(min=1e-9)
        assert err <= tol, (what, float(err))
    # y itself is compared away from the ReLU kink: a bf16-rounded y that is exactly 0 vs a tiny positive reference
    close("y", y, yr, 1e-2)
    close("dx", xa.grad, xr.grad, 2e-2)
    if with_res:
        # masked dy; rows where the fp64 reference and the bf16 kernel disagree on the sign of a ~0 pre-activation excluded
        stable = (yr.detach().abs() > 1e-2) | (not relu)
        diff = ((ra.grad.double().cpu() - rr.grad) * stable).abs().max()
        assert diff <= 1e-2 * rr.grad.abs().max(), float(diff)
    close("dgamma", m.weight.grad, ref.weight.grad, 5e-3)
    close("dbeta", m.bias.grad, ref.bias.grad, 5e-3)
    close("running_mean", m.running_mean, ref.running_mean, 1e-4)
    close("running_var", m.running_var, ref.running_var, 1e-4)
    assert int(m.num_batches_tracked) == 1


from sparse2dense_amd import hip_ops as H


@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 64), (64, 64), (128, 128), (64, 128), (16, 32)])
@pytest.mark.parametrize("n_in,n_out", [(3000, 9100), (500, 333), (40, 1)])
def test_s16_epilogue_statistics_equal_the_sums_of_the_stored_rows(cin, cout, n_in, n_out):
    """the per-workgroup (sum, sum of squares) rows of s2d_spconv_s16_fwd_stats fold to the column sums of the bf16 output the same
    launch stored (the statistics pass of the BatchNorm1d that follows), and the output is bit-identical to the plain launch"""
    from sparse2dense_amd import hip_ops as H
    torch.manual_seed(cin + cout + n_out)
    feat = torch.randn(n_in, cin, device=DEV).to(torch.bfloat16)
    w = torch.randn(27, cin, cout, device=DEV) * 0.1
    b = torch.randn(cout, device=DEV)
    nbr = _random_map(27, n_in, n_out)
    packed, kvol, ci, co = H.spconv_s16_pack(w, n_out)
    plain = H.spconv_s16_run(feat, packed, kvol, ci, co, b, nbr, n_out)
    out, partial = H.spconv_s16_run(feat, packed, kvol, ci, co, b, nbr, n_out, bn_stats=True)
    assert torch.equal(out, plain) and partial.shape[1:] == (2, cout)
    s1, s2 = out.double().sum(0), (out.double() ** 2).sum(0)
    assert (partial[:, 0].double().sum(0) - s1).abs().max() <= 1e-5 * out.double().abs().sum(0).max() + 1e-6
    assert (partial[:, 1].double().sum(0) - s2).abs().max() <= 1e-5 * s2.max() + 1e-6


def test_sparse_conv_bn_pair_with_epilogue_statistics_matches_the_separate_pass():
    """SubMConv3d -> BatchNorm1d -> ReLU (spconv.SparseSequential) in the bf16-storage mode: statistics from the conv epilogue vs the
    batch norm's own pass - outputs, running statistics and gradients agree to summation-order noise"""
    from sparse2dense_amd import hip_ops as H, spconv as SP
    old = H.SPARSE_COMPUTE_DTYPE
    H.SPARSE_COMPUTE_DTYPE = "s16"
    try:
        torch.manual_seed(3)
        g = torch.Generator().manual_seed(1)
        cells = torch.randperm(2 * 9 * 40 * 40, generator=g)[:6000].sort().values
        coors = torch.stack([cells // (9 * 40 * 40), (cells // (40 * 40)) % 9, (cells // 40) % 40, cells % 40], 1).int().to(DEV)
        feats = torch.randn(6000, 16, device=DEV)
        res = []
        for emit in (True, False):
            torch.manual_seed(7)
            seq = SP.SparseSequential(SP.SubMConv3d(16, 32, 3, bias=False, indice_key="a"), SP.FeatureBatchNorm1d(32, eps=1e-3, momentum=0.01),
                                      torch.nn.ReLU()).to(DEV).train()
            assert seq[0].emit_bn_stats
            seq[0].emit_bn_stats = emit
            f = feats.clone().requires_grad_(True)
            y = seq(SP.SparseConvTensor(f, coors, [9, 40, 40], 2)).features
            y.float().square().sum().backward()
            res.append((y.detach().float(), seq[1].running_mean.clone(), seq[1].running_var.clone(), f.grad.clone(), seq[0].weight.grad.clone()))
        for a, b in zip(*res):
            assert (a - b).abs().max() <= 2e-2 * b.abs().max() + 1e-6   # bf16 outputs: a last-bit flip of the statistics moves one rounding
        assert (res[0][1] - res[1][1]).abs().max() <= 1e-6 * res[1][1].abs().max() + 1e-8
        assert (res[0][2] - res[1][2]).abs().max() <= 1e-6 * res[1][2].abs().max() + 1e-8
    finally:
        H.SPARSE_COMPUTE_DTYPE = old


@pytest.mark.parametrize("cin,cout,n_in,n_out,kvol,p_empty", [
    (128, 128, 47890, 47890, 27, 0.44),   # conv4 SubM stage of the 4 x 150 k-point benchmark scene
    (128, 128, 47890, 38277, 3, 0.4),     # extra_conv
    (128, 128, 3000, 70000, 27, 0.4),     # more than one round of workgroups
    (64, 128, 126079, 47890, 27, 0.7),    # conv3 -> conv4
    (128, 64, 47890, 126079, 27, 0.7),    # its data gradient
    (64, 64, 126079, 126079, 27, 0.5),    # conv3 SubM stage (LDS-staged kernel)
])
def test_rg_kernel_at_benchmark_row_counts(cin, cout, n_in, n_out, kvol, p_empty):
    """The register-gather kernel deals more than one 16-row tile to a wave only above 32 768 output rows - the benchmark's conv4 /
    extra_conv row counts, which the small-shape tests above never reach.  r04: the <128,128,MI=2,WAVES=8> instantiation got the second
    tile of every wave wrong there (a third of the rows); rg_plan no longer selects it.  fp32 reference on the device, every row."""
    from sparse2dense_amd import hip_ops as H
    torch.manual_seed(cin + cout)
    feat = torch.randn(n_in, cin, device=DEV).to(torch.bfloat16)
    w = torch.randn(kvol, cin, cout, device=DEV) * 0.05
    nbr = _random_map(kvol, n_in, n_out, p_empty=p_empty, seed=1)
    ref = torch.zeros(n_out, cout, device=DEV, dtype=torch.float32)
    fr, wr = feat.float(), w.to(torch.bfloat16).float()
    for k in range(kvol):
        o = torch.nonzero(nbr[k] >= 0).squeeze(1)
        ref[o] += fr[nbr[k][o].long()] @ wr[k]
    for with_stats in (False, True):
        poison = torch.full((n_out, cout), float("nan"), device=DEV, dtype=torch.bfloat16)   # an unwritten row must not look written
        del poison
        packed, kv, ci, co = H.spconv_s16_pack(w, n_out)
        r = H.spconv_s16_run(feat, packed, kv, ci, co, None, nbr, n_out, None, "fwd", bn_stats=with_stats)
        out = (r[0] if with_stats else r).float()
        bad = ((out - ref).abs() > 1.2e-2 * ref.abs().max()).any(1) | (~torch.isfinite(out)).any(1)
        assert int(bad.sum()) == 0, (with_stats, int(bad.sum()), bad.nonzero().flatten()[:8].tolist())
        if with_stats:   # the statistics rows of the epilogue against the stored output
            part = r[1].double().sum(0)
            o64 = r[0].double()
            assert torch.allclose(part[0], o64.sum(0), rtol=1e-4, atol=1e-2) and torch.allclose(part[1], (o64 * o64).sum(0), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("cin,cout,n_in,n_out,kvol,p_empty", [
    (128, 128, 47890, 47890, 27, 0.44), (64, 64, 126079, 126079, 27, 0.5), (64, 128, 126079, 47890, 27, 0.7),
    (32, 32, 265026, 265026, 27, 0.5), (16, 16, 288892, 288892, 27, 0.75)])
def test_sparse_weight_gradient_at_benchmark_row_counts(cin, cout, n_in, n_out, kvol, p_empty):
    """dW of the sparse convs at the row counts of the 4 x 150 k-point benchmark scene (the small-shape test above stops at 3 100 rows):
    split plans, ring wrap-around and the workgroup-cooperative 128-channel kernel at their real sizes, against fp32 matmuls on the device"""
    from sparse2dense_amd import hip_ops as H
    torch.manual_seed(5)
    feat = torch.randn(n_in, cin, device=DEV).to(torch.bfloat16)
    dout = torch.randn(n_out, cout, device=DEV).to(torch.bfloat16)
    nbr = _random_map(kvol, n_in, n_out, p_empty=p_empty, seed=2)
    dw = H.spconv_s16_wgrad(feat, dout, nbr, kvol)
    ref = torch.zeros(kvol, cin, cout, device=DEV, dtype=torch.float64)
    for k in range(kvol):
        o = torch.nonzero(nbr[k] >= 0).squeeze(1)
        ref[k] = (feat[nbr[k][o].long()].double().t() @ dout[o].double())
    assert torch.isfinite(dw).all()
    assert (dw.double() - ref).abs().max() <= 2e-3 * ref.abs().max(), float((dw.double() - ref).abs().max() / ref.abs().max())
    assert torch.equal(dw, H.spconv_s16_wgrad(feat, dout, nbr, kvol))   # fixed-order split reduction: bit-reproducible


# ---- r06: rows grouped by neighbour mask (csrc/rulebook_sort.hip) and the implicit GEMM over them (spconv_rg_kernel<..., SORTED>) -------------
def _shaped_map(n, shapes, seed, p_drop=0.0):
    """a submanifold-like gather map [27][n] whose rows take their neighbour mask from a small set of `shapes` (bit masks; every mask keeps the
    centre offset 13, as a real rulebook does), rows in random order - what the sort has to group; p_drop: extra random holes"""
    g = torch.Generator(device=DEV).manual_seed(seed)
    pick = torch.randint(0, len(shapes), (n,), device=DEV, generator=g)
    masks = torch.tensor(shapes, device=DEV, dtype=torch.int64)[pick] | (1 << 13)
    bits = ((masks[None, :] >> torch.arange(27, device=DEV)[:, None]) & 1).bool()
    if p_drop:
        bits &= torch.rand(27, n, device=DEV, generator=g) >= p_drop
        bits[13] = True
    nbr = torch.randint(0, n, (27, n), device=DEV, dtype=torch.int32, generator=g)
    nbr[~bits] = -1
    return nbr


def _subm_rulebook(nbr):
    from sparse2dense_amd import hip_ops as H
    n = nbr.shape[1]
    return H.Rulebook(True, 27, n, n, nbr.contiguous(), None, (nbr >= 0).sum(1).int(), None, (41, 1504, 1504))


_SHAPES = {
    "plane": [0b000000000_111111111_000000000, 0b000000000_000111000_000000000, 0b010010010_010010010_010010010, 0b000010000_000010000_000010000],
    "mixed": [int(v) for v in np.random.RandomState(3).randint(1, 1 << 27, 40)],
    "full": [(1 << 27) - 1],
    "centre": [1 << 13],
}


@pytest.mark.parametrize("n,kind", [(47890, "mixed"), (126079, "plane"), (1000, "mixed"), (17, "full"), (1, "centre")])
def test_rulebook_rows_sorted_by_neighbour_mask(n, kind):
    from sparse2dense_amd import hip_ops as H
    nbr = _shaped_map(n, _SHAPES[kind], seed=n, p_drop=0.1 if kind == "mixed" else 0.0)
    perm, pmask, nbr_perm = H.rulebook_sorted_rows(_subm_rulebook(nbr))
    mask = ((nbr >= 0).long() << torch.arange(27, device=DEV)[:, None]).sum(0)
    pm = pmask.long() & 0xffffffff
    from sparse2dense_amd import _lib
    chunk = int(_lib.load().s2d_rulebook_sort_chunk_rows(n))   # the sort stays inside the rows one XCD's workgroups consume
    assert chunk >= 1 and -(-n // chunk) <= 8
    assert torch.equal(torch.sort(perm.long())[0], torch.arange(n, device=DEV))            # a permutation
    assert torch.equal(perm.long() // chunk, torch.arange(n, device=DEV) // chunk)          # ... of each chunk onto itself
    key = ((perm.long() // chunk) << 27) | pm
    assert bool((key[1:] >= key[:-1]).all())                                                # ascending masks inside a chunk
    assert torch.equal(pm, mask[perm.long()])                                               # ... of the rows they name
    same = key[1:] == key[:-1]
    assert bool((perm[1:][same] > perm[:-1][same]).all())                                   # stable: equal masks keep the canonical order
    assert torch.equal(nbr_perm, nbr[:, perm.long()])


@pytest.mark.parametrize("c,n,kind,p_drop", [
    (128, 47890, "mixed", 0.1),     # conv4 SubM stage of the 4 x 150 k-point benchmark scene (two tiles per wave: above 32 768 rows)
    (128, 47890, "plane", 0.0),     # few offsets per workgroup: short step lists, odd step counts
    (64, 126079, "mixed", 0.2),     # conv3 SubM stage
    (64, 126079, "plane", 0.0),
    (128, 70000, "full", 0.0),      # nothing to skip: every offset in every tile
    (64, 3000, "centre", 0.0),      # one offset: a single (padded) step
    (128, 333, "mixed", 0.3), (64, 1, "centre", 0.0), (64, 40, "full", 0.5),
])
def test_rg_kernel_over_mask_sorted_rows_matches_fp32_at_benchmark_row_counts(c, n, kind, p_drop):
    """the sorted-row instantiations against an fp32 restatement on the device, every row, with and without the statistics epilogue,
    and against the plain kernel's output (same packed image): the stored rows must be in the CANONICAL order"""
    from sparse2dense_amd import hip_ops as H
    torch.manual_seed(c + n)
    nbr = _shaped_map(n, _SHAPES[kind], seed=7 * n + c, p_drop=p_drop)
    rb = _subm_rulebook(nbr)
    was = H.set_sorted_rows(True)
    try:
        assert H.spconv_s16_sorted_ok(rb, 27, c, c, n)
        _check_sorted_kernel(H, c, n, nbr, rb)
    finally:
        H.set_sorted_rows(was)


def _check_sorted_kernel(H, c, n, nbr, rb):
    feat = torch.randn(n, c, device=DEV).to(torch.bfloat16)
    w = torch.randn(27, c, c, device=DEV) * 0.05
    bias = torch.randn(c, device=DEV)
    ref = torch.zeros(n, c, device=DEV, dtype=torch.float32) + bias
    fr, wr = feat.float(), w.to(torch.bfloat16).float()
    for k in range(27):
        o = torch.nonzero(nbr[k] >= 0).squeeze(1)
        ref[o] += fr[nbr[k][o].long()] @ wr[k]
    packed, kv, ci, co = H.spconv_s16_pack(w, n)
    plain = H.spconv_s16_run(feat, packed, kv, ci, co, bias, nbr, n, None, "fwd").float()
    for with_stats in (False, True):
        r = H.spconv_s16_run_sorted(feat, packed, kv, ci, co, bias, rb, "fwd", bn_stats=with_stats)
        out = (r[0] if with_stats else r).float()
        bad = ((out - ref).abs() > 1.2e-2 * ref.abs().max()).any(1) | (~torch.isfinite(out)).any(1)
        assert int(bad.sum()) == 0, (with_stats, int(bad.sum()), bad.nonzero().flatten()[:8].tolist())
        # same products, same fp32 accumulation order per row (offsets ascending in both kernels): the two kernels agree to the last bf16 ulp
        assert float((out - plain).abs().max()) <= 8e-3 * float(ref.abs().max())
        if with_stats:
            part = r[1].double().sum(0)
            o64 = r[0].double()
            assert torch.allclose(part[0], o64.sum(0), rtol=1e-4, atol=1e-2) and torch.allclose(part[1], (o64 * o64).sum(0), rtol=1e-4, atol=1e-2)


def test_sparse_conv_layer_uses_the_sorted_rows_and_matches_the_plain_kernel():
    """spconv.SubMConv3d in the bf16-storage mode: forward and data gradient through the sorted-row kernel == through the plain one
    (S2D_RG_SORTED=0 path), 64 channels"""
    from sparse2dense_amd import hip_ops as H
    from sparse2dense_amd.spconv import _SparseConvFn
    n, c = 20000, 64
    nbr = _shaped_map(n, _SHAPES["mixed"], seed=5, p_drop=0.1)
    rb = _subm_rulebook(nbr)
    torch.manual_seed(1)
    w = (torch.randn(3, 3, 3, c, c, device=DEV) * 0.05).requires_grad_(True)
    x = torch.randn(n, c, device=DEV).to(torch.bfloat16).requires_grad_(True)
    g = torch.randn(n, c, device=DEV).to(torch.bfloat16)
    res = {}
    old = H.SORTED_ROWS
    try:
        for mode in (True, False):
            H.set_sorted_rows(mode)
            if hasattr(rb, "_sorted_rows"):
                del rb._sorted_rows
            x.grad = w.grad = None
            y = _SparseConvFn.apply(x, w, None, rb)
            y.backward(g)
            res[mode] = (y.detach().float(), x.grad.float(), w.grad.clone())
            assert hasattr(rb, "_sorted_rows") == mode
    finally:
        H.set_sorted_rows(old)
    for a, b in zip(res[True], res[False]):
        assert float((a - b).abs().max()) <= 8e-3 * float(b.abs().max())
