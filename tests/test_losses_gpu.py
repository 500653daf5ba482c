"""PCR losses from the sparse recon voxels (csrc/losses.hip) against the reference's dense formulation
(/root/reference/det3d/models/detectors/voxelnet.py:171-185,203-249 = heads.mask_offset_loss on the densified target and
heads.metric_grid, both pinned to the reference by tests/golden/losses.npz) evaluated in float64 on the host.
Tolerance: fp32 sums of ~1e5..1e7 softplus terms: 1e-5 relative on the values, 1e-5 of max on the gradients."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from sparse2dense_amd import heads


def _case(b, d, h, w, m, seed, special=True):
    g = torch.Generator().manual_seed(seed)
    cells = torch.randperm(b * d * h * w, generator=g)[:m]
    coors = torch.stack([cells // (d * h * w), (cells // (h * w)) % d, (cells // w) % h, cells % w], 1).int()
    grid = heads.metric_grid(b, d, h, w, torch.zeros(1))
    centre = grid[coors[:, 0].long(), :, coors[:, 1].long(), coors[:, 2].long(), coors[:, 3].long()]   # [m,3]
    feats = torch.cat([centre + torch.randn(m, 3, generator=g) * 0.05, torch.rand(m, 2, generator=g)], 1).float()
    if special and m >= 4:
        feats[0] = torch.tensor([1.0, -1.0, 0.25, -0.25, 0.0])       # feature sum exactly 0: not an occupied cell, but tgt = f != 0
        feats[1, 0] = centre[1, 0]                                     # one target component exactly 0: not selected
        feats[2] = 0.0                                                 # all-zero voxel: contributes nothing
    gen_off = torch.randn(b, 3, d, h, w, generator=g)
    gen_mask = torch.randn(b, 1, d, h, w, generator=g) * 3
    return coors, feats, gen_off, gen_mask


@pytest.mark.parametrize("b,d,h,w,m,pad", [(2, 6, 20, 24, 300, 0), (1, 5, 17, 9, 40, 0), (4, 10, 94, 94, 20000, 0),
                                             (4, 1, 468, 468, 9000, 3000)])   # last: the pillar grid (point_pillars.py:180-215) with a padded list
def test_pcr_sparse_losses_match_dense_formulation(b, d, h, w, m, pad):
    coors, feats, gen_off, gen_mask = _case(b, d, h, w, m, seed=b * 7 + m)
    # reference: dense target + metric grid, float64, host
    gt = torch.zeros(b, d, h, w, 5, dtype=torch.float64)
    c = coors.long()
    gt[c[:, 0], c[:, 1], c[:, 2], c[:, 3]] = feats.double()
    gt = gt.permute(0, 4, 1, 2, 3).contiguous()
    grid = heads.metric_grid(b, d, h, w, torch.zeros(1)).double()
    # the grid must be the fp32 one the reference computes (tgt != 0 is an exact test)
    ro, rm = gen_off.double().requires_grad_(True), gen_mask.double().requires_grad_(True)
    ml_ref, ol_ref = heads.mask_offset_loss(ro, rm, gt, grid)
    (1.7 * ml_ref + 0.6 * ol_ref).backward()
    go = gen_off.cuda().requires_grad_(True)
    gm = gen_mask.cuda().requires_grad_(True)
    if pad:   # rows past the list: batch index -1, skipped by the kernels (the capacity-sized static buffers of the graphed segment)
        coors = torch.cat([coors, torch.full((pad, 4), -1, dtype=torch.int32)])
        feats = torch.cat([feats, torch.randn(pad, 5)])
    ml, ol = heads.mask_offset_loss_sparse(go, gm, coors.cuda(), feats.cuda())
    (1.7 * ml + 0.6 * ol).backward()
    np.testing.assert_allclose(ml.item(), ml_ref.item(), rtol=1e-5)
    np.testing.assert_allclose(ol.item(), ol_ref.item(), rtol=1e-5)
    for a, r in ((go.grad, ro.grad), (gm.grad, rm.grad)):
        assert (a.double().cpu() - r).abs().max() <= 1e-5 * r.abs().max(), float((a.double().cpu() - r).abs().max() / r.abs().max())
    assert int((go.grad != 0).sum()) == int((ro.grad != 0).sum())


@pytest.mark.parametrize("b,c,co,d,h,w,m", [(2, 32, 16, 6, 20, 24, 300), (2, 3, 0, 6, 20, 24, 300), (1, 32, 0, 4, 10, 12, 50),
                                              (3, 32, 16, 10, 94, 94, 20000), (2, 3, 0, 20, 188, 188, 30000)])
def test_pcr_level_heads_match_conv_plus_dense_loss(b, c, co, d, h, w, m):
    """fused level (csrc/losses.hip "Fused PCR level heads") == 1x1x1 convs -> mask_offset_loss on the dense target, float64 host:
    loss values, d/dg, every head parameter gradient, and the next conv's output and gradients"""
    from torch import nn
    coors, feats, _, _ = _case(b, d, h, w, m, seed=b * 11 + m + c)
    gen = torch.Generator().manual_seed(5 + c + m)
    g0 = torch.randn(b, c, d, h, w, generator=gen).relu()   # a post-ReLU feature volume
    mask_conv, off_conv = nn.Conv3d(c, 1, 1), nn.Conv3d(c, 3, 1)
    nxt = nn.Conv3d(c, co, 1) if co else None
    with torch.no_grad():
        for mod in (mask_conv, off_conv, nxt):
            if mod is not None:
                mod.weight.copy_(torch.randn(mod.weight.shape, generator=gen) * 0.3)
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=gen) * 0.3)
    r = torch.randn(b, co, d, h, w, generator=gen) / (b * d * h * w) if co else None

    def run(dev, dtype, fused):
        mods = [None if mm is None else __import__("copy").deepcopy(mm).to(dev, dtype) for mm in (mask_conv, off_conv, nxt)]
        g = g0.to(dev, dtype).requires_grad_(True)
        if fused:
            ml, ol, z = heads.pcr_level(g, mods[0], mods[1], coors.to(dev), feats.to(dev), next_conv=mods[2])
        else:
            gt = torch.zeros(b, d, h, w, 5, dtype=dtype)
            cc = coors.long()
            gt[cc[:, 0], cc[:, 1], cc[:, 2], cc[:, 3]] = feats.to(dtype)
            gt = gt.permute(0, 4, 1, 2, 3).contiguous()
            grid = heads.metric_grid(b, d, h, w, torch.zeros(1)).to(dtype)
            ml, ol = heads.mask_offset_loss(mods[1](g), mods[0](g), gt, grid)
            z = mods[2](g) if co else None
        total = 1.7 * ml + 0.6 * ol
        if co:
            total = total + (z * r.to(dev, dtype)).sum()
        total.backward()
        grads = [g.grad] + [p.grad for mm in mods if mm is not None for p in (mm.weight, mm.bias)]
        return ml, ol, z, grads

    ml_r, ol_r, z_r, gr_r = run("cpu", torch.float64, False)
    assert heads.pcr_level_supported(g0.cuda(), nxt)
    ml, ol, z, gr = run("cuda", torch.float32, True)
    np.testing.assert_allclose(ml.item(), ml_r.item(), rtol=2e-5)
    np.testing.assert_allclose(ol.item(), ol_r.item(), rtol=2e-5)
    if co:
        assert (z.double().cpu() - z_r).abs().max() <= 1e-5 * z_r.abs().max()
    for a, ref in zip(gr, gr_r):
        err = float((a.double().cpu() - ref).abs().max() / ref.abs().max())
        assert err <= 3e-5, err


@pytest.mark.parametrize("b,c,co,d,h,w,m", [(2, 32, 16, 6, 20, 24, 300), (2, 3, 0, 6, 20, 24, 300), (1, 32, 0, 4, 10, 12, 50),
                                              (3, 32, 16, 10, 94, 94, 20000), (2, 3, 0, 20, 188, 188, 30000)])
def test_pcr_level_with_folded_batchnorm_matches_unfused(b, c, co, d, h, w, m):
    """heads.pcr_level_norm (BatchNorm3d + ReLU + heads + losses + next conv from the RAW up-sampler output) == BatchNorm3d (train) ->
    ReLU -> 1x1x1 convs -> mask_offset_loss on the dense target in float64 on the host: losses, z, d/dy, every parameter gradient
    (batch norm included) and the running statistics"""
    import copy
    from torch import nn
    from sparse2dense_amd.dense3d import FastBatchNorm3d
    coors, feats, _, _ = _case(b, d, h, w, m, seed=b * 13 + m + c)
    gen = torch.Generator().manual_seed(9 + c + m)
    y0 = torch.randn(b, c, d, h, w, generator=gen) * 1.5 + 0.2
    bn = FastBatchNorm3d(c, fused_relu=True)
    mask_conv, off_conv = nn.Conv3d(c, 1, 1), nn.Conv3d(c, 3, 1)
    nxt = nn.Conv3d(c, co, 1) if co else None
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=gen) + 0.5); bn.bias.copy_(torch.randn(c, generator=gen) * 0.3)
        for mod in (mask_conv, off_conv, nxt):
            if mod is not None:
                mod.weight.copy_(torch.randn(mod.weight.shape, generator=gen) * 0.3)
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=gen) * 0.3)
    r = torch.randn(b, co, d, h, w, generator=gen) / (b * d * h * w) if co else None

    def run(dev, dtype, fused):
        mods = [None if mm is None else copy.deepcopy(mm).to(dev, dtype) for mm in (bn, mask_conv, off_conv, nxt)]
        mods[0].train()
        y = y0.to(dev, dtype).requires_grad_(True)
        if fused:
            ml, ol, z = heads.pcr_level_norm(y, mods[0], mods[1], mods[2], coors.to(dev), feats.to(dev), next_conv=mods[3])
        else:
            g = torch.relu(nn.functional.batch_norm(y, mods[0].running_mean, mods[0].running_var, mods[0].weight, mods[0].bias, True,
                                                    mods[0].momentum, mods[0].eps))
            gt = torch.zeros(b, d, h, w, 5, dtype=dtype)
            cc = coors.long()
            gt[cc[:, 0], cc[:, 1], cc[:, 2], cc[:, 3]] = feats.to(dtype)
            gt = gt.permute(0, 4, 1, 2, 3).contiguous()
            grid = heads.metric_grid(b, d, h, w, torch.zeros(1)).to(dtype)
            ml, ol = heads.mask_offset_loss(mods[2](g), mods[1](g), gt, grid)
            z = mods[3](g) if co else None
        total = 1.7 * ml + 0.6 * ol
        if co:
            total = total + (z * r.to(dev, dtype)).sum()
        total.backward()
        grads = [y.grad] + [p.grad for mm in mods if mm is not None for p in (mm.weight, mm.bias)]
        return ml, ol, z, grads, mods[0]

    ml_r, ol_r, z_r, gr_r, bn_r = run("cpu", torch.float64, False)
    ml, ol, z, gr, bn_h = run("cuda", torch.float32, True)
    np.testing.assert_allclose(ml.item(), ml_r.item(), rtol=3e-5)
    np.testing.assert_allclose(ol.item(), ol_r.item(), rtol=3e-5)
    if co:
        assert (z.double().cpu() - z_r).abs().max() <= 2e-5 * z_r.abs().max()
        # the kernel that wrote z also hands the statistics of the BatchNorm3d behind it over (sum | sum of squares per channel)
        st = getattr(z, "_s2d_bn_stats", None)
        assert st is not None and st.numel() == 2 * co
        zd = z.detach().double()
        ref_st = torch.cat([zd.sum((0, 2, 3, 4)), (zd * zd).sum((0, 2, 3, 4))]).cpu()
        assert (st.double().cpu() - ref_st).abs().max() <= 1e-5 * ref_st.abs().max() + 1e-3
    names = ["dy", "dgamma", "dbeta", "dw_mask", "db_mask", "dw_off", "db_off", "dw2", "db2"]
    for name, a, ref in zip(names, gr, gr_r):
        err = float((a.double().cpu() - ref).abs().max() / ref.abs().max())
        assert err <= 2e-4, (name, err)
    assert (bn_h.running_mean.double().cpu() - bn_r.running_mean).abs().max() <= 1e-5
    assert (bn_h.running_var.double().cpu() - bn_r.running_var).abs().max() <= 1e-4 * bn_r.running_var.abs().max()


@pytest.mark.parametrize("sdt,tdt,s_cl,t_cl", [(torch.bfloat16, torch.bfloat16, True, True), (torch.bfloat16, torch.float32, True, False),
                                              (torch.float32, torch.float32, False, False), (torch.float32, torch.bfloat16, False, True)])
@pytest.mark.parametrize("shape", [(4, 256, 47, 48), (2, 8, 5, 12)])
def test_masked_mse_pair_fused_matches_float64(sdt, tdt, s_cl, t_cl, shape):
    """heads.masked_mse_pair (trainer.py:783-789) through the fused kernels vs float64 on the host over the same stored values:
    value 1e-5 relative; gradient one output rounding (6e-3 of max for a bf16 student, 1e-5 for fp32)."""
    g = torch.Generator().manual_seed(3)
    fmt = lambda cl: torch.channels_last if cl else torch.contiguous_format
    s0 = torch.randn(shape, generator=g).to(sdt).contiguous(memory_format=fmt(s_cl))
    t0 = torch.randn(shape, generator=g).relu().to(tdt).contiguous(memory_format=fmt(t_cl))   # ~half the teacher entries are 0
    sr = s0.double().requires_grad_(True)
    tr = t0.double()
    pos = tr > 0
    ref = 10.0 * ((sr - tr)[pos] ** 2).mean() + 20.0 * ((sr - tr)[~pos] ** 2).mean()
    ref.backward()
    sh = s0.cuda().requires_grad_(True)
    out = heads.masked_mse_pair(sh, t0.cuda(), 10.0, 20.0)
    (out * 1.0).backward()
    np.testing.assert_allclose(out.item(), ref.item(), rtol=1e-5)
    assert sh.grad.dtype == sdt and sh.grad.stride() == sh.stride()
    err = float((sh.grad.double().cpu() - sr.grad).abs().max() / sr.grad.abs().max())
    assert err <= (6e-3 if sdt == torch.bfloat16 else 1e-5), err


def _focal_ref(out, target, ind, mask, cat):
    """centernet_loss.py:33-54 composed from torch ops (float64 on the host)"""
    mask = mask.double()
    neg = (torch.log(1 - out) * out.pow(2) * (1 - target).pow(4)).sum()
    b, c = out.shape[:2]
    pos_pix = out.permute(0, 2, 3, 1).reshape(b, -1, c).gather(1, ind.unsqueeze(2).expand(b, ind.shape[1], c))
    pos_pred = pos_pix.gather(2, cat.unsqueeze(2))
    num_pos = mask.sum()
    pos = (torch.log(pos_pred) * (1 - pos_pred).pow(2) * mask.unsqueeze(2)).sum()
    return -neg if num_pos == 0 else -(pos + neg) / num_pos


def _reg_ref(output, mask, ind, target):
    """centernet_loss.py:9-31"""
    b, c = output.shape[:2]
    pred = output.permute(0, 2, 3, 1).reshape(b, -1, c).gather(1, ind.unsqueeze(2).expand(b, ind.shape[1], c))
    m = mask.double().unsqueeze(2)
    loss = torch.nn.functional.l1_loss(pred * m, target * m, reduction="none") / (m.sum() + 1e-4)
    return loss.transpose(2, 0).sum(dim=2).sum(dim=1)


@pytest.mark.parametrize("b,c,h,w,m,n_pos", [(4, 3, 188, 188, 500, 137), (2, 1, 9, 7, 16, 5), (1, 3, 20, 12, 8, 0), (3, 2, 33, 40, 64, 64)])
def test_fused_focal_loss_vs_torch_float64(b, c, h, w, m, n_pos):
    """heads.fast_focal_loss (csrc/center_loss.hip) vs the reference's torch composition in float64 on the host: value 2e-5, gradient
    1e-4 of max; two objects in one centre cell (duplicate indices) and the no-positive batch included"""
    from sparse2dense_amd.heads import fast_focal_loss
    g = torch.Generator().manual_seed(b * 100 + c * 10 + m)
    out = torch.rand(b, c, h, w, generator=g).clamp(1e-4, 1 - 1e-4)
    target = torch.rand(b, c, h, w, generator=g) ** 3
    ind = torch.randint(0, h * w, (b, m), generator=g)
    if m > 2:
        ind[:, 1] = ind[:, 0]   # two objects share a cell
    cat = torch.randint(0, c, (b, m), generator=g)
    if m > 2:
        cat[:, 1] = cat[:, 0]
    mask = torch.zeros(b, m, dtype=torch.uint8)
    mask.view(-1)[torch.randperm(b * m, generator=g)[:n_pos]] = 1
    if n_pos >= 2 and m > 2:
        mask[0, 0] = mask[0, 1] = 1
    od = out.double().requires_grad_(True)
    ref = _focal_ref(od, target.double(), ind, mask, cat)
    ref.backward()
    oc = out.cuda().requires_grad_(True)
    got = fast_focal_loss(oc, target.cuda(), ind.cuda(), mask.cuda(), cat.cuda())
    assert got.dtype == torch.float32 and got.dim() == 0
    (got * 1.7).backward()
    assert abs(got.item() - ref.item()) <= 2e-5 * abs(ref.item())
    gref = od.grad * 1.7
    assert (oc.grad.cpu().double() - gref).abs().max() <= 1e-4 * gref.abs().max()


@pytest.mark.parametrize("b,c,h,w,m,n_pos", [(4, 8, 188, 188, 500, 137), (2, 10, 9, 7, 16, 5), (1, 8, 20, 12, 8, 0)])
def test_fused_reg_loss_vs_torch_float64(b, c, h, w, m, n_pos):
    """heads.RegLoss (csrc/center_loss.hip) vs the reference's torch composition in float64: per-channel losses 1e-5, gradient 1e-5"""
    from sparse2dense_amd.heads import RegLoss
    g = torch.Generator().manual_seed(b * 100 + c * 10 + m + 1)
    feat = torch.randn(b, c, h, w, generator=g)
    target = torch.randn(b, m, c, generator=g)
    ind = torch.randint(0, h * w, (b, m), generator=g)
    if m > 2:
        ind[:, 1] = ind[:, 0]
    mask = torch.zeros(b, m, dtype=torch.uint8)
    mask.view(-1)[torch.randperm(b * m, generator=g)[:n_pos]] = 1
    if n_pos >= 2 and m > 2:
        mask[0, 0] = mask[0, 1] = 1
    wts = torch.rand(c, generator=g) + 0.5
    fd = feat.double().requires_grad_(True)
    ref = _reg_ref(fd, mask, ind, target.double())
    (ref * wts.double()).sum().backward()
    fc = feat.cuda().requires_grad_(True)
    got = RegLoss()(fc, mask.cuda(), ind.cuda(), target.cuda())
    assert got.shape == (c,)
    (got * wts.cuda()).sum().backward()
    assert (got.cpu().double() - ref.detach()).abs().max() <= 1e-5 * ref.detach().abs().max() + 1e-9
    assert (fc.grad.cpu().double() - fd.grad).abs().max() <= 1e-5 * fd.grad.abs().max() + 1e-12


@pytest.mark.parametrize("b,c,co,d,h,w,m", [(2, 32, 16, 6, 20, 24, 300), (2, 3, 0, 6, 20, 24, 300), (3, 32, 16, 10, 94, 94, 20000),
                                              (2, 3, 0, 20, 188, 188, 30000)])
def test_pcr_level_with_bf16_stored_up_sampler_output(b, c, co, d, h, w, m):
    """r04 (s2d_pcr_level_*_y16): the fused level reading its input y as a bf16 tensor == the same level on the fp32 tensor holding the
    SAME (bf16-representable) values - the storage type changes the bytes moved, not the arithmetic: losses, z, dy and every parameter
    gradient agree to fp32 round-off, the statistics handed in are used as they are."""
    import copy
    from torch import nn
    from sparse2dense_amd.dense3d import FastBatchNorm3d
    coors, feats, _, _ = _case(b, d, h, w, m, seed=b * 13 + m + c)
    gen = torch.Generator().manual_seed(19 + c + m)
    y0 = (torch.randn(b, c, d, h, w, generator=gen) * 1.5 + 0.2).to(torch.bfloat16)
    bn = FastBatchNorm3d(c, fused_relu=True)
    mask_conv, off_conv = nn.Conv3d(c, 1, 1), nn.Conv3d(c, 3, 1)
    nxt = nn.Conv3d(c, co, 1) if co else None
    if nxt is not None:
        nxt.bf16_compute = True
    r = torch.randn(b, co, d, h, w, generator=gen) / (b * d * h * w) if co else None

    def run(dtype):
        mods = [None if mm is None else copy.deepcopy(mm).to("cuda") for mm in (bn, mask_conv, off_conv, nxt)]
        if mods[3] is not None:
            mods[3].bf16_compute = True
        mods[0].train()
        yf = y0.to("cuda").float()
        stats = torch.cat([yf.double().sum((0, 2, 3, 4)), (yf.double() ** 2).sum((0, 2, 3, 4))]).float()
        y = y0.to("cuda", dtype).requires_grad_(True)
        y._s2d_bn_stats = stats
        ml, ol, z = heads.pcr_level_norm(y, mods[0], mods[1], mods[2], coors.to("cuda"), feats.to("cuda"), next_conv=mods[3])
        total = 1.7 * ml + 0.6 * ol
        if co:
            total = total + (z * r.to("cuda")).sum()
        total.backward()
        return ml, ol, z, [y.grad] + [p.grad for mm in mods if mm is not None for p in (mm.weight, mm.bias)]

    ml_r, ol_r, z_r, gr_r = run(torch.float32)
    ml, ol, z, gr = run(torch.bfloat16)
    np.testing.assert_allclose(ml.item(), ml_r.item(), rtol=1e-5)
    np.testing.assert_allclose(ol.item(), ol_r.item(), rtol=1e-5)
    if co:
        assert torch.equal(z, z_r)
    assert gr[0].dtype == torch.float32 or gr[0].dtype == torch.bfloat16
    for i, (a, ref) in enumerate(zip(gr, gr_r)):
        tol = 8e-3 if i == 0 and a.dtype == torch.bfloat16 else 2e-5    # (autograd stores the gradient of a bf16 leaf in bf16)
        err = float((a.double() - ref.double()).abs().max() / ref.double().abs().max())
        assert err <= tol, (i, err)


@pytest.mark.parametrize("cin,cout,co,b,d,h,w,m", [(32, 32, 16, 2, 3, 6, 12, 400), (16, 3, 0, 2, 3, 5, 16, 500), (16, 3, 0, 1, 2, 4, 12, 90),
                                                   (32, 3, 0, 1, 2, 4, 8, 60), (16, 32, 0, 1, 2, 3, 8, 50)])
def test_upsample_level_bf16_gradient_storage(cin, cout, co, b, d, h, w, m, monkeypatch):
    """r04 heads.upsample_level (ConvTranspose3d(4,2,1) + the fused level as one node): with the raw output y stored in bf16 the node also
    stores dy in bf16 (s2d_pcr_level_bwd_apply_y16_d16 -> s2d_convt3d_mfma_{dgrad,wgrad}_d16).  The up-sampler's matrix-core kernels round
    dy to bf16 when they load it, so against the same node with an fp32 dy (S2D_PCR_DY16=0) nothing changes except at the <= m recon
    cells, where the sparse correction is added to an already rounded value (one extra bf16 rounding): forward identical, every gradient
    produced before dy identical, dx / dW of the up-sampler within 2e-3 of their norm.  (The bf16 storage of y itself is pinned piecewise
    with tight bars by test_convtranspose3d_bf16_stored_output and test_pcr_level_with_bf16_stored_up_sampler_output; end to end on
    these tiny volumes the level's ReLU / L1-sign decisions flip with the rounding of y and move dx by 4e-2..7e-2 from run to run.)"""
    import copy
    from torch import nn
    from sparse2dense_amd import dense3d
    from sparse2dense_amd.dense3d import ConvTranspose3dK4S2, FastBatchNorm3d
    od, oh, ow = 2 * d, 2 * h, 2 * w
    coors, feats, _, _ = _case(b, od, oh, ow, m, seed=cin + cout + m)
    gen = torch.Generator().manual_seed(5 + cin + m)
    x0 = torch.randn(b, cin, d, h, w, generator=gen)
    ct = ConvTranspose3dK4S2(cin, cout, 4, 2, 1)
    ct.bf16_compute = True
    bn = FastBatchNorm3d(cout, fused_relu=True)
    mask_conv, off_conv = nn.Conv3d(cout, 1, 1), nn.Conv3d(cout, 3, 1)
    nxt = nn.Conv3d(cout, co, 1) if co else None
    r = torch.randn(b, co, od, oh, ow, generator=gen) / (b * od * oh * ow) if co else None
    seen = []
    orig = dense3d._ConvT3dFn.backward
    monkeypatch.setattr(dense3d._ConvT3dFn, "backward", staticmethod(lambda ctx, dout, *a, **k: (seen.append(dout.dtype), orig(ctx, dout, *a, **k))[1]))

    def run(y16, dy16):
        monkeypatch.setenv("S2D_PCR_DY16", "1" if dy16 else "0")
        mods = [None if mm is None else copy.deepcopy(mm).to("cuda") for mm in (ct, bn, mask_conv, off_conv, nxt)]
        mods[0].bf16_compute = True
        if mods[4] is not None:
            mods[4].bf16_compute = True
        mods[1].train()
        assert heads.upsample_level_supported(mods[0], (d, h, w), mods[4])
        x = x0.to("cuda").requires_grad_(True)
        ml, ol, z = heads.upsample_level(mods[0], x, mods[1], mods[2], mods[3], coors.to("cuda"), feats.to("cuda"), next_conv=mods[4], y16=y16)
        total = 1.7 * ml + 0.6 * ol
        if co:
            total = total + (z * r.to("cuda")).sum()
        total.backward()
        return ml, ol, z, [x.grad] + [p.grad for mm in mods if mm is not None for p in (mm.weight, mm.bias)]

    a = run(True, False)
    bb = run(True, True)
    assert seen == [torch.float32, torch.bfloat16], seen
    rel = lambda u, v: float((u.double() - v.double()).norm() / v.double().norm().clamp_min(1e-30))
    assert bb[0].item() == a[0].item() and bb[1].item() == a[1].item()
    if co:
        assert torch.equal(bb[2], a[2])
    for i, (u, v) in enumerate(zip(bb[3], a[3])):
        if i in (0, 1):                       # dx and the up-sampler's weight gradient read dy
            assert rel(u, v) <= 2e-3, (i, rel(u, v))
        elif i == 2:                          # its bias gradient: analytic, ~0 behind the batch norm (noise against noise)
            assert float((u - v).abs().max()) <= 1e-4 * float(a[3][1].abs().max()), i
        else:
            assert torch.equal(u, v), i


@pytest.mark.parametrize("cin,cout,co,b,d,h,w,m", [(16, 3, 0, 2, 3, 5, 16, 500), (16, 3, 0, 1, 2, 4, 8, 90), (32, 3, 0, 1, 2, 4, 8, 60)])
def test_upsample_level_with_the_batch_norm_in_front_folded_in(cin, cout, co, b, d, h, w, m):
    """r04 heads.upsample_level(pre_bn=...): the BatchNorm3d + ReLU in front of the up-sampler inside the node - the up-sampler's forward and
    weight-gradient kernels normalise the raw input on load (s2d_convt3d_mfma_fwd_stats_y16_norm / _wgrad_d16_norm), the normalised
    tensor is never written.  Same arithmetic as the separate FastBatchNorm3d node in front (relu(fma(x, scale, shift)), then the bf16
    rounding of the matrix-core operand): losses, z and every gradient - including the batch norm's and the raw input's - are equal, the
    running statistics are updated once."""
    import copy
    from torch import nn
    from sparse2dense_amd.dense3d import ConvTranspose3dK4S2, FastBatchNorm3d
    od, oh, ow = 2 * d, 2 * h, 2 * w
    coors, feats, _, _ = _case(b, od, oh, ow, m, seed=3 * cin + cout + m)
    gen = torch.Generator().manual_seed(11 + cin + m)
    x0 = torch.randn(b, cin, d, h, w, generator=gen) * 1.3 + 0.4
    pre = FastBatchNorm3d(cin, fused_relu=True)
    with torch.no_grad():
        pre.weight.copy_(torch.rand(cin, generator=gen) + 0.5)
        pre.bias.copy_(torch.randn(cin, generator=gen) * 0.3)
    ct = ConvTranspose3dK4S2(cin, cout, 4, 2, 1)
    bn = FastBatchNorm3d(cout, fused_relu=True)
    mask_conv, off_conv = nn.Conv3d(cout, 1, 1), nn.Conv3d(cout, 3, 1)
    nxt = nn.Conv3d(cout, co, 1) if co else None
    r = torch.randn(b, co, od, oh, ow, generator=gen) / (b * od * oh * ow) if co else None

    def run(fold):
        mods = [None if mm is None else copy.deepcopy(mm).to("cuda") for mm in (pre, ct, bn, mask_conv, off_conv, nxt)]
        mods[1].bf16_compute = True
        if mods[5] is not None:
            mods[5].bf16_compute = True
        mods[0].train(); mods[2].train()
        x = x0.to("cuda").requires_grad_(True)
        if fold:
            assert heads.upsample_level_pre_bn_supported(mods[1], (d, h, w), mods[0])
            ml, ol, z = heads.upsample_level(mods[1], x, mods[2], mods[3], mods[4], coors.to("cuda"), feats.to("cuda"), next_conv=mods[5], pre_bn=mods[0])
        else:
            ml, ol, z = heads.upsample_level(mods[1], mods[0](x), mods[2], mods[3], mods[4], coors.to("cuda"), feats.to("cuda"), next_conv=mods[5])
        total = 1.7 * ml + 0.6 * ol
        if co:
            total = total + (z * r.to("cuda")).sum()
        total.backward()
        return ml, ol, z, [x.grad] + [p.grad for mm in mods if mm is not None for p in (mm.weight, mm.bias)], mods[0]

    ref = run(False)
    got = run(True)
    assert got[0].item() == ref[0].item() and got[1].item() == ref[1].item()
    if co:
        assert torch.equal(got[2], ref[2])
    for i, (u, v) in enumerate(zip(got[3], ref[3])):
        err = float((u.double() - v.double()).abs().max() / v.double().abs().max().clamp_min(1e-30))
        assert err <= (1e-4 if i == 4 else 1e-6), (i, err)   # (i == 4: the up-sampler's bias gradient, noise around zero)
    assert torch.equal(got[4].running_mean, ref[4].running_mean) and torch.equal(got[4].running_var, ref[4].running_var)
    assert int(got[4].num_batches_tracked) == 1


def _level_modules(cin, cout, co, gen, pre=False):
    from torch import nn
    from sparse2dense_amd.dense3d import ConvTranspose3dK4S2, FastBatchNorm3d
    pre_bn = None
    if pre:
        pre_bn = FastBatchNorm3d(cin, fused_relu=True)
        with torch.no_grad():
            pre_bn.weight.copy_(torch.rand(cin, generator=gen) + 0.5)
            pre_bn.bias.copy_(torch.randn(cin, generator=gen) * 0.3)
    ct = ConvTranspose3dK4S2(cin, cout, 4, 2, 1)
    bn = FastBatchNorm3d(cout, fused_relu=True)
    return [pre_bn, ct, bn, nn.Conv3d(cout, 1, 1), nn.Conv3d(cout, 3, 1), nn.Conv3d(cout, co, 1) if co else None]


def _to_cuda(mods):
    import copy
    out = [None if mm is None else copy.deepcopy(mm).to("cuda") for mm in mods]
    out[1].bf16_compute = True
    if out[5] is not None:
        out[5].bf16_compute = True
    for mm in (out[0], out[2]):
        if mm is not None:
            mm.train()
    return out


@pytest.mark.parametrize("b,d,h,w,m", [(2, 2, 3, 8, 700), (1, 3, 5, 16, 900)])
def test_pcr_level_writes_z_and_reads_its_gradient_in_bf16(b, d, h, w, m):
    """r06 heads.upsample_level(z16=True) on the 32 -> 32 up-sampler + level + 32 -> 16 conv: z is STORED in bf16 (s2d_pcr_level_fwd_y16_z16) and
    the gradient that comes back for it is read in bf16 (s2d_pcr_level_bwd_{sums,apply}_*_z16, s2d_pointwise_conv_wgrad_norm_x16_d16).  Against
    the fp32-z node on the same inputs: losses identical, z == the fp32 z rounded to bf16 bit for bit, the z statistics handed to the next batch
    norm are those of the STORED values, and with a bf16-representable upstream gradient every gradient is identical bit for bit (the kernels
    widen bf16 exactly)."""
    od, oh, ow = 2 * d, 2 * h, 2 * w
    coors, feats, _, _ = _case(b, od, oh, ow, m, seed=77 + m)
    gen = torch.Generator().manual_seed(m)
    x0 = torch.randn(b, 32, d, h, w, generator=gen)
    mods0 = _level_modules(32, 32, 16, gen)
    r = (torch.randn(b, 16, od, oh, ow, generator=gen) / (b * od * oh * ow)).to(torch.bfloat16).float()

    def run(z16):
        mods = _to_cuda(mods0)
        x = x0.to("cuda").requires_grad_(True)
        ml, ol, z = heads.upsample_level(mods[1], x, mods[2], mods[3], mods[4], coors.to("cuda"), feats.to("cuda"), next_conv=mods[5], z16=z16)
        (1.7 * ml + 0.6 * ol + (z.float() * r.to("cuda")).sum()).backward()
        return ml, ol, z, [x.grad] + [p.grad for mm in mods if mm is not None for p in (mm.weight, mm.bias)]

    ref = run(False)
    got = run(True)
    assert ref[2].dtype == torch.float32 and got[2].dtype == torch.bfloat16
    assert got[0].item() == ref[0].item() and got[1].item() == ref[1].item()
    assert torch.equal(got[2], ref[2].to(torch.bfloat16))
    zs = got[2]._s2d_bn_stats.detach()
    zf = got[2].detach().double()
    want = torch.cat([zf.sum(dim=(0, 2, 3, 4)), (zf * zf).sum(dim=(0, 2, 3, 4))])
    assert float((zs.double() - want).abs().max() / want.abs().max()) <= 1e-5
    for i, (u, v) in enumerate(zip(got[3], ref[3])):
        assert torch.equal(u, v), (i, float((u.double() - v.double()).abs().max()))


@pytest.mark.parametrize("b,d,h,w,m", [(2, 4, 6, 16, 3000), (1, 2, 5, 24, 1500)])
def test_upsample_level_reads_a_bf16_stored_input_and_returns_its_gradient_in_bf16(b, d, h, w, m):
    """r06: the 16 -> 3 up-sampler node with the batch norm in front folded in, fed the bf16-STORED z (s2d_convt3d_mfma_fwd_stats_y16_norm_x16,
    _wgrad_d16_norm_x16, _dgrad_d16_x16, s2d_bncm_bwd_*_typed).  Against the same node fed the widened fp32 copy: the forward is the same
    arithmetic on the same values (losses and every gradient that does not pass the input gradient identical); the input gradient dx' is stored
    in bf16 before the batch norm's backward reads it (one more bf16 rounding, 2^-9 relative per element, over sums that largely cancel on these
    tiny volumes): the batch norm's dgamma / dbeta and the returned gradient - bf16 itself - within 1e-2 of their norm (measured 1e-3 .. 3e-3)."""
    from sparse2dense_amd import _lib
    od, oh, ow = 2 * d, 2 * h, 2 * w
    coors, feats, _, _ = _case(b, od, oh, ow, m, seed=5 + m)
    gen = torch.Generator().manual_seed(3 * m)
    x16 = (torch.randn(b, 16, d, h, w, generator=gen) * 1.3 + 0.4).to(torch.bfloat16)
    mods0 = _level_modules(16, 3, 0, gen, pre=True)
    xf = x16.double()
    stats = torch.cat([xf.sum(dim=(0, 2, 3, 4)), (xf * xf).sum(dim=(0, 2, 3, 4))]).float().to("cuda")
    assert _lib.load().s2d_convt3d_mfma_x16_supported(16, 3, d, h, w)

    def run(as_bf16):
        mods = _to_cuda(mods0)
        x = (x16 if as_bf16 else x16.float()).to("cuda").requires_grad_(True)
        x._s2d_bn_stats = stats
        assert heads.upsample_level_x16_supported(mods[1], (d, h, w), mods[0])
        ml, ol, _ = heads.upsample_level(mods[1], x, mods[2], mods[3], mods[4], coors.to("cuda"), feats.to("cuda"), pre_bn=mods[0])
        (1.7 * ml + 0.6 * ol).backward()
        return ml, ol, x.grad, [p.grad for mm in mods if mm is not None for p in (mm.weight, mm.bias)], mods[0]

    ref = run(False)
    got = run(True)
    assert got[2].dtype == torch.bfloat16 and ref[2].dtype == torch.float32
    assert got[0].item() == ref[0].item() and got[1].item() == ref[1].item()
    rel = lambda u, v: float((u.double() - v.double()).norm() / v.double().norm().clamp_min(1e-30))
    assert rel(got[2], ref[2]) <= 1e-2, rel(got[2], ref[2])
    for i, (u, v) in enumerate(zip(got[3], ref[3])):
        if i in (0, 1):     # the folded batch norm's dgamma / dbeta: sums over the bf16-stored dx'
            assert rel(u, v) <= 1e-2, (i, rel(u, v))
        else:
            assert torch.equal(u, v), (i, rel(u, v))
    assert torch.equal(got[4].running_mean, ref[4].running_mean) and torch.equal(got[4].running_var, ref[4].running_var)


@pytest.mark.parametrize("b,c,co,d,h,w,m,dtype", [(2, 32, 16, 6, 20, 24, 300, torch.bfloat16), (2, 3, 0, 6, 20, 24, 300, torch.bfloat16),
                                                  (3, 32, 16, 10, 94, 94, 20000, torch.bfloat16), (2, 3, 0, 20, 188, 188, 30000, torch.bfloat16),
                                                  (2, 32, 0, 4, 10, 12, 50, torch.float32), (1, 3, 0, 4, 10, 12, 1, torch.float32)])
def test_pcr_level_site_cache_changes_nothing_but_the_loads(b, c, co, d, h, w, m, dtype, monkeypatch):
    """r06 `s2d_pcr_level_site_cache`: the forward's per-voxel pass writes every recon voxel's c raw values as one row, the backward's per-voxel passes
    read the row instead of c scattered loads - the SAME values, so losses, z and every gradient are bit-equal to the run without the cache."""
    import copy
    from torch import nn
    from sparse2dense_amd.dense3d import FastBatchNorm3d
    coors, feats, _, _ = _case(b, d, h, w, m, seed=b * 7 + m + c, special=m >= 4)
    gen = torch.Generator().manual_seed(23 + c + m)
    y0 = (torch.randn(b, c, d, h, w, generator=gen) * 1.5 + 0.2).to(torch.bfloat16)
    mods0 = (FastBatchNorm3d(c, fused_relu=True), nn.Conv3d(c, 1, 1), nn.Conv3d(c, 3, 1), nn.Conv3d(c, co, 1) if co else None)
    r = torch.randn(b, co, d, h, w, generator=gen) / (b * d * h * w) if co else None

    def run(cache):
        monkeypatch.setenv("S2D_PCR_SITE_CACHE", "1" if cache else "0")
        mods = [None if mm is None else copy.deepcopy(mm).to("cuda") for mm in mods0]
        if mods[3] is not None:
            mods[3].bf16_compute = True
        mods[0].train()
        yf = y0.to("cuda").float()
        y = y0.to("cuda", dtype).requires_grad_(True)
        y._s2d_bn_stats = torch.cat([yf.double().sum((0, 2, 3, 4)), (yf.double() ** 2).sum((0, 2, 3, 4))]).float()
        ml, ol, z = heads.pcr_level_norm(y, mods[0], mods[1], mods[2], coors.to("cuda"), feats.to("cuda"), next_conv=mods[3])
        total = 1.7 * ml + 0.6 * ol
        if co:
            total = total + (z * r.to("cuda")).sum()
        total.backward()
        return [ml.detach(), ol.detach()] + ([z.detach()] if co else []) + [y.grad] + [p.grad for mm in mods if mm is not None for p in (mm.weight, mm.bias)]

    ref = run(False)
    got = run(True)
    for i, (u, v) in enumerate(zip(got, ref)):
        assert torch.equal(u, v), (i, float((u.double() - v.double()).abs().max()))

