"""Dense 3x3 NHWC bf16 MFMA convolution (csrc/conv2d_nhwc.hip) against a CPU convolution (torch on the host, fp32
accumulate - not MIOpen on the same GPU) on the same bf16-rounded operands.

Tolerance: inputs/weights are rounded to bf16 once (both sides see the rounded values), products accumulate in
fp32 on both sides, the kernel rounds its output to bf16 -> |err| <= 2^-8 relative to the output scale (+ accumulation
order noise ~1e-6); asserted at 6e-3 of max|ref|.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

def _cpu_conv(x, w, b, padding, stride=1):
    """the reference convolution: on the HOST (never MIOpen), returned on the device for comparison"""
    out = F.conv2d(x.detach().float().cpu().contiguous(), w.detach().float().cpu(), None if b is None else b.detach().float().cpu(),
                   padding=padding, stride=stride)
    return out.to(x.device)


def _cpu_conv_grads(x, w, dy, padding, stride=1):
    xr = x.detach().float().cpu().contiguous().requires_grad_(True)
    wr = w.detach().float().cpu().requires_grad_(True)
    F.conv2d(xr, wr, None, padding=padding, stride=stride).backward(dy.detach().float().cpu().contiguous())
    return xr.grad.to(x.device), wr.grad.to(x.device)


CASES = [(2, 64, 64, 20, 24, 1), (1, 128, 192, 17, 19, 0), (3, 64, 128, 33, 9, 1), (1, 256, 64, 130, 7, 0),
         (2, 128, 128, 47, 47, 1)]


def _mk(n, cin, cout, h, w, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(n, cin, h, w, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * 0.05
    b = torch.randn(cout, device="cuda", generator=g)
    return x, wt, b


@pytest.mark.parametrize("n,cin,cout,h,w,pad", CASES)
@pytest.mark.parametrize("w_nhwc", [False, True])
def test_forward_and_dgrad(n, cin, cout, h, w, pad, w_nhwc):
    from sparse2dense_amd import dense2d as D
    x, wt, b = _mk(n, cin, cout, h, w)
    wsrc = wt.contiguous(memory_format=torch.channels_last) if w_nhwc else wt
    wr = wt.to(torch.bfloat16).float()
    y = D.conv3x3_nhwc(x, D.pack_weights(wsrc), b, cin, cout, pad)
    ref = _cpu_conv(x, wr, b, pad)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert (y.float() - ref).abs().max() <= 6e-3 * ref.abs().max()
    dy = torch.randn_like(ref).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dx_ref, _ = _cpu_conv_grads(x, wr, dy, pad)
    src = dy if pad == 1 else F.pad(dy, (1, 1, 1, 1)).contiguous(memory_format=torch.channels_last)
    dx = D.conv3x3_nhwc(src, D.pack_weights(wsrc, True), None, cout, cin, 1)
    assert dx.shape == x.shape
    assert (dx.float() - dx_ref).abs().max() <= 6e-3 * dx_ref.abs().max()


@pytest.mark.parametrize("n,cin,cout,h,w,pad,rows", [(4, 128, 128, 188, 188, 1, 128), (4, 256, 256, 94, 94, 1, 96),
                                                     (1, 128, 128, 188, 188, 1, 64), (4, 128, 256, 95, 93, 0, None)])
def test_bev_sized_launches_cover_every_tile_height(n, cin, cout, h, w, pad, rows):
    """The k32 kernel picks 128-, 96- or 64-pixel tiles per launch (csrc/conv2d_nhwc.hip conv_k32_rows); the BASELINE
    BEV shapes exercise all three.  Checks the output, the tile count the host allocates the statistics slabs for, and the
    epilogue statistics (sum over tiles == column sums of the stored output)."""
    from sparse2dense_amd import _lib, dense2d as D
    x, wt, b = _mk(n, cin, cout, h, w, seed=11)
    y, part = D.conv3x3_nhwc(x, D.pack_weights(wt), b, cin, cout, pad, bn_stats=True)
    ref = _cpu_conv(x, wt.to(torch.bfloat16).float(), b, pad)
    assert (y.float() - ref).abs().max() <= 6e-3 * ref.abs().max()
    m = n * ref.shape[2] * ref.shape[3]
    tiles = _lib.load().s2d_conv2d3x3_stats_tiles(n, h, w, cin, cout, pad, 1)
    assert part.shape == (tiles, 2, cout) and tiles in [-(-m // r) for r in (128, 96, 64)]
    if rows is not None and torch.cuda.get_device_properties(0).multi_processor_count == 256:
        assert tiles == -(-m // rows)
    yf = y.float()
    s1, s2 = yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))
    assert (part[:, 0].sum(0) - s1).abs().max() <= 1e-3 * s1.abs().max() + 1e-2
    assert (part[:, 1].sum(0) - s2).abs().max() <= 1e-3 * s2.abs().max()


def test_unsupported_channels_raise():
    from sparse2dense_amd import _lib, dense2d as D
    w = torch.randn(3, 64, 3, 3, device="cuda")
    with pytest.raises(_lib.S2DError):
        D.pack_weights(w)


@pytest.mark.parametrize("wgrad_hip", [False, True])
@pytest.mark.parametrize("pad,bias", [(1, True), (0, False)])
def test_module_autograd_matches_stock_conv(pad, bias, wgrad_hip, monkeypatch):
    from sparse2dense_amd import dense2d as D
    monkeypatch.setattr(D, "WGRAD_HIP", wgrad_hip)   # MIOpen's and the hand-written weight-gradient kernel
    torch.manual_seed(1)
    m = D.Conv3x3(64, 128, 3, padding=pad, bias=bias).cuda()
    ref = torch.nn.Conv2d(64, 128, 3, padding=pad, bias=bias)   # stays on the HOST: the reference is a CPU convolution
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    x = torch.randn(2, 64, 40, 36, device="cuda")
    dy = torch.randn(2, 128, 40 + 2 * pad - 2, 36 + 2 * pad - 2, device="cuda")
    xa = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ya = m(xa)
    assert ya.dtype == torch.bfloat16
    ya.backward(dy.to(torch.bfloat16))
    # reference in fp32 on bf16-rounded operands, on the CPU
    xb = x.to(torch.bfloat16).float().cpu().requires_grad_(True)
    with torch.no_grad():
        ref.weight.copy_(ref.weight.to(torch.bfloat16).float())
    yr = ref(xb)
    yr.backward(dy.to(torch.bfloat16).float().cpu())

    def close(a, r, tol):
        a, r = a.float().cpu(), r.float().cpu()
        assert (a - r).abs().max() <= tol * r.abs().max(), (a - r).abs().max() / r.abs().max()
    close(ya, yr, 6e-3)
    close(xa.grad, xb.grad, 6e-3)
    close(m.weight.grad, ref.weight.grad, 1e-2)      # MIOpen bf16 wgrad (bf16 output rounding)
    if bias:
        close(m.bias.grad, ref.bias.grad, 1e-3)
    # fp32 (no autocast) and CPU inputs take the stock layer
    y32 = m(x)
    assert y32.dtype == torch.float32


# ---------------------------------------------------------------------------------------------------
# FastBatchNorm2d: row-major bf16 kernels vs torch fp32 batch norm on the same bf16-rounded input.
# Tolerance: outputs/gradients are rounded to bf16 (2^-8 relative) -> 1e-2 of the tensor's max; fp32 per-channel
# quantities (running stats, dgamma, dbeta) 2e-3 relative to their max (bf16 dy/x products, fp32 sums).
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,c,h,w", [(2, 64, 20, 24), (4, 128, 47, 47), (1, 256, 33, 9), (2, 8, 5, 7), (1, 512, 16, 16)])
@pytest.mark.parametrize("relu", [False, True, 2])   # 2 = fused exact GELU (the S2D module's conv-BN-GELU groups)
def test_fast_batchnorm2d_training(n, c, h, w, relu):
    from sparse2dense_amd import dense2d as D
    torch.manual_seed(2)
    m = D.FastBatchNorm2d(c, eps=1e-3, momentum=0.01, fused_relu=relu).cuda()
    ref = torch.nn.BatchNorm2d(c, eps=1e-3, momentum=0.01).double()   # float64 on the CPU: an oracle, not MIOpen
    with torch.no_grad():
        m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.5, 0.5)
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    x = (torch.randn(n, c, h, w, device="cuda") * 2 + 0.3).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(n, c, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xa = x.clone().requires_grad_(True)
    ya = m(xa)
    assert ya.dtype == torch.bfloat16 and ya.is_contiguous(memory_format=torch.channels_last)
    ya.backward(dy)
    xr = x.double().cpu().contiguous().requires_grad_(True)
    yr = ref(xr)
    if relu:
        yr = torch.relu(yr) if relu is True else F.gelu(yr)
    yr.backward(dy.double().cpu().contiguous())

    def close(what, a, r, tol):
        err = (a.double().cpu() - r).abs().max() / r.abs().max().clamp(min=1e-6)
        assert err <= tol, (what, float(err))
    close("y", ya, yr, 1e-2)
    close("dx", xa.grad, xr.grad, 1.5e-2)
    close("dgamma", m.weight.grad, ref.weight.grad, 2e-3)
    close("dbeta", m.bias.grad, ref.bias.grad, 2e-3)
    close("running_mean", m.running_mean, ref.running_mean, 1e-4)
    close("running_var", m.running_var, ref.running_var, 1e-4)
    assert int(m.num_batches_tracked) == 1


def test_fast_batchnorm2d_eval_and_fallbacks():
    from sparse2dense_amd import dense2d as D
    torch.manual_seed(3)
    m = D.FastBatchNorm2d(64, fused_relu=True).cuda()
    with torch.no_grad():
        m.running_mean.uniform_(-1, 1); m.running_var.uniform_(0.5, 2)
    m.eval()
    x = torch.randn(2, 64, 12, 10, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xa = x.clone().requires_grad_(True)
    y = m(xa)
    y.backward(torch.ones_like(y))
    xr = x.float().contiguous().requires_grad_(True)
    yr = torch.relu(F.batch_norm(xr, m.running_mean, m.running_var, m.weight, m.bias, False, 0.0, m.eps))
    yr.backward(torch.ones_like(yr))
    assert (y.float() - yr).abs().max() <= 1e-2 * yr.abs().max()
    assert (xa.grad.float() - xr.grad).abs().max() <= 1e-2 * xr.grad.abs().max()
    # fp32 / NCHW inputs: stock path, same fused-ReLU semantics
    x32 = torch.randn(2, 64, 12, 10, device="cuda")
    assert (m(x32) - torch.relu(F.batch_norm(x32, m.running_mean, m.running_var, m.weight, m.bias, False, 0.0, m.eps))).abs().max() < 1e-5


def test_densify_bev_nhwc_bf16_matches_dense_view():
    """dense_bev(nhwc_bf16=True) == dense().view(N, C*D, H, W) rounded to bf16 (bit-exact), gradient = the gather."""
    from sparse2dense_amd.spconv import SparseConvTensor
    g = torch.Generator().manual_seed(5)
    batch, d, h, w, c = 3, 2, 23, 17, 128
    cells = torch.randperm(batch * d * h * w, generator=g)[:700]
    coors = torch.stack([cells // (d * h * w), (cells // (h * w)) % d, (cells // w) % h, cells % w], 1).int().cuda()
    feat = torch.randn(700, c, generator=g).cuda()
    fa = feat.clone().requires_grad_(True)
    fb = feat.clone().requires_grad_(True)
    a = SparseConvTensor(fa, coors, [d, h, w], batch).dense_bev(nhwc_bf16=True)
    b = SparseConvTensor(fb, coors, [d, h, w], batch).dense_bev(nhwc_bf16=False)
    assert a.shape == b.shape and a.dtype == torch.bfloat16 and a.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(a.float(), b.to(torch.bfloat16).float())
    dy = torch.randn(b.shape, device="cuda").to(torch.bfloat16)
    a.backward(dy.contiguous(memory_format=torch.channels_last))
    b.backward(dy.float())
    assert torch.equal(fa.grad, fb.grad)


@pytest.mark.parametrize("route", ["torch", "direct"])
def test_bn_rows_sync_path_equals_local_path_on_one_rank(monkeypatch, route):
    """The SyncBN route (split statistics kernels -> all-reduce of [sum, sumsq, count] -> finalize; backward likewise)
    must reproduce the fused single-GPU route when the world has one rank (RCCL all-reduce of one contribution), both
    through torch.distributed and through the communicator of csrc/comm.hip on the compute stream."""
    import torch.distributed as dist
    from sparse2dense_amd import _lib, collective, dense2d as D
    from sparse2dense_amd.spconv import FeatureBatchNorm1d
    if dist.is_initialized():
        pytest.skip("a process group is already up")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{29577 + (route == 'direct')}", rank=0, world_size=1)
    try:
        if route == "direct":
            assert collective.init_direct(torch.cuda.current_device()) and collective.direct_enabled()
            t = torch.arange(5, dtype=torch.float32, device="cuda")
            collective.allreduce_sum_(t)
            assert t.tolist() == [0.0, 1.0, 2.0, 3.0, 4.0] and _lib.load().s2d_comm_ranks() == 1
        torch.manual_seed(4)
        for make, shape in [(lambda: D.FastBatchNorm2d(64, eps=1e-3, momentum=0.01, fused_relu=True), (2, 64, 30, 26)),
                            (lambda: FeatureBatchNorm1d(32, eps=1e-3, momentum=0.01), (3000, 32))]:
            a, b = make().cuda(), make().cuda()
            b.load_state_dict(a.state_dict())
            x = torch.randn(*shape, device="cuda").to(torch.bfloat16)
            if len(shape) == 4:
                x = x.contiguous(memory_format=torch.channels_last)
            dy = torch.randn_like(x)
            outs = []
            for m, force in ((a, "0"), (b, "1")):
                monkeypatch.setenv("S2D_FORCE_DDP", force)
                xi = x.clone().requires_grad_(True)
                y = m(xi)
                y.backward(dy)
                outs.append((y, xi.grad, m.weight.grad, m.bias.grad, m.running_mean, m.running_var, m.num_batches_tracked))
            for u, v in zip(*outs):
                assert torch.allclose(u.float(), v.float(), rtol=1e-5, atol=1e-6), (u.float() - v.float()).abs().max()
    finally:
        collective._DIRECT = False
        _lib.load().s2d_comm_shutdown()
        dist.destroy_process_group()


@pytest.mark.parametrize("n,cin,cout,h,w,pad", [(2, 64, 64, 20, 24, 1), (1, 128, 192, 17, 19, 0), (3, 64, 128, 33, 9, 1),
                                                (1, 256, 64, 70, 37, 0), (2, 128, 128, 47, 47, 1), (1, 128, 128, 5, 100, 1)])
def test_conv3x3_wgrad_matches_float64(n, cin, cout, h, w, pad):
    """dW of the dense 3x3 conv (LDS transpose-read MFMA kernel) vs a float64 autograd reference on the same bf16
    operands; fp32 accumulation over up to ~1e4 pixels -> 1e-3 of max|dW| (measured ~1e-5)."""
    from sparse2dense_amd import dense2d as D
    x, wt, _ = _mk(n, cin, cout, h, w, seed=3)
    ho, wo = h + 2 * pad - 2, w + 2 * pad - 2
    dy = torch.randn(n, cout, ho, wo, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dw = D.conv3x3_wgrad(x, dy, pad)
    wr = wt.double().cpu().requires_grad_(True)   # float64 on the host
    F.conv2d(x.double().cpu().contiguous(), wr, None, padding=pad).backward(dy.double().cpu().contiguous())
    assert dw.shape == wr.grad.shape and dw.dtype == torch.float32
    err = (dw.double().cpu() - wr.grad).abs().max() / wr.grad.abs().max()
    assert err <= 1e-3, float(err)


@pytest.mark.parametrize("pad", [0, 1])
def test_stride2_forward_and_module_backward(pad):
    """stride-2 3x3 conv (first layer of the second RPN block, rpn.py:126-131): forward on the HIP kernel, gradients via MIOpen."""
    from sparse2dense_amd import dense2d as D
    x, wt, b = _mk(2, 128, 256, 41, 38, seed=5)
    y = D.conv3x3_nhwc(x, D.pack_weights(wt), b, 128, 256, pad, stride=2)
    ref = _cpu_conv(x, wt.to(torch.bfloat16).float(), b, pad, stride=2)
    assert y.shape == ref.shape
    assert (y.float() - ref).abs().max() <= 6e-3 * ref.abs().max()
    m = D.Conv3x3(128, 256, 3, stride=2, padding=pad, bias=False).cuda()
    xa = x.float().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ya = m(xa)
    ya.float().sum().backward()
    xr = x.float().cpu().contiguous().requires_grad_(True)
    yr = F.conv2d(xr, m.weight.detach().to(torch.bfloat16).float().cpu(), None, padding=pad, stride=2)
    yr.sum().backward()
    assert (ya.float().cpu() - yr).abs().max() <= 6e-3 * yr.abs().max()
    assert (xa.grad.cpu() - xr.grad).abs().max() <= 1e-2 * xr.grad.abs().max()


@pytest.mark.parametrize("cin,cout,pad,stride", [(64, 128, 1, 1), (128, 64, 0, 1), (128, 256, 0, 2), (512, 64, 1, 1)])
def test_conv_epilogue_bn_statistics_match_the_reduction_kernel(cin, cout, pad, stride):
    """Conv3x3 -> FastBatchNorm2d(+ReLU): batch statistics produced in the conv epilogue (per-tile partial sums) must give
    the same normalisation as the stand-alone statistics pass over the stored bf16 tensor (different fp32 summation order)."""
    import copy
    from sparse2dense_amd import dense2d as D
    torch.manual_seed(9)
    fused = torch.nn.Sequential(*D.fuse_bn_relu([D.Conv3x3(cin, cout, 3, stride=stride, padding=pad, bias=False),
                                                 D.FastBatchNorm2d(cout, eps=1e-3, momentum=0.01), torch.nn.ReLU()])).cuda().train()
    assert fused[0].emit_bn_stats and fused[1].fused_relu
    plain = copy.deepcopy(fused)
    plain[0].emit_bn_stats = False
    x = torch.randn(3, cin, 37, 29, device="cuda")
    outs = []
    for m in (fused, plain):
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(xi)
        y.float().square().sum().backward()
        outs.append((y, xi.grad, m[0].weight.grad, m[1].weight.grad, m[1].bias.grad, m[1].running_mean, m[1].running_var))
    for name, u, v in zip(["y", "dx", "dw", "dgamma", "dbeta", "rmean", "rvar"], *outs):
        err = (u.float() - v.float()).abs().max() / v.float().abs().max().clamp(min=1e-9)
        assert err <= (1e-2 if name in ("y", "dx", "dw") else 1e-4), (name, float(err))   # bf16 tensors may flip one rounding


@pytest.mark.parametrize("kind,c0,c1,c2,act,hw", [("3x3", 64, 128, 128, "relu", (37, 29)), ("3x3", 128, 128, 64, "gelu", (21, 40)),
                                                  ("1x1", 128, 256, 128, "gelu", (33, 20)), ("1x1", 256, 128, 256, "relu", (188, 188)),
                                                  ("3x3", 128, 128, 128, "relu", (188, 188)), ("3x3pad0", 64, 64, 64, "relu", (30, 31))])
def test_batch_norm_backward_sums_from_the_data_gradient_epilogue(kind, c0, c1, c2, act, hw, monkeypatch):
    """conv -> BN -> act -> conv -> BN -> act: the second conv's data gradient IS the first batch norm's dY and writes that layer's backward
    sums (sum g, sum g z per tile) from its epilogue (csrc/conv2d_nhwc.hip `BnBwd`); the batch norm then folds the slabs instead of reading
    (dY, z) again.  Same sums in a different fp32 order: per-channel results within 1e-4 of the stand-alone reduction pass, bf16 tensors within
    one rounding; and the float64 CPU chain on the same bf16-rounded weights bounds both.  A batch-norm output with TWO consumers (autograd
    sums the two gradients) must not take the slabs of either."""
    import copy
    from sparse2dense_amd import dense2d as D

    def conv(ci, co):
        if kind == "1x1":
            return D.Conv1x1(ci, co, 1, bias=False)
        return D.Conv3x3(ci, co, 3, padding=0 if kind == "3x3pad0" else 1, bias=False)
    torch.manual_seed(11)
    A = torch.nn.GELU if act == "gelu" else torch.nn.ReLU
    net = torch.nn.Sequential(*D.fuse_bn_relu([conv(c0, c1), D.FastBatchNorm2d(c1, eps=1e-3, momentum=0.01), A(),
                                               conv(c1, c2), D.FastBatchNorm2d(c2, eps=1e-3, momentum=0.01), A()])).cuda().train()
    with torch.no_grad():
        for m in net:
            if isinstance(m, D.FastBatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.5, 0.5)
    n = 4 if hw == (188, 188) else 3
    x = torch.randn(n, c0, *hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    outs = {}
    for fold in (True, False):
        monkeypatch.setattr(D, "BN_BWD_FOLD", 2 if fold else 0)   # 2: also behind a GELU (default 1: ReLU / none only)
        m = copy.deepcopy(net)
        for k in D.STATS:
            D.STATS[k] = 0
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(xi)
        (y.float() * torch.linspace(-1, 1, y.shape[-1], device="cuda")).sum().backward()
        assert D.STATS["bn_bwd_folded"] == (1 if fold else 0) and D.STATS["bn_bwd_reduced"] == (1 if fold else 2), (fold, D.STATS)
        outs[fold] = dict(y=y, dx=xi.grad, dw0=m[0].weight.grad, dw1=m[3].weight.grad, dgamma0=m[1].weight.grad, dbeta0=m[1].bias.grad,
                          dgamma1=m[4].weight.grad, dbeta1=m[4].bias.grad)
    for name, u in outs[True].items():
        v = outs[False][name]
        err = (u.float() - v.float()).abs().max() / v.float().abs().max().clamp(min=1e-9)
        assert err <= (1e-2 if name in ("dx", "dw0") else 1e-4), (name, float(err))   # (dx / dw0 sit behind bf16 dz: one rounding may flip)
    if hw != (188, 188):   # float64 on the CPU, bf16-rounded weights: the documented bf16-storage tolerance for both routes
        ref = torch.nn.Sequential(*[torch.nn.Conv2d(l.in_channels, l.out_channels, l.kernel_size, padding=l.padding, bias=False)
                                    if isinstance(l, torch.nn.Conv2d) else torch.nn.BatchNorm2d(l.num_features, eps=1e-3, momentum=0.01)
                                    if isinstance(l, D.FastBatchNorm2d) else torch.nn.Identity() for l in net]).double()
        with torch.no_grad():
            for l, r in zip(net, ref):
                if isinstance(l, torch.nn.Conv2d):
                    r.weight.copy_(l.weight.to(torch.bfloat16).double().cpu())
                elif isinstance(l, D.FastBatchNorm2d):
                    r.weight.copy_(l.weight.double().cpu()); r.bias.copy_(l.bias.double().cpu())
        f = torch.nn.functional.gelu if act == "gelu" else torch.relu
        xr = x.double().cpu().contiguous().requires_grad_(True)
        yr = f(ref[4](ref[3](f(ref[1](ref[0](xr))))))
        (yr * torch.linspace(-1, 1, yr.shape[-1], dtype=torch.float64)).sum().backward()
        for name, r in dict(dgamma0=ref[1].weight.grad, dbeta0=ref[1].bias.grad, dx=xr.grad, dw0=ref[0].weight.grad).items():
            for fold in (True, False):
                u = outs[fold][name].double().cpu()
                # (a ReLU decision flipped by the bf16 rounding of z moves single elements of dx by more: norm-wise there)
                err = (u - r).norm() / r.norm() if name in ("dx", "dw0") else (u - r).abs().max() / r.abs().max()
                assert err <= 5e-2, (name, fold, float(err))
    # two consumers of the first batch norm's output: neither data gradient may stand in for the sum
    monkeypatch.setattr(D, "BN_BWD_FOLD", 2)
    m = copy.deepcopy(net)
    second = copy.deepcopy(net[3])
    for k in D.STATS:
        D.STATS[k] = 0
    xi = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        h = m[2](m[1](m[0](xi)))
        y = m[3](h).float().sum() + second(h).float().square().sum()
    y.backward()
    assert D.STATS["bn_bwd_folded"] == 0 and D.STATS["bn_bwd_reduced"] == 1, D.STATS
    monkeypatch.setattr(D, "BN_BWD_FOLD", 0)
    m2 = copy.deepcopy(net)
    xj = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        h = m2[2](m2[1](m2[0](xj)))
        y2 = m2[3](h).float().sum() + second(h).float().square().sum()
    y2.backward()
    assert torch.equal(xi.grad, xj.grad) and torch.equal(m[1].weight.grad, m2[1].weight.grad)


@pytest.mark.parametrize("n,c,h,w", [(4, 256, 94, 94), (2, 64, 5, 7), (1, 8, 1, 1), (3, 136, 9, 4)])
def test_space_to_depth_kernel_equals_the_permuted_view(n, c, h, w, monkeypatch):
    """csrc/layout.hip `space_depth2_kernel` (the rearrangement around the 2x2 / stride-2 conv and the 2x2 transposed conv run as 1x1 tile kernels):
    a pure permutation - bit-equal to torch's permute + reshape of the same tensor, both directions, and the two are inverses."""
    from sparse2dense_amd import dense2d as D
    torch.manual_seed(n + c + h)
    x = torch.randn(n, c, 2 * h, 2 * w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    got = D._space_to_depth(x)
    monkeypatch.setattr(D, "ENABLED", False)
    want = D._space_to_depth(x)
    monkeypatch.setattr(D, "ENABLED", True)
    assert got.shape == want.shape == (n, 4 * c, h, w) and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, want)
    back = D._depth_to_space(got, c)
    assert back.is_contiguous(memory_format=torch.channels_last) and torch.equal(back, x)
    monkeypatch.setattr(D, "ENABLED", False)
    assert torch.equal(D._depth_to_space(got, c), x)


def test_wide_layernorm_matches_stock():
    from sparse2dense_amd import dense2d as D
    torch.manual_seed(0)
    m = D.WideLayerNorm([64, 47, 47], eps=1e-6).cuda()
    ref = torch.nn.LayerNorm([64, 47, 47], eps=1e-6).cuda()
    with torch.no_grad():
        m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-1, 1)
    ref.load_state_dict(m.state_dict())
    x = torch.randn(2, 64, 47, 47, device="cuda") * 3 + 1
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = m(xa), ref(xb)
    g = torch.randn_like(x)
    ya.backward(g); yb.backward(g)
    assert torch.allclose(ya, yb, rtol=1e-4, atol=1e-5)
    assert torch.allclose(xa.grad, xb.grad, rtol=1e-3, atol=1e-5)
    assert torch.allclose(m.weight.grad, ref.weight.grad, rtol=1e-4, atol=1e-5) and torch.allclose(m.bias.grad, ref.bias.grad, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("n,c,h,w,nhwc", [(4, 256, 47, 47, True), (2, 64, 47, 47, False), (3, 8, 96, 100, True), (1, 256, 47, 47, True)])
def test_wide_layernorm_bf16_kernels_vs_float64(n, c, h, w, nhwc):
    """csrc/layernorm.hip on bf16 maps (NHWC and NCHW memory order) vs nn.LayerNorm in float64 on the HOST over the same bf16
    input / gradient.  bf16 outputs (y, dx): one rounding = 6e-3 of max; fp32 dW / db: 1e-4 of max."""
    from sparse2dense_amd import dense2d as D
    torch.manual_seed(7)
    m = D.WideLayerNorm([c, h, w], eps=1e-6).cuda()
    with torch.no_grad():
        m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-1, 1)
    ref = torch.nn.LayerNorm([c, h, w], eps=1e-6).double()
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    fmt = torch.channels_last if nhwc else torch.contiguous_format
    x = (torch.randn(n, c, h, w, device="cuda") * 2 + 0.5).to(torch.bfloat16).contiguous(memory_format=fmt)
    dy = torch.randn(n, c, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=fmt)
    xa = x.clone(memory_format=torch.preserve_format).requires_grad_(True)
    ya = m(xa)
    assert ya.dtype == torch.bfloat16 and ya.stride() == x.stride()
    ya.backward(dy)
    xr = x.double().cpu().contiguous().requires_grad_(True)
    yr = ref(xr)
    yr.backward(dy.double().cpu().contiguous())
    for name, a, r, tol in (("y", ya, yr, 6e-3), ("dx", xa.grad, xr.grad, 6e-3), ("dw", m.weight.grad, ref.weight.grad, 1e-4),
                            ("db", m.bias.grad, ref.bias.grad, 1e-4)):
        err = float((a.detach().double().cpu() - r.detach()).abs().max() / r.detach().abs().max())
        assert err <= tol, (name, err)
    # a second call after an in-place parameter update sees the new weights (packed copies are keyed on the version counter)
    with torch.no_grad():
        m.weight.mul_(2.0)
    y2 = m(x)
    assert float((y2.float() - (ya.float() - m.bias.view(1, c, h, w)) * 2 - m.bias.view(1, c, h, w)).abs().max()) <= 0.1


# ---------------------------------------------------------------------------------------------------
# Depth-wise 7x7 (csrc/dwconv.hip) vs torch's grouped conv in float64 on the HOST, same bf16-rounded operands.
# Tolerance: bf16 outputs (y, dx) one output rounding = 6e-3 of max; fp32 outputs (dW, db) 2e-3 of max.
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,c,h,w", [(4, 256, 47, 47), (1, 64, 9, 5), (2, 8, 20, 33), (1, 512, 59, 59)])
@pytest.mark.parametrize("bias", [True, False])
def test_depthwise7_forward_backward_vs_cpu_float64(n, c, h, w, bias):
    from sparse2dense_amd import dense2d as D
    torch.manual_seed(4)
    m = D.DepthwiseConv7(c, c, 7, padding=3, groups=c, bias=bias).cuda()
    x = torch.randn(n, c, h, w, device="cuda").to(torch.bfloat16).float()
    dy = torch.randn(n, c, h, w, device="cuda").to(torch.bfloat16).float()
    xa = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert m._hip_ok(xa)
        ya = m(xa)
    assert ya.dtype == torch.bfloat16 and ya.is_contiguous(memory_format=torch.channels_last)
    ya.backward(dy.to(torch.bfloat16))
    ref = torch.nn.Conv2d(c, c, 7, padding=3, groups=c, bias=bias).double()
    ref.load_state_dict({k: v.cpu().double() for k, v in m.state_dict().items()})
    xr = x.cpu().double().requires_grad_(True)
    yr = ref(xr)
    yr.backward(dy.cpu().double())

    def close(a, r, tol):
        a, r = a.double().cpu(), r.double()
        assert (a - r).abs().max() <= tol * r.abs().max(), float((a - r).abs().max() / r.abs().max())
    close(ya, yr, 6e-3)
    close(xa.grad, xr.grad, 6e-3)
    close(m.weight.grad, ref.weight.grad, 2e-3)
    if bias:
        close(m.bias.grad, ref.bias.grad, 2e-3)
    assert m(x).dtype == torch.float32   # no autocast: the stock layer


@pytest.mark.parametrize("n,c,h,w", [(2, 640, 47, 48), (4, 64, 10, 6), (1, 8, 3, 4), (2, 72, 33, 20), (3, 136, 5, 52)])
def test_planar_handover_transposes_bit_exact(n, c, h, w):
    """NHWC bf16 -> NCHW fp32 and its gradient path (csrc/layout.hip) are pure data movement: bit-exact vs torch's copies"""
    from sparse2dense_amd.necks import _ToPlanarF32
    torch.manual_seed(1)
    x = torch.randn(n, c, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = _ToPlanarF32.apply(x)
    assert y.dtype == torch.float32 and y.is_contiguous() and torch.equal(y, x.detach().float().contiguous())
    g = torch.randn(n, c, h, w, device="cuda")
    y.backward(g)
    assert x.grad.dtype == torch.bfloat16 and x.grad.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(x.grad.float(), g.to(torch.bfloat16).float())


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 256, 1024, 47, 47), (2, 1024, 256, 20, 12), (2, 256, 640, 33, 20), (1, 64, 64, 5, 7),
                                            (4, 256, 256, 94, 94), (1, 128, 192, 9, 130)])
@pytest.mark.parametrize("bias", [True, False])
def test_conv1x1_forward_backward_vs_cpu_float64(n, cin, cout, h, w, bias):
    """Conv1x1 (the 3x3 tile kernels with one tap) vs torch conv2d in float64 on the HOST over the same bf16-rounded operands.
    bf16 outputs (y, dx): one rounding = 6e-3 of max; fp32 outputs (dW, db): 2e-3 of max; epilogue statistics == sums of the
    stored outputs."""
    from sparse2dense_amd import dense2d as D
    torch.manual_seed(n + cin + cout)
    m = D.Conv1x1(cin, cout, 1, 1, 0, bias=bias).cuda()
    rb = lambda t: t.to(torch.bfloat16).float()
    with torch.no_grad():
        m.weight.copy_(rb(m.weight))
    x = rb(torch.randn(n, cin, h, w, device="cuda"))
    dy = rb(torch.randn(n, cout, h, w, device="cuda"))
    xa = x.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ya = m(xa)
    assert ya.dtype == torch.bfloat16 and ya.is_contiguous(memory_format=torch.channels_last)
    ya.backward(dy.to(torch.bfloat16))
    ref = torch.nn.Conv2d(cin, cout, 1, bias=bias).double()
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    xr = x.double().cpu().requires_grad_(True)
    yr = ref(xr)
    yr.backward(dy.double().cpu())
    checks = [("y", ya, yr, 6e-3), ("dx", xa.grad, xr.grad, 6e-3), ("dw", m.weight.grad, ref.weight.grad, 2e-3)]
    if bias:
        checks.append(("db", m.bias.grad, ref.bias.grad, 2e-3))
    for name, a, r, tol in checks:
        err = float((a.detach().double().cpu() - r.detach()).abs().max() / r.detach().abs().max())
        assert err <= tol, (name, err)
    # batch-norm statistics from the epilogue
    m.emit_bn_stats = True
    m.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y2 = m(x.contiguous(memory_format=torch.channels_last).requires_grad_(True))
    part = y2._s2d_bn_partial
    yf = y2.detach().float()
    s1, s2 = yf.sum(dim=(0, 2, 3)), (yf * yf).sum(dim=(0, 2, 3))
    assert (part[:, 0].sum(0) - s1).abs().max() <= 1e-3 * s1.abs().max() + 1e-2
    assert (part[:, 1].sum(0) - s2).abs().max() <= 1e-3 * s2.abs().max()


@pytest.mark.parametrize("n,cin,cout,h,w", [(4, 256, 256, 188, 188), (2, 64, 128, 10, 6), (1, 128, 64, 34, 18)])
def test_conv2x2_stride2_forward_vs_cpu_float64(n, cin, cout, h, w):
    """Conv2x2S2 (encoder_1[0] of the S2D module) forward on the tile kernel with 4 taps vs float64 on the host over bf16-rounded
    operands (one output rounding: 6e-3 of max) + its epilogue statistics; the backward (1x1 kernels over the space-to-depth image): dx 6e-3
    (bf16), dW / db 2e-3 (fp32 sums)."""
    from sparse2dense_amd import dense2d as D
    torch.manual_seed(cin + cout + h)
    m = D.Conv2x2S2(cin, cout, 2, 2).cuda()
    rb = lambda t: t.to(torch.bfloat16).float()
    with torch.no_grad():
        m.weight.copy_(rb(m.weight))
    x = rb(torch.randn(n, cin, h, w, device="cuda"))
    dy = rb(torch.randn(n, cout, h // 2, w // 2, device="cuda"))
    xa = x.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    m.emit_bn_stats = True
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ya = m(xa)
    part = ya._s2d_bn_partial
    assert ya.dtype == torch.bfloat16 and ya.is_contiguous(memory_format=torch.channels_last)
    ya.backward(dy.to(torch.bfloat16))
    ref = torch.nn.Conv2d(cin, cout, 2, 2).double()
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    xr = x.double().cpu().requires_grad_(True)
    yr = ref(xr)
    yr.backward(dy.double().cpu())
    for name, a, r, tol in (("y", ya, yr, 6e-3), ("dx", xa.grad, xr.grad, 6e-3), ("dw", m.weight.grad, ref.weight.grad, 2e-3),
                            ("db", m.bias.grad, ref.bias.grad, 2e-3)):
        err = float((a.detach().double().cpu() - r.detach()).abs().max() / r.detach().abs().max())
        assert err <= tol, (name, err)
    yf = ya.detach().float()
    s1, s2 = yf.sum(dim=(0, 2, 3)), (yf * yf).sum(dim=(0, 2, 3))
    assert (part[:, 0].sum(0) - s1).abs().max() <= 1e-3 * s1.abs().max() + 1e-2
    assert (part[:, 1].sum(0) - s2).abs().max() <= 1e-3 * s2.abs().max()


@pytest.mark.parametrize("n,cin,cout,h,w", [(4, 64, 3, 188, 188), (2, 64, 1, 33, 20), (1, 64, 2, 7, 5), (2, 128, 4, 19, 30), (3, 8, 3, 1, 9)])
@pytest.mark.parametrize("bias", [True, False])
def test_small_cout_conv3x3_forward_backward_vs_cpu_float64(n, cin, cout, h, w, bias):
    """SmallConv3x3 (the last conv of every CenterHead branch; streaming kernels of csrc/smallconv.hip) vs torch conv2d in float64 on
    the HOST over the same bf16-rounded input.  Weights, products and sums are fp32, the predictions fp32: y, dW, db within 1e-5 of
    max; dx is stored as bf16 (one rounding: 4e-3 of max).  Two runs are bit-identical (fixed-order folds)."""
    from sparse2dense_amd import dense2d as D
    torch.manual_seed(n + cin + cout + h)
    m = D.SmallConv3x3(cin, cout, 3, 1, 1, bias=bias).cuda()
    x = torch.randn(n, cin, h, w, device="cuda").to(torch.bfloat16).float()
    dy = torch.randn(n, cout, h, w, device="cuda")
    outs = []
    for _ in range(2):
        m.zero_grad()
        xa = x.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ya = m(xa)
        assert ya.dtype == torch.float32 and ya.is_contiguous()
        ya.backward(dy)
        outs.append((ya.detach().clone(), xa.grad.clone(), m.weight.grad.clone(), None if not bias else m.bias.grad.clone()))
    for a, b in zip(*outs):
        assert a is None or torch.equal(a, b)
    ref = torch.nn.Conv2d(cin, cout, 3, padding=1, bias=bias).double()
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    xr = x.double().cpu().requires_grad_(True)
    yr = ref(xr)
    yr.backward(dy.double().cpu())
    checks = [("y", ya, yr, 1e-5), ("dx", xa.grad, xr.grad, 4e-3), ("dw", m.weight.grad, ref.weight.grad, 1e-5)]
    if bias:
        checks.append(("db", m.bias.grad, ref.bias.grad, 1e-5))
    for name, a, r, tol in checks:
        err = float((a.detach().double().cpu() - r.detach()).abs().max() / r.detach().abs().max())
        assert err <= tol, (name, err)


def test_center_head_branches_end_in_the_streaming_conv():
    from sparse2dense_amd import dense2d as D
    from sparse2dense_amd.heads import SepHead
    head = SepHead(64, dict(reg=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), hm=(3, 2)), bn=True, final_kernel=3)
    for name in head.heads:
        assert isinstance(getattr(head, name)[-1], D.SmallConv3x3), name
    # same parameter names as the reference's nn.Sequential(Conv2d, BN, ReLU, Conv2d)
    assert {k for k in head.state_dict() if k.startswith("hm.")} >= {"hm.0.weight", "hm.0.bias", "hm.1.weight", "hm.3.weight", "hm.3.bias"}


@pytest.mark.parametrize("n,cin,cout,h,w", [(4, 256, 256, 94, 94), (2, 64, 128, 5, 7), (1, 128, 64, 33, 20)])
@pytest.mark.parametrize("bias", [False, True])
def test_conv_transpose_2x2_stride2_vs_cpu_float64(n, cin, cout, h, w, bias):
    """ConvT2x2S2 (the RPN's up-sampling deblock: a 1x1 conv to 4x the channels on the tile kernels + depth-to-space; backward =
    space-to-depth + 1x1 data / weight gradient) vs torch ConvTranspose2d in float64 on the HOST over the same bf16-rounded operands.
    bf16 outputs (y, dx): 6e-3 of max; fp32 outputs (dW, db): 2e-3 of max."""
    from sparse2dense_amd import dense2d as D
    torch.manual_seed(n + cin + cout + h)
    m = D.ConvT2x2S2(cin, cout, 2, stride=2, bias=bias).cuda().to(memory_format=torch.channels_last)
    rb = lambda t: t.to(torch.bfloat16).float()
    with torch.no_grad():
        m.weight.copy_(rb(m.weight))
    x = rb(torch.randn(n, cin, h, w, device="cuda"))
    dy = rb(torch.randn(n, cout, 2 * h, 2 * w, device="cuda"))
    xa = x.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ya = m(xa)
    assert ya.dtype == torch.bfloat16 and ya.shape == dy.shape and ya.is_contiguous(memory_format=torch.channels_last)
    ya.backward(dy.to(torch.bfloat16))
    ref = torch.nn.ConvTranspose2d(cin, cout, 2, stride=2, bias=bias).double()
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    xr = x.double().cpu().requires_grad_(True)
    yr = ref(xr)
    yr.backward(dy.double().cpu())
    checks = [("y", ya, yr, 6e-3), ("dx", xa.grad, xr.grad, 6e-3), ("dw", m.weight.grad, ref.weight.grad, 2e-3)]
    if bias:
        checks.append(("db", m.bias.grad, ref.bias.grad, 2e-3))
    for name, a, r, tol in checks:
        err = float((a.detach().double().cpu() - r.detach()).abs().max() / r.detach().abs().max())
        assert err <= tol, (name, err)
    assert set(m.state_dict()) == set(ref.state_dict())


@pytest.mark.parametrize("n,cin,cout,h,w,pad", [(4, 128, 128, 188, 188, 1), (4, 256, 256, 94, 94, 1), (4, 64, 64, 188, 188, 1)])
def test_conv3x3_wgrad_at_bev_size(n, cin, cout, h, w, pad):
    """the weight gradient at the benchmark's BEV maps (1 105 tiles, split-K over the whole map) against a float64 contraction on the device:
    per tap dW[:, :, ky, kx] = dy^T x_shifted (the small-shape test above stops at 47 x 47)"""
    from sparse2dense_amd import dense2d as D
    x, wt, _ = _mk(n, cin, cout, h, w, seed=9)
    dy = torch.randn(n, cout, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dw, db = D.conv3x3_wgrad(x, dy, pad, want_db=True)
    xp = F.pad(x.float(), (1, 1, 1, 1)).permute(0, 2, 3, 1)            # [n, h+2, w+2, cin]
    dyr = dy.float().permute(0, 2, 3, 1).reshape(-1, cout).double()   # [n*h*w, cout]
    ref = torch.empty(cout, cin, 3, 3, device="cuda", dtype=torch.float64)
    for ky in range(3):
        for kx in range(3):
            ref[:, :, ky, kx] = dyr.t() @ xp[:, ky:ky + h, kx:kx + w, :].reshape(-1, cin).double()
    assert torch.isfinite(dw).all()
    assert (dw.double() - ref).abs().max() <= 1e-3 * ref.abs().max()
    assert (db.double() - dyr.sum(0)).abs().max() <= 1e-3 * dyr.sum(0).abs().max()
