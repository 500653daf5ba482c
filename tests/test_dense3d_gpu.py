"""PCR-head streaming kernels (csrc/dense3d.hip) against torch's own Conv3d / ConvTranspose3d on the
CPU in float64 (the reference uses exactly these nn layers, necks/rpn.py:263-296)."""
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu

from sparse2dense_amd.dense3d import ConvTranspose3dK4S2, PointwiseConv3d

DEV = "cuda:0"


def _compare(layer_hip, layer_ref, x, rtol=1e-4, atol=1e-4):
    layer_ref.load_state_dict(layer_hip.state_dict())
    layer_ref = layer_ref.double()
    xr = x.double().requires_grad_(True)
    yr = layer_ref(xr)
    g = torch.randn(yr.shape, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    gr = torch.autograd.grad(yr, [xr] + list(layer_ref.parameters()), g)
    layer_hip = layer_hip.to(DEV)
    xh = x.to(DEV).requires_grad_(True)
    yh = layer_hip(xh)
    gh = torch.autograd.grad(yh, [xh] + list(layer_hip.parameters()), g.float().to(DEV))
    torch.testing.assert_close(yh.cpu().double(), yr, rtol=rtol, atol=atol)
    for a, b in zip(gh, gr):
        scale = b.abs().max().item() + 1e-12
        assert (a.cpu().double() - b).abs().max().item() <= 2e-4 * scale + atol, (a.shape, scale)


@pytest.mark.parametrize("cin,cout,shape", [(128, 32, (2, 5, 12, 16)), (32, 3, (1, 4, 10, 12)), (32, 1, (2, 3, 8, 8)),
                                            (32, 16, (2, 3, 8, 12)), (3, 3, (1, 6, 10, 8)), (3, 1, (2, 2, 6, 4)),
                                            (16, 8, (1, 1, 20, 20)), (64, 32, (2, 1, 12, 12)), (5, 7, (1, 3, 5, 7))])
def test_pointwise_conv3d(cin, cout, shape):
    torch.manual_seed(cin + cout)
    n, d, h, w = shape
    x = torch.randn(n, cin, d, h, w)
    _compare(PointwiseConv3d(cin, cout, 1, 1, 0), nn.Conv3d(cin, cout, 1, 1, 0), x)


@pytest.mark.parametrize("cin,cout,shape", [(32, 32, (2, 3, 6, 8)), (16, 3, (1, 4, 7, 9)), (4, 5, (2, 2, 5, 6)), (1, 1, (1, 1, 1, 2))])
def test_convtranspose3d_k4s2p1(cin, cout, shape):
    torch.manual_seed(cin * 3 + cout)
    n, d, h, w = shape
    x = torch.randn(n, cin, d, h, w)
    _compare(ConvTranspose3dK4S2(cin, cout, 4, 2, 1), nn.ConvTranspose3d(cin, cout, 4, 2, 1), x)


def test_full_size_pcr_shapes_run():
    """rpn.py:317-323 at the real extents: [1,128,5,188,188] -> [1,32,10,376,376] -> [1,3,20,752,752]"""
    torch.manual_seed(0)
    g1 = nn.Sequential(PointwiseConv3d(128, 32, 1, 1, 0), nn.ReLU(), ConvTranspose3dK4S2(32, 32, 4, 2, 1)).to(DEV)
    g2 = nn.Sequential(PointwiseConv3d(32, 16, 1, 1, 0), nn.ReLU(), ConvTranspose3dK4S2(16, 3, 4, 2, 1)).to(DEV)
    x = torch.randn(1, 128, 5, 188, 188, device=DEV, requires_grad=True)
    y1 = g1(x)
    y2 = g2(y1)
    assert y1.shape == (1, 32, 10, 376, 376) and y2.shape == (1, 3, 20, 752, 752)
    y2.mean().backward()
    assert torch.isfinite(x.grad).all() and g1[2].weight.grad.abs().sum() > 0
    # spot-check a strided sub-volume against torch on the CPU
    ref = nn.ConvTranspose3d(16, 3, 4, 2, 1)
    ref.load_state_dict(g2[2].state_dict())
    z = g2[1](g2[0](y1)).detach()
    sub = ref(z[:, :, :4, :8, :8].cpu())
    torch.testing.assert_close(y2[:, :, :6, :14, :14].detach().cpu(), sub[:, :, :6, :14, :14], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("c,shape,relu", [(32, (2, 5, 12, 16), True), (3, (2, 20, 24, 28), True), (16, (1, 10, 16, 12), False),
                                          (8, (2, 1, 20, 20), False)])
@pytest.mark.parametrize("train", [True, False])
def test_channel_major_batchnorm3d(c, shape, relu, train):
    from sparse2dense_amd.dense3d import FastBatchNorm3d
    torch.manual_seed(c)
    n, d, h, w = shape
    x = torch.randn(n, c, d, h, w) * 2 + 0.3
    bn = FastBatchNorm3d(c, eps=1e-5, momentum=0.1, fused_relu=relu)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2); bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 2)
    ref = nn.BatchNorm3d(c, eps=1e-5, momentum=0.1).double()
    ref.load_state_dict(bn.state_dict())
    bn.train(train); ref.train(train)
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    if relu:
        yr = yr.relu()
    g = torch.randn(yr.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    gr = torch.autograd.grad(yr, [xr, ref.weight, ref.bias], g)
    bn = bn.to(DEV)
    xh = x.to(DEV).requires_grad_(True)
    yh = bn(xh)
    gh = torch.autograd.grad(yh, [xh, bn.weight, bn.bias], g.float().to(DEV))
    torch.testing.assert_close(yh.cpu().double(), yr, rtol=1e-4, atol=1e-4)
    for a, b in zip(gh, gr):
        assert (a.cpu().double() - b).abs().max().item() <= 1e-3 * (b.abs().max().item() + 1e-9) + 1e-5
    if train:
        torch.testing.assert_close(bn.running_mean.cpu().double(), ref.running_mean, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(bn.running_var.cpu().double(), ref.running_var, rtol=1e-4, atol=1e-5)


# ---------------------------------------------------------------------------------------------------
# bf16 matrix-core ConvTranspose3d (csrc/convt3d_mfma.hip): float64 torch reference on the HOST evaluated on the SAME
# bf16-rounded operands (x, weight, and - for the gradients - dout), so what is left is fp32 accumulation order:
# 1e-4 of max on outputs and gradients (K <= 256 products per output, <= ~1e6 per weight-gradient entry).
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cin,cout,shape", [(32, 32, (2, 3, 6, 70)), (16, 3, (1, 4, 7, 66)), (32, 3, (1, 2, 5, 9)), (16, 32, (2, 2, 3, 130)),
                                            (32, 32, (1, 5, 20, 188)), (16, 3, (1, 3, 9, 376)), (32, 3, (2, 2, 3, 40)), (16, 3, (2, 2, 3, 24)),
                                            (16, 2, (1, 1, 1, 8)), (16, 6, (1, 2, 3, 24)), (32, 4, (1, 3, 2, 70))])   # 5..8 channels: the channel-pair mapping
def test_convtranspose3d_bf16_mfma(cin, cout, shape):
    torch.manual_seed(cin * 5 + cout)
    n, d, h, w = shape
    rb = lambda t: t.to(torch.bfloat16).float()
    x = rb(torch.randn(n, cin, d, h, w))
    hip = ConvTranspose3dK4S2(cin, cout, 4, 2, 1)
    with torch.no_grad():
        hip.weight.copy_(rb(hip.weight))
    hip.bf16_compute = True
    ref = nn.ConvTranspose3d(cin, cout, 4, 2, 1)
    ref.load_state_dict(hip.state_dict())
    ref = ref.double()
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    g = rb(torch.randn(yr.shape, generator=torch.Generator().manual_seed(3)))
    gr = torch.autograd.grad(yr, [xr] + list(ref.parameters()), g.double())
    hip = hip.to(DEV)
    xh = x.to(DEV).requires_grad_(True)
    yh = hip(xh)
    gh = torch.autograd.grad(yh, [xh] + list(hip.parameters()), g.to(DEV))
    for name, a, b in [("y", yh, yr)] + [(f"grad{i}", a, b) for i, (a, b) in enumerate(zip(gh, gr))]:
        err = (a.detach().cpu().double() - b.detach()).abs().max().item() / (b.abs().max().item() + 1e-12)
        assert err <= 1e-4, (name, err)


@pytest.mark.parametrize("cin,cout,pos_shape", [(128, 32, (2, 5, 12, 16)), (32, 3, (1, 4, 10, 12)), (32, 1, (2, 3, 8, 8)), (32, 16, (4, 10, 94, 94)),
                                                (3, 3, (2, 20, 94, 94)), (3, 1, (2, 2, 6, 4))])
def test_pointwise_conv3d_weight_gradient_kernel(cin, cout, pos_shape):
    """dW / db of the 1x1x1 convs through the streaming reduction (s2d_pointwise_conv_wgrad_f32) vs float64 on the host"""
    torch.manual_seed(cin + 3 * cout)
    n, d, h, w = pos_shape
    x = torch.randn(n, cin, d, h, w)
    dy = torch.randn(n, cout, d, h, w)
    layer = PointwiseConv3d(cin, cout, 1, 1, 0).to(DEV)
    xh = x.to(DEV).requires_grad_(True)
    layer(xh).backward(dy.to(DEV))
    dw = torch.einsum("ncp,nkp->ck", dy.double().reshape(n, cout, -1), x.double().reshape(n, cin, -1))
    db = dy.double().sum(dim=(0, 2, 3, 4))
    assert (layer.weight.grad.cpu().double().reshape(cout, cin) - dw).abs().max() <= 2e-5 * dw.abs().max() + 1e-4
    assert (layer.bias.grad.cpu().double() - db).abs().max() <= 2e-5 * db.abs().max() + 1e-4


@pytest.mark.parametrize("cin,cout,pos_shape", [(128, 32, (2, 5, 12, 16)), (32, 16, (4, 10, 94, 94)), (32, 16, (1, 3, 5, 4)), (128, 32, (3, 5, 47, 48)),
                                                (32, 7, (2, 2, 9, 12)), (128, 20, (1, 1, 3, 4))])
def test_pointwise_conv3d_weight_gradient_bf16_mfma(cin, cout, pos_shape):
    """s2d_pointwise_conv_wgrad_bf16 (bf16 operands on the matrix cores, one pass) vs float64 on the host over the SAME
    bf16-rounded x and dy; db sums the unrounded dy.  What is left is fp32 accumulation order: 1e-4 of max."""
    from sparse2dense_amd import _lib
    torch.manual_seed(cin + 5 * cout)
    n, d, h, w = pos_shape
    assert _lib.load().s2d_pointwise_conv_wgrad_bf16_supported(cin, cout, d * h * w)
    rb = lambda t: t.to(torch.bfloat16).float()
    x = torch.randn(n, cin, d, h, w)
    dy = torch.randn(n, cout, d, h, w)
    layer = PointwiseConv3d(cin, cout, 1, 1, 0).to(DEV)
    layer.bf16_compute = True
    xh = x.to(DEV).requires_grad_(True)
    layer(xh).backward(dy.to(DEV))
    dw = torch.einsum("ncp,nkp->ck", rb(dy).double().reshape(n, cout, -1), rb(x).double().reshape(n, cin, -1))
    db = dy.double().sum(dim=(0, 2, 3, 4))
    assert (layer.weight.grad.cpu().double().reshape(cout, cin) - dw).abs().max() <= 1e-4 * dw.abs().max()
    assert (layer.bias.grad.cpu().double() - db).abs().max() <= 2e-5 * db.abs().max() + 1e-4


@pytest.mark.parametrize("cin,cout,shape", [(32, 32, (2, 3, 6, 70)), (16, 3, (2, 4, 7, 376)), (16, 3, (1, 2, 3, 24)),
                                            (16, 3, (2, 8, 47, 188)), (32, 32, (1, 8, 94, 188))])   # last two: > 8192 tiles, two-stage fold
def test_convtranspose3d_epilogue_batchnorm_statistics(cin, cout, shape):
    """the (sum, sum of squares) per output channel that the MFMA forward kernel's epilogue emits == the sums over the written output,
    and FastBatchNorm3d fed with them == FastBatchNorm3d doing its own statistics pass"""
    from sparse2dense_amd.dense3d import FastBatchNorm3d
    torch.manual_seed(cin + cout)
    n, d, h, w = shape
    m = ConvTranspose3dK4S2(cin, cout, 4, 2, 1).to(DEV).train()
    m.bf16_compute, m.emit_bn_stats = True, True
    x = torch.randn(n, cin, d, h, w, device=DEV).requires_grad_(True)
    y = m(x)
    st = y._s2d_bn_stats
    s1, s2 = y.detach().double().sum(dim=(0, 2, 3, 4)), (y.detach().double() ** 2).sum(dim=(0, 2, 3, 4))
    assert (st[:cout].double() - s1).abs().max() <= 1e-5 * s2.sqrt().max() * (y[0, 0].numel() * n) ** 0.5
    assert (st[cout:].double() - s2).abs().max() <= 1e-5 * s2.abs().max()
    bn_a, bn_b = FastBatchNorm3d(cout, fused_relu=True).to(DEV).train(), FastBatchNorm3d(cout, fused_relu=True).to(DEV).train()
    za = bn_a(y)                       # consumes y._s2d_bn_stats
    zb = bn_b(y.detach().clone())      # own statistics pass
    assert (za - zb).abs().max() <= 1e-5 * zb.abs().max() + 1e-6
    assert (bn_a.running_var - bn_b.running_var).abs().max() <= 1e-6 * bn_b.running_var.abs().max() + 1e-7
    za.sum().backward()                # the two-output autograd node still back-propagates
    assert x.grad is not None and torch.isfinite(x.grad).all()


@pytest.mark.parametrize("nblocks,c", [(20000, 3), (300001, 32), (9000, 128), (5000, 16), (70000, 200)])
def test_partial_sums_fold_of_long_lists(nblocks, c):
    """s2d_bn_partials_sum_ws_f32: column sums of [nblocks][2c] partial rows (two stages above 1536 rows when 2c <= 256) vs float64,
    bit-identical between runs, count written on request"""
    from sparse2dense_amd import _lib
    lib = _lib.load()
    torch.manual_seed(nblocks % 97 + c)
    part = torch.randn(nblocks, 2 * c, device=DEV)
    ws_bytes = lib.s2d_bn_partials_sum_workspace_bytes(nblocks, c)
    assert (ws_bytes > 0) == (nblocks > 1536 and 2 * c <= 256)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=DEV)
    outs = []
    for _ in range(2):
        st = torch.full((2 * c + 1,), -7.0, device=DEV)
        _lib.check(lib.s2d_bn_partials_sum_ws_f32(part.data_ptr(), nblocks, 12345, c, st.data_ptr(), 1, ws.data_ptr(), ws_bytes,
                                                  torch.cuda.current_stream().cuda_stream), "s2d_bn_partials_sum_ws_f32")
        outs.append(st)
    assert torch.equal(outs[0], outs[1])
    ref = part.double().sum(0)
    assert (outs[0][:2 * c].double() - ref).abs().max() <= 2e-6 * part.abs().sum(0).max()
    assert outs[0][2 * c].item() == 12345.0


@pytest.mark.parametrize("cin,cout,shape", [(32, 32, (2, 3, 6, 70)), (16, 3, (2, 4, 7, 376)), (16, 3, (1, 2, 3, 24)), (32, 32, (1, 5, 47, 188))])
def test_convtranspose3d_bf16_stored_output(cin, cout, shape):
    """r04 (s2d_convt3d_mfma_fwd_stats_y16): the up-sampler writing bf16 == its fp32 output rounded to bf16, bit for bit; the epilogue
    statistics are those of the ROUNDED values; the backward (fp32 gradient in) is the fp32-output layer's"""
    torch.manual_seed(cin * 3 + cout)
    n, d, h, w = shape
    m = ConvTranspose3dK4S2(cin, cout, 4, 2, 1).to(DEV).train()
    m.bf16_compute, m.emit_bn_stats = True, True
    x = torch.randn(n, cin, d, h, w, device=DEV)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    y32 = m(xa)
    m.out_bf16 = True
    y16 = m(xb)
    m.out_bf16 = False
    assert y16.dtype == torch.bfloat16 and y32.dtype == torch.float32
    assert torch.equal(y16, y32.detach().to(torch.bfloat16))
    st = y16._s2d_bn_stats
    yd = y16.detach().double()
    s1, s2 = yd.sum(dim=(0, 2, 3, 4)), (yd ** 2).sum(dim=(0, 2, 3, 4))
    assert (st[:cout].double() - s1).abs().max() <= 1e-5 * s2.sqrt().max() * (yd[0, 0].numel() * n) ** 0.5
    assert (st[cout:].double() - s2).abs().max() <= 1e-5 * s2.abs().max()
    g = torch.randn_like(y32)
    gw32 = torch.autograd.grad(y32, [xa, m.weight], g)
    gw16 = torch.autograd.grad(y16, [xb, m.weight], g)
    assert torch.equal(gw32[0], gw16[0]) and torch.equal(gw32[1], gw16[1])


@pytest.mark.parametrize("n,c,co,depth,h,w", [(2, 128, 32, 5, 12, 16), (1, 128, 32, 5, 47, 48), (2, 64, 16, 2, 10, 6)])
def test_pointwise_conv3d_straight_from_the_nhwc_map(n, c, co, depth, h, w):
    """r05 `dense3d._PwConvNhwcFn`: `Conv3d(C, co, 1)(x.view(n, C, depth, h, w))` as one block-diagonal 1x1 conv on the NHWC bf16 map (no fp32 planar
    copy of x, necks/rpn.py:283-285 + 263-266) against the reference formulation in float64 over the same bf16-rounded operands:
    output / dx one bf16 rounding (6e-3 of max), dW / db 2e-3"""
    from sparse2dense_amd.dense3d import pw_conv_from_nhwc, pw_conv_from_nhwc_supported
    torch.manual_seed(n + c + h)
    conv = PointwiseConv3d(c, co, 1, 1, 0).to(DEV)
    conv.bf16_compute = True
    rb = lambda t: t.to(torch.bfloat16).float()
    with torch.no_grad():
        conv.weight.copy_(rb(conv.weight))
    x = rb(torch.randn(n, c * depth, h, w, device=DEV))
    g = rb(torch.randn(n, co, depth, h, w, device=DEV))
    xa = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    assert pw_conv_from_nhwc_supported(xa, conv, depth)
    ya = pw_conv_from_nhwc(xa, conv, depth)
    assert ya.shape == (n, co, depth, h, w) and ya.dtype == torch.float32 and ya.is_contiguous()
    ya.backward(g)
    ref = nn.Conv3d(c, co, 1).double().to(DEV)
    ref.load_state_dict(conv.state_dict())
    xr = x.double().view(n, c, depth, h, w).requires_grad_(True)
    yr = ref(xr)
    yr.backward(g.double())
    rel = lambda a, r: float((a.double() - r).abs().max() / r.abs().max())
    assert rel(ya.detach(), yr.detach()) <= 6e-3
    assert rel(xa.grad.float().view(n, c, depth, h, w), xr.grad) <= 6e-3
    assert rel(conv.weight.grad, ref.weight.grad) <= 2e-3
    assert rel(conv.bias.grad, ref.bias.grad) <= 2e-3
