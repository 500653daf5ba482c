"""Stride-2 transposed convolutions on the tile kernels (csrc/conv2d_nhwc.hip UP mode, conv2d_wgrad.hip STRIDE = 2) against float64
torch restatements on the HOST over the same bf16-rounded operands: ConvTranspose2d(4,2,1) forward / data gradient / weight gradient
(decoder_1 / decoder_2 of the S2D module, /root/reference/det3d/models/necks/rpn.py:217-231) and the backward of a stride-2 3x3 conv
(rpn.py:126-133).  Bars: one bf16 output rounding (6e-3 of max) for bf16 outputs, 2e-3 for fp32 weight gradients."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _nhwc(t):
    return t.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())


@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 47, 47, 256, 256), (1, 9, 13, 128, 256), (1, 94, 94, 256, 128), (3, 5, 3, 128, 128), (1, 58, 58, 64, 64), (2, 7, 9, 64, 128), (1, 6, 5, 192, 64)])
def test_convtranspose2d_k4s2_forward_and_gradients(n, h, w, cin, cout):
    from sparse2dense_amd import dense2d as D
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n * 1000 + h)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cin, cout, 4, 4, generator=g) * 0.05
    bias = torch.randn(cout, generator=g)
    dy = torch.randn(n, cout, 2 * h, 2 * w, generator=g)
    xb, dyb, wb = x.to(torch.bfloat16).double(), dy.to(torch.bfloat16).double(), wt.to(torch.bfloat16).double()
    y_ref = F.conv_transpose2d(xb, wb, bias.double(), stride=2, padding=1)
    dx_ref = F.conv2d(dyb, wb, None, stride=2, padding=1)
    xr = xb.clone().requires_grad_(True)
    wr = wb.clone().requires_grad_(True)
    (F.conv_transpose2d(xr, wr, None, stride=2, padding=1) * dyb).sum().backward()
    torch.testing.assert_close(xr.grad, dx_ref, rtol=1e-9, atol=1e-9)     # the restatement of the data gradient is the conv it claims to be

    xd, dyd, wd = _nhwc(x.to(dev)), _nhwc(dy.to(dev)), wt.to(dev)
    y, partial = D.conv_up(xd, wd, bias.to(dev), 4, bn_stats=True)
    assert y.shape == (n, cout, 2 * h, 2 * w) and y.is_contiguous(memory_format=torch.channels_last)
    assert _rel(y, y_ref) <= 6e-3
    # batch-norm statistics of the STORED outputs
    yf = y.double().cpu()
    s = partial.double().sum(0).cpu()
    assert _rel(s[0], yf.sum((0, 2, 3))) <= 1e-4 or float((s[0] - yf.sum((0, 2, 3))).abs().max()) <= 1e-3 * float(yf.abs().sum((0, 2, 3)).max())
    assert _rel(s[1], (yf * yf).sum((0, 2, 3))) <= 1e-4
    assert _rel(D.conv_up(xd, wd, None, 4), y_ref - bias.double().view(1, -1, 1, 1)) <= 6e-3
    assert _rel(D.conv4x4s2(dyd, wd), dx_ref) <= 6e-3
    assert _rel(D.conv_s2_wgrad(xd, dyd, 4), wr.grad) <= 2e-3


@pytest.mark.parametrize("n,ho,wo,cin,cout", [(2, 47, 47, 128, 256), (1, 7, 10, 256, 256), (1, 94, 94, 128, 128), (1, 58, 58, 64, 128), (2, 9, 6, 64, 192)])
def test_stride2_conv3x3_backward(n, ho, wo, cin, cout):
    from sparse2dense_amd import dense2d as D
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n * 77 + ho)
    x = torch.randn(n, cin, 2 * ho, 2 * wo, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    dy = torch.randn(n, cout, ho, wo, generator=g)
    xr = x.to(torch.bfloat16).double().requires_grad_(True)
    wr = wt.to(torch.bfloat16).double().requires_grad_(True)
    out = F.conv2d(xr, wr, None, stride=2, padding=1)
    assert out.shape[2:] == (ho, wo)
    (out * dy.to(torch.bfloat16).double()).sum().backward()
    xd, dyd, wd = _nhwc(x.to(dev)), _nhwc(dy.to(dev)), wt.to(dev)
    assert _rel(D.conv_up(dyd, wd, None, 3), xr.grad) <= 6e-3
    assert _rel(D.conv_s2_wgrad(dyd, xd, 3), wr.grad) <= 2e-3


def test_modules_take_the_kernels_and_match_the_stock_layers():
    """ConvT4x4S2 / the stride-2 Conv3x3 as modules under bf16 autocast: outputs and all gradients against the stock torch layer in
    float64 on the same bf16-rounded operands"""
    from sparse2dense_amd import dense2d as D
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    for make, ref_fn, shape in (
            (lambda: D.ConvT4x4S2(256, 256, 4, 2, 1), lambda x, w, b: F.conv_transpose2d(x, w, b, stride=2, padding=1), (2, 256, 12, 10)),
            (lambda: D.Conv3x3(128, 256, 3, stride=2, padding=1, bias=False), lambda x, w, b: F.conv2d(x, w, b, stride=2, padding=1), (2, 128, 24, 20))):
        m = make().to(dev)
        x = torch.randn(*shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x)
        dy = torch.randn_like(y)
        y.backward(dy)
        xr = x.detach().double().cpu().requires_grad_(True)
        wr = m.weight.detach().to(torch.bfloat16).double().cpu().requires_grad_(True)
        br = None if m.bias is None else m.bias.detach().double().cpu().requires_grad_(True)
        yr = ref_fn(xr, wr, br)
        yr.backward(dy.double().cpu())
        assert _rel(y, yr) <= 6e-3
        assert _rel(x.grad, xr.grad) <= 6e-3
        assert _rel(m.weight.grad, wr.grad) <= 2e-3
        if br is not None:
            assert _rel(m.bias.grad, br.grad) <= 2e-3


@pytest.mark.parametrize("kind,cin,cout,shape", [("3x3", 128, 128, (2, 37, 41)), ("3x3", 64, 192, (1, 20, 33)), ("1x1", 256, 128, (2, 47, 47)),
                                                  ("1x1", 1024, 256, (1, 13, 9))])
def test_bias_gradient_rides_on_the_weight_gradient_launch(kind, cin, cout, shape):
    """nn.Conv2d bias gradient = per-channel sums of dY, accumulated by the weight-gradient kernel's first kernel-row / input-channel
    workgroups (csrc/conv2d_wgrad.hip do_db) instead of a separate pass over dY: against float64 sums of the same bf16-rounded dY"""
    from sparse2dense_amd import dense2d as D
    dev = torch.device("cuda:0")
    torch.manual_seed(cin + cout)
    m = (D.Conv3x3(cin, cout, 3, 1, 1) if kind == "3x3" else D.Conv1x1(cin, cout, 1)).to(dev)
    n, h, w = shape
    x = torch.randn(n, cin, h, w, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(x)
    dy = torch.randn_like(y) + 0.25
    y.backward(dy)
    ref_db = dy.double().sum((0, 2, 3))
    assert _rel(m.bias.grad, ref_db) <= 1e-5
    wr = m.weight.detach().to(torch.bfloat16).double().cpu().requires_grad_(True)
    xr = x.detach().double().cpu()
    F.conv2d(xr, wr, None, padding=1 if kind == "3x3" else 0).backward(dy.double().cpu())
    assert _rel(m.weight.grad, wr.grad) <= 2e-3
