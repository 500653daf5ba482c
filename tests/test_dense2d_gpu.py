"""Dense 3x3 NHWC bf16 MFMA convolution (csrc/conv2d_nhwc.hip) against torch fp32 on the same bf16-rounded operands.

Tolerance: inputs/weights are rounded to bf16 once (both sides see the rounded values), products accumulate in
fp32 on both sides, the kernel rounds its output to bf16 -> |err| <= 2^-8 relative to the output scale (+ accumulation
order noise ~1e-6); asserted at 6e-3 of max|ref|.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

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
    ref = F.conv2d(x.float(), wr, b, padding=pad)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert (y.float() - ref).abs().max() <= 6e-3 * ref.abs().max()
    dy = torch.randn_like(ref).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xr = x.float().requires_grad_(True)
    F.conv2d(xr, wr, None, padding=pad).backward(dy.float())
    src = dy if pad == 1 else F.pad(dy, (1, 1, 1, 1)).contiguous(memory_format=torch.channels_last)
    dx = D.conv3x3_nhwc(src, D.pack_weights(wsrc, True), None, cout, cin, 1)
    assert dx.shape == x.shape
    assert (dx.float() - xr.grad).abs().max() <= 6e-3 * xr.grad.abs().max()


def test_unsupported_channels_raise():
    from sparse2dense_amd import _lib, dense2d as D
    w = torch.randn(3, 64, 3, 3, device="cuda")
    with pytest.raises(_lib.S2DError):
        D.pack_weights(w)


@pytest.mark.parametrize("pad,bias", [(1, True), (0, False)])
def test_module_autograd_matches_stock_conv(pad, bias):
    from sparse2dense_amd import dense2d as D
    torch.manual_seed(1)
    m = D.Conv3x3(64, 128, 3, padding=pad, bias=bias).cuda()
    ref = torch.nn.Conv2d(64, 128, 3, padding=pad, bias=bias).cuda()
    ref.load_state_dict(m.state_dict())
    x = torch.randn(2, 64, 40, 36, device="cuda")
    dy = torch.randn(2, 128, 40 + 2 * pad - 2, 36 + 2 * pad - 2, device="cuda")
    xa = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ya = m(xa)
    assert ya.dtype == torch.bfloat16
    ya.backward(dy.to(torch.bfloat16))
    # reference in fp32 on bf16-rounded operands
    xb = x.to(torch.bfloat16).float().requires_grad_(True)
    with torch.no_grad():
        ref.weight.copy_(ref.weight.to(torch.bfloat16).float())
    yr = ref(xb)
    yr.backward(dy.to(torch.bfloat16).float())

    def close(a, r, tol):
        assert (a.float() - r).abs().max() <= tol * r.abs().max(), (a.float() - r).abs().max() / r.abs().max()
    close(ya, yr, 6e-3)
    close(xa.grad, xb.grad, 6e-3)
    close(m.weight.grad, ref.weight.grad, 1e-2)      # MIOpen bf16 wgrad (bf16 output rounding)
    if bias:
        close(m.bias.grad, ref.bias.grad, 1e-3)
    # fp32 (no autocast) and CPU inputs take the stock layer
    y32 = m(x)
    assert y32.dtype == torch.float32
