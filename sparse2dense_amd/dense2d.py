"""BEV 3x3 convolutions on the hand-written NHWC bf16 MFMA kernel (csrc/conv2d_nhwc.hip).

`Conv3x3(nn.Conv2d)` keeps nn.Conv2d's parameters and state_dict keys (it IS one), so the
reference's checkpoints and our golden fixtures load unchanged
(/root/reference/det3d/models/necks/rpn.py:126-145, bbox_heads/center_head.py:209-232).
The HIP path is taken for CUDA inputs under bf16 autocast when the layer is 3x3 / stride 1 /
padding 0|1 / no groups / no dilation and both channel counts are multiples of 64; forward and
data gradient run on the kernel, the weight gradient is MIOpen's (aten.convolution_backward).
Everything else (fp32 runs, CPU goldens, the 2-channel output convs) is the stock layer.
"""
import ctypes

import torch
from torch import nn

from . import _lib
from ._lib import check

ENABLED = True   # bench / tests can switch the kernel off to A/B against MIOpen

_zero_pages = {}


def _zero_page(device):
    z = _zero_pages.get(device)
    if z is None:
        z = torch.zeros(64, dtype=torch.uint8, device=device)
        _zero_pages[device] = z
    return z


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def supported(cin, cout):
    return bool(_lib.load().s2d_conv2d3x3_supported(int(cin), int(cout)))


def pack_weights(weight, transpose_flip=False):
    """weight fp32 [Cout,Cin,3,3] -> bf16 LDS image of the forward (or data-gradient) operand"""
    lib = _lib.load()
    cout, cin = weight.shape[0], weight.shape[1]
    w = weight.detach().float()
    nhwc = (not w.is_contiguous()) and w.is_contiguous(memory_format=torch.channels_last)
    if not nhwc:
        w = w.contiguous()
    packed = torch.empty(9 * cin * cout, dtype=torch.bfloat16, device=weight.device)
    pc_in, pc_out = (cout, cin) if transpose_flip else (cin, cout)
    check(lib.s2d_conv2d3x3_pack_weights_bf16(_ptr(w), pc_in, pc_out, int(transpose_flip), int(nhwc), _ptr(packed), _stream()),
          "s2d_conv2d3x3_pack_weights_bf16")
    return packed


def conv3x3_nhwc(x, packed, bias, cin, cout, pad):
    """x: bf16 [N,cin,H,W] in channels_last memory -> bf16 [N,cout,Ho,Wo] channels_last"""
    lib = _lib.load()
    assert x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last) and x.shape[1] == cin
    n, _, h, w = x.shape
    ho, wo = h + 2 * pad - 2, w + 2 * pad - 2
    y = torch.empty((n, cout, ho, wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    check(lib.s2d_conv2d3x3_nhwc_bf16(_ptr(x), _ptr(packed), _ptr(bias), _ptr(_zero_page(x.device)), n, h, w, cin, cout,
                                      pad, _ptr(y), _stream()), "s2d_conv2d3x3_nhwc_bf16")
    return y


def _nhwc_bf16(t):
    return t.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


class _Conv3x3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, pad):
        xb = _nhwc_bf16(x)
        cout, cin = weight.shape[0], weight.shape[1]
        ctx.save_for_backward(xb, weight)
        ctx.pad = pad
        ctx.has_bias = bias is not None
        b = None if bias is None else bias.detach().float().contiguous()
        return conv3x3_nhwc(xb, pack_weights(weight), b, cin, cout, pad)

    @staticmethod
    def backward(ctx, dy):
        xb, weight = ctx.saved_tensors
        pad = ctx.pad
        cout, cin = weight.shape[0], weight.shape[1]
        dyb = _nhwc_bf16(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # dX = conv(dY, flip(W)^T) with padding 2 - pad_fwd - ... : for a 3x3 stride-1 conv the data gradient is a
            # "full" correlation with padding (2 - pad); pad=1 -> 1.  pad=0 -> 2 is not a kernel mode: pad dY by one
            # ring of zeros and run pad=1.
            if pad == 1:
                src = dyb
            else:
                src = torch.nn.functional.pad(dyb, (1, 1, 1, 1)).contiguous(memory_format=torch.channels_last)
            dx = conv3x3_nhwc(src, pack_weights(weight, transpose_flip=True), None, cout, cin, 1)
        if ctx.needs_input_grad[1]:
            wb = weight.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            _, dwb, _ = torch.ops.aten.convolution_backward(dyb, xb, wb, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1,
                                                            [False, True, False])
            dw = dwb.to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dyb.float().sum(dim=(0, 2, 3))
        return dx, dw, db, None


class Conv3x3(nn.Conv2d):
    def _hip_ok(self, x):
        return (ENABLED and x.is_cuda and x.dim() == 4 and torch.is_autocast_enabled()
                and torch.get_autocast_gpu_dtype() == torch.bfloat16
                and self.kernel_size == (3, 3) and self.stride == (1, 1) and self.dilation == (1, 1) and self.groups == 1
                and self.padding in ((0, 0), (1, 1)) and self.padding_mode == "zeros"
                and self.in_channels % 64 == 0 and self.out_channels % 64 == 0)

    def forward(self, x):
        if self._hip_ok(x):
            return _Conv3x3Fn.apply(x, self.weight, self.bias, self.padding[0])
        return super().forward(x)
