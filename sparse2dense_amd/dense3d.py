"""PCR-head layers on the hand-written streaming kernels (csrc/dense3d.hip).

`PointwiseConv3d` / `ConvTranspose3dK4S2` are drop-in subclasses of `nn.Conv3d(k=1)` and
`nn.ConvTranspose3d(4, 2, 1)` (same parameters, same state_dict keys) used by `S2D_RPN` and the
pillar S2D backbone (/root/reference/det3d/models/necks/rpn.py:263-296,
readers/pillar_encoder.py:304-323).  CUDA fp32 inputs take the HIP path; the weight gradients
are plain GEMMs over the position axis (hipBLASLt through torch.matmul).  Other inputs (CPU
goldens, non-contiguous exotic cases) fall through to the stock torch layer.
"""
import ctypes

import os

import torch

from . import collective as _collective
import torch.nn.functional as F
from torch import nn

from . import _lib
from ._lib import check


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def pointwise_conv(x, w2d, bias):
    """x [N,Cin,*S] fp32 cuda contiguous; w2d [Cout,Cin]; -> [N,Cout,*S]"""
    lib = _lib.load()
    n, cin = x.shape[0], x.shape[1]
    cout = w2d.shape[0]
    pos = x[0, 0].numel()
    out = torch.empty((n, cout) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    check(lib.s2d_pointwise_conv_f32(_ptr(x), _ptr(w2d), _ptr(bias), n, cin, cout, pos, _ptr(out), _stream()),
          "s2d_pointwise_conv_f32")
    return out


def pointwise_conv_wgrad(x, dout, want_db, bf16=False, norm=None):
    """(dW [Cout,Cin], db [Cout] | None) of a 1x1x1 conv over planar fp32 tensors (position count % 4 == 0).  bf16: operands
    rounded to bf16 on the matrix cores (one pass over both tensors) where the shape is supported.  norm (f32[2*Cin] = scale |
    shift): the conv's input was relu(x*scale + shift), applied on the fly (bf16 route) or materialised (fp32 route).  A bf16-stored x
    (r04: the raw up-sampler output) takes the matrix-core kernel only."""
    if x.dtype == torch.bfloat16:
        lib = _lib.load()
        n, cin, cout = dout.shape[0], x.shape[1], dout.shape[1]
        pos = x[0, 0].numel()
        if not lib.s2d_pointwise_conv_wgrad_bf16_supported(cin, cout, pos):
            return pointwise_conv_wgrad(x.float(), dout, want_db, bf16, norm)
        dw = torch.empty((cout, cin), dtype=torch.float32, device=x.device)
        db = torch.empty((cout,), dtype=torch.float32, device=x.device) if want_db else None
        ws = _ws(lib.s2d_pointwise_conv_wgrad_workspace_bytes(cin, cout), x.device)
        if dout.dtype == torch.bfloat16 and cin == 32:   # r06: the bf16-stored gradient of the bf16-stored z
            check(lib.s2d_pointwise_conv_wgrad_norm_x16_d16(_ptr(x), _ptr(norm), _ptr(dout), n, cin, cout, pos, _ptr(dw), _ptr(db), _ptr(ws), ws.numel(),
                                                            _stream()), "s2d_pointwise_conv_wgrad_norm_x16_d16")
            return dw, db
        dout = dout.float() if dout.dtype != torch.float32 else dout
        check(lib.s2d_pointwise_conv_wgrad_norm_x16(_ptr(x), _ptr(norm), _ptr(dout), n, cin, cout, pos, _ptr(dw), _ptr(db), _ptr(ws), ws.numel(),
                                                    _stream()), "s2d_pointwise_conv_wgrad_norm_x16")
        return dw, db
    lib = _lib.load()
    dout = dout.float() if dout.dtype != torch.float32 else dout
    n, cin, cout = dout.shape[0], x.shape[1], dout.shape[1]
    pos = x[0, 0].numel()
    dw = torch.empty((cout, cin), dtype=torch.float32, device=x.device)
    db = torch.empty((cout,), dtype=torch.float32, device=x.device) if want_db else None
    ws = _ws(lib.s2d_pointwise_conv_wgrad_workspace_bytes(cin, cout), x.device)
    if bf16 and lib.s2d_pointwise_conv_wgrad_bf16_supported(cin, cout, pos):
        check(lib.s2d_pointwise_conv_wgrad_norm_bf16(_ptr(x), _ptr(norm), _ptr(dout), n, cin, cout, pos, _ptr(dw), _ptr(db), _ptr(ws), ws.numel(),
                                                     _stream()), "s2d_pointwise_conv_wgrad_norm_bf16")
    else:
        if norm is not None:
            shape = (1, cin) + (1,) * (x.dim() - 2)
            x = torch.relu(x * norm[:cin].view(shape) + norm[cin:].view(shape))
        check(lib.s2d_pointwise_conv_wgrad_f32(_ptr(x), _ptr(dout), n, cin, cout, pos, _ptr(dw), _ptr(db), _ptr(ws), ws.numel(), _stream()),
              "s2d_pointwise_conv_wgrad_f32")
    return dw, db


_DEBUG_CT_WGRAD = os.environ.get("S2D_DEBUG_CT_WGRAD", "")


class _PwConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, bf16=False):
        ctx.bf16 = bool(bf16)
        x = x.contiguous()
        w2d = weight.reshape(weight.shape[0], weight.shape[1]).contiguous()
        ctx.save_for_backward(x, w2d)
        ctx.wshape = weight.shape
        ctx.has_bias = bias is not None
        ctx.weight_p, ctx.bias_p = weight, bias   # (the parameters themselves: side.run hands deferred gradients to them)
        return pointwise_conv(x, w2d, bias)

    @staticmethod
    def backward(ctx, dout):
        x, w2d = ctx.saved_tensors
        dout = dout.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = pointwise_conv(dout, w2d.t().contiguous(), None)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_db:
            n, cin, cout = dout.shape[0], x.shape[1], dout.shape[1]
            pos = x[0, 0].numel()
            if pos % 4 == 0:   # streaming reduction over the positions (csrc/convt3d_mfma.hip pw_wgrad_*)
                from . import side

                def wgrad():
                    dwf, dbf = pointwise_conv_wgrad(x, dout, want_db, ctx.bf16)
                    return dwf.reshape(ctx.wshape), dbf
                # kind "pcr": never on the eager weight-gradient stream (r04: not bit-equal there), but part of the graphed segment's second graph
                dw, db = side.run(ctx.weight_p, wgrad, x, dout, kind="pcr", bias=ctx.bias_p if want_db else None, pair=True)
                dw, db = side.undefer(dw), side.undefer(db)
            else:
                dw = torch.matmul(dout.reshape(n, cout, -1), x.reshape(n, cin, -1).transpose(1, 2)).sum(0).reshape(ctx.wshape)
                db = dout.sum(dim=[0] + list(range(2, dout.dim()))) if want_db else None
        return dx, dw, db, None


# --------------------------------------------------------------------------------------------------
# 1x1x1 conv straight from the NHWC bf16 map (r05)
# --------------------------------------------------------------------------------------------------
class _PwConvNhwcFn(torch.autograd.Function):
    """`Conv3d(C, co, 1)(x.view(n, C, depth, h, w))` for x = an NHWC bf16 map [n, C*depth, h, w] (channel index = c*depth + d, the
    `.view` of necks/rpn.py:283-285), result fp32 planar [n, co, depth, h, w] - WITHOUT the fp32 planar copy of x (362 MB at B = 4) and
    without its gradient: the conv is ONE 1x1 conv on the NHWC map with the block-diagonal weight W'[(o, d), (c, d')] = W[o, c] [d = d'],
    output channels padded to the tile kernels' multiple of 64 (5x the FLOPs of the planar form, on the matrix cores: 45 us instead of
    0.13 ms of fp32 streaming + 0.10 ms of layout hand-over, per direction), followed by the hand-over of the 45 MB result.  Operands are
    rounded to bf16 (as in every conv of the bf16 mode), the result once more before it is widened."""

    @staticmethod
    def _expanded(weight, depth, cop):
        co, c = weight.shape[0], weight.shape[1]
        w2 = weight.detach().reshape(co, c).float()
        wp = torch.zeros((cop, c * depth), dtype=torch.float32, device=weight.device)
        eye = torch.eye(depth, dtype=torch.float32, device=weight.device)
        wp[:co * depth] = torch.einsum("oc,de->odce", w2, eye).reshape(co * depth, c * depth)
        return wp

    @staticmethod
    def forward(ctx, x, weight, bias, depth):
        from .dense2d import _pack_matrix_1x1, conv1x1_nhwc
        lib = _lib.load()
        n, cd, h, w = x.shape
        co = weight.shape[0]
        cop = -(-co * depth // 64) * 64
        packed = _pack_matrix_1x1(weight, ("pw_nhwc", depth, "fwd"), lambda: _PwConvNhwcFn._expanded(weight, depth, cop))
        bp = None
        if bias is not None:
            bp = torch.zeros(cop, dtype=torch.float32, device=x.device)
            bp[:co * depth] = bias.detach().float().repeat_interleave(depth)
        y = conv1x1_nhwc(x, packed, bp, cd, cop)
        out = torch.empty((n, co, depth, h, w), dtype=torch.float32, device=x.device)
        check(lib.s2d_nhwc_bf16_to_nchw_f32_ld(y.data_ptr(), n, co * depth, cop, h * w, out.data_ptr(), _stream()), "s2d_nhwc_bf16_to_nchw_f32_ld")
        ctx.save_for_backward(x, weight)
        ctx.depth, ctx.cop, ctx.has_bias = depth, cop, bias is not None
        ctx.weight_p, ctx.bias_p = weight, bias
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import side
        from .dense2d import _pack_matrix_1x1, _wgrad_1x1, conv1x1_nhwc
        lib = _lib.load()
        x, weight = ctx.saved_tensors
        n, cd, h, w = x.shape
        depth, cop = ctx.depth, ctx.cop
        co, c = weight.shape[0], weight.shape[1]
        dout = dout.float().contiguous()
        dyn = torch.empty((n, cop, h, w), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
        check(lib.s2d_nchw_f32_to_nhwc_bf16_ld(dout.data_ptr(), n, co * depth, cop, h * w, dyn.data_ptr(), _stream()), "s2d_nchw_f32_to_nhwc_bf16_ld")
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            packed_t = _pack_matrix_1x1(weight, ("pw_nhwc", depth, "dgrad"), lambda: _PwConvNhwcFn._expanded(weight, depth, cop), transpose=True)
            dx = conv1x1_nhwc(dyn, packed_t, None, cop, cd)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_db:
            def wgrad():
                dwp = _wgrad_1x1(x, dyn, cd, cop)                                          # [cop, C*depth]
                dwf = dwp[:co * depth].view(co, depth, c, depth).diagonal(dim1=1, dim2=3).sum(-1)   # the diagonal blocks, summed over depth
                dbf = None
                if want_db:   # per-channel sums of the planar gradient
                    dbf = _bncm_reduce("s2d_bncm_stats_f32", (_ptr(dout),), n, co, dout[0, 0].numel(), x.device)[:co]
                return dwf.reshape(weight.shape).to(weight.dtype).contiguous(), dbf
            dw, db = side.run(ctx.weight_p, wgrad, x, dyn, dout, kind="pcr", bias=ctx.bias_p if want_db else None, pair=True)
            dw, db = side.undefer(dw), side.undefer(db)
        return dx, dw, db, None


def pw_conv_from_nhwc_supported(x, conv, depth):
    """x: NHWC bf16 [n, C*depth, h, w] map; conv: a PointwiseConv3d in its bf16-compute mode"""
    import os
    if os.environ.get("S2D_PCR_PW_NHWC", "1") == "0":
        return False
    if not (isinstance(conv, PointwiseConv3d) and conv.bf16_compute and conv.kernel_size == (1, 1, 1) and conv.groups == 1):
        return False
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)):
        return False
    n, cd, h, w = x.shape
    co, c = conv.weight.shape[0], conv.weight.shape[1]
    cop = -(-co * depth // 64) * 64
    lib = _lib.load()
    return bool(cd == c * depth and (co * depth) % 8 == 0 and (h * w) % 4 == 0 and n <= 65535
                and lib.s2d_conv2d3x3_supported(cd, cop) and lib.s2d_conv2d3x3_supported(cop, cd))


def pw_conv_from_nhwc(x, conv, depth):
    return _PwConvNhwcFn.apply(x, conv.weight, conv.bias, depth)


def _convt_packed(weight):
    """bf16 weight images (forward + data gradient) of the MFMA ConvTranspose3d kernels, cached per parameter version"""
    from .dense2d import cached_pack

    def build():
        lib = _lib.load()
        cin, cout = weight.shape[0], weight.shape[1]
        packed = torch.empty(lib.s2d_convt3d_mfma_packed_elems(cin, cout), dtype=torch.bfloat16, device=weight.device)
        wsrc = weight.detach().float().contiguous()
        launch = lambda: check(lib.s2d_convt3d_mfma_pack_weights(_ptr(wsrc), cin, cout, _ptr(packed), _stream()), "s2d_convt3d_mfma_pack_weights")
        launch()
        from .dense2d import register_repack
        register_repack(weight, [("convt3d_mfma",)], wsrc, launch)
        return packed
    return cached_pack(weight, ("convt3d_mfma",), build)


class _ConvT3dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, bf16=False, bn_stats=False, out_bf16=False, in_norm=None):
        """in_norm (direct calls from heads._UpsampleLevelFn only): scale | shift of the BatchNorm3d + ReLU in front of the layer - x is then
        the RAW tensor and the kernels normalise it on load (forward and weight gradient; bf16-stored output and gradient only)"""
        lib = _lib.load()
        x = x.contiguous()
        weight_param = weight
        weight = weight.contiguous()
        n, cin, d, h, w = x.shape
        cout = weight.shape[1]
        ctx.mfma = bool(bf16 and lib.s2d_convt3d_mfma_supported(cin, cout))
        ctx.in_norm = in_norm
        assert in_norm is None or (ctx.mfma and bn_stats and out_bf16 and lib.s2d_convt3d_mfma_norm_supported(cin, cout, d, h, w))
        # r06: a bf16-STORED raw input (direct calls from heads._UpsampleLevelFn only, with in_norm): read by the x16 kernels, gradient returned in bf16
        x16 = x.dtype == torch.bfloat16
        assert not x16 or (in_norm is not None and lib.s2d_convt3d_mfma_x16_supported(cin, cout, d, h, w)), "bf16-stored input: the pre-norm x16 form only"
        # r04: the raw output may be STORED in bf16 (half the bytes for the four passes of the fused PCR level that read it); only with
        # the matrix-core kernel and the statistics epilogue, i.e. on the fused training path
        out_bf16 = bool(out_bf16 and ctx.mfma and bn_stats)
        out = torch.empty((n, cout, 2 * d, 2 * h, 2 * w), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
        stats = None
        if ctx.mfma:   # bf16 compute mode: operands rounded to bf16 in the kernel, fp32 accumulate (csrc/convt3d_mfma.hip)
            partial = None
            if bn_stats:   # the epilogue also produces the statistics of the batch norm that follows
                tiles = lib.s2d_convt3d_mfma_stats_tiles(n, cin, d, h, w)
                partial = torch.empty((tiles, 2, cout), dtype=torch.float32, device=x.device)
            if in_norm is not None:
                fwd_norm = lib.s2d_convt3d_mfma_fwd_stats_y16_norm_x16 if x16 else lib.s2d_convt3d_mfma_fwd_stats_y16_norm
                check(fwd_norm(_ptr(x), _ptr(in_norm), _ptr(_convt_packed(weight)), _ptr(bias), n, cin, cout, d, h, w,
                               _ptr(out), _ptr(partial), _stream()), "s2d_convt3d_mfma_fwd_stats_y16_norm")
            else:
                entry = lib.s2d_convt3d_mfma_fwd_stats_y16 if out_bf16 else lib.s2d_convt3d_mfma_fwd_stats
                check(entry(_ptr(x), _ptr(_convt_packed(weight)), _ptr(bias), n, cin, cout, d, h, w, _ptr(out), _ptr(partial), _stream()),
                      "s2d_convt3d_mfma_fwd_stats")
            if bn_stats:
                stats = torch.empty((2 * cout,), dtype=torch.float32, device=x.device)
                ws = _ws(lib.s2d_bn_partials_sum_workspace_bytes(partial.shape[0], cout), x.device)   # one row per tile: two-stage fold
                check(lib.s2d_bn_partials_sum_ws_f32(_ptr(partial), partial.shape[0], out.numel() // cout, cout, _ptr(stats), 0, _ptr(ws),
                                                     ws.numel(), _stream()), "s2d_bn_partials_sum_ws_f32")
        else:
            check(lib.s2d_convt3d_k4s2p1_fwd_f32(_ptr(x), _ptr(weight), _ptr(bias), n, cin, cout, d, h, w, _ptr(out), _stream()),
                  "s2d_convt3d_k4s2p1_fwd_f32")
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.weight_p = weight_param
        if bn_stats:
            if stats is None:   # fp32 kernels: the separate statistics pass
                stats = _bncm_reduce("s2d_bncm_stats_f32", (_ptr(out),), n, cout, out[0, 0].numel(), x.device)
            ctx.mark_non_differentiable(stats)
            ctx.set_materialize_grads(False)   # no zero-filled gradient tensor for the statistics output in every backward
            return out, stats
        return out

    @staticmethod
    def backward(ctx, dout, *_unused, dout_sum=None):
        """dout_sum (direct calls from heads._UpsampleLevelFn only): per-channel sums of dout the caller already has = the bias gradient"""
        lib = _lib.load()
        x, weight = ctx.saved_tensors
        n, cin, d, h, w = x.shape
        cout = weight.shape[1]
        # r04: the fused PCR level hands its gradient over in bf16 when the matrix-core kernels cover the layer (they round dout to bf16 on load
        # anyway); any other bf16 gradient is widened
        d16 = bool(dout.dtype == torch.bfloat16 and ctx.mfma and lib.s2d_convt3d_mfma_d16_supported(cin, cout, d, h, w))
        in_norm = getattr(ctx, "in_norm", None)
        assert in_norm is None or d16, "the input-norm fold runs with the bf16-stored gradient only"
        dout = dout.contiguous() if d16 else dout.float().contiguous()
        dx = dw = db = None
        x16 = x.dtype == torch.bfloat16   # (forward: the pre-norm x16 form, which implies d16)
        assert not x16 or d16
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if x16:
                check(lib.s2d_convt3d_mfma_dgrad_d16_x16(_ptr(dout), _ptr(_convt_packed(weight)), n, cin, cout, d, h, w, _ptr(dx), _stream()),
                      "s2d_convt3d_mfma_dgrad_d16_x16")
            elif d16:
                check(lib.s2d_convt3d_mfma_dgrad_d16(_ptr(dout), _ptr(_convt_packed(weight)), n, cin, cout, d, h, w, _ptr(dx), _stream()),
                      "s2d_convt3d_mfma_dgrad_d16")
            elif ctx.mfma:
                check(lib.s2d_convt3d_mfma_dgrad(_ptr(dout), _ptr(_convt_packed(weight)), n, cin, cout, d, h, w, _ptr(dx), _stream()),
                      "s2d_convt3d_mfma_dgrad")
            else:
                check(lib.s2d_convt3d_k4s2p1_dgrad_f32(_ptr(dout), _ptr(weight), n, cin, cout, d, h, w, _ptr(dx), _stream()),
                      "s2d_convt3d_k4s2p1_dgrad_f32")
        def wgrad():
            dw = torch.empty_like(weight)
            if _DEBUG_CT_WGRAD and in_norm is not None:   # debugging aid (S2D_DEBUG_CT_WGRAD, tools/side_stress.py): stand-ins for this launch
                from . import _debug
                return _debug.ct_wgrad_stand_in(_DEBUG_CT_WGRAD, dw, x.float() if x16 else x, dout, in_norm, (n, cin, cout, d, h, w),
                                                getattr(ctx, "_dbg_clones", None))
            if x16:
                ws = _ws(lib.s2d_convt3d_mfma_wgrad_workspace_bytes(n, cin, cout, d, h, w), x.device)
                check(lib.s2d_convt3d_mfma_wgrad_d16_norm_x16(_ptr(x), _ptr(in_norm), _ptr(dout), n, cin, cout, d, h, w, _ptr(dw), _ptr(ws), ws.numel(),
                                                              _stream()), "s2d_convt3d_mfma_wgrad_d16_norm_x16")
            elif d16 and in_norm is not None:
                ws = _ws(lib.s2d_convt3d_mfma_wgrad_workspace_bytes(n, cin, cout, d, h, w), x.device)
                check(lib.s2d_convt3d_mfma_wgrad_d16_norm(_ptr(x), _ptr(in_norm), _ptr(dout), n, cin, cout, d, h, w, _ptr(dw), _ptr(ws), ws.numel(),
                                                          _stream()), "s2d_convt3d_mfma_wgrad_d16_norm")
            elif d16:
                ws = _ws(lib.s2d_convt3d_mfma_wgrad_workspace_bytes(n, cin, cout, d, h, w), x.device)
                check(lib.s2d_convt3d_mfma_wgrad_d16(_ptr(x), _ptr(dout), n, cin, cout, d, h, w, _ptr(dw), _ptr(ws), ws.numel(), _stream()),
                      "s2d_convt3d_mfma_wgrad_d16")
            elif ctx.mfma:
                ws = _ws(lib.s2d_convt3d_mfma_wgrad_workspace_bytes(n, cin, cout, d, h, w), x.device)
                check(lib.s2d_convt3d_mfma_wgrad(_ptr(x), _ptr(dout), n, cin, cout, d, h, w, _ptr(dw), _ptr(ws), ws.numel(), _stream()),
                      "s2d_convt3d_mfma_wgrad")
            elif cin <= 32 and cout <= 32:
                ws = torch.empty(max(lib.s2d_convt3d_k4s2p1_wgrad_workspace_bytes(n, cin, cout, d, h, w), 256),
                                 dtype=torch.uint8, device=x.device)
                check(lib.s2d_convt3d_k4s2p1_wgrad_f32(_ptr(x), _ptr(dout), n, cin, cout, d, h, w, _ptr(dw), _ptr(ws),
                                                       ws.numel(), _stream()), "s2d_convt3d_k4s2p1_wgrad_f32")
            else:   # wide layers: 64 plain GEMMs over the positions (hipBLASLt)
                dp = F.pad(dout, (1, 1, 1, 1, 1, 1))
                xf = x.reshape(n, cin, -1)
                for kz in range(4):
                    for ky in range(4):
                        for kx in range(4):
                            sl = dp[:, :, kz:kz + 2 * d:2, ky:ky + 2 * h:2, kx:kx + 2 * w:2].reshape(n, cout, -1)
                            dw[:, :, kz, ky, kx] = torch.matmul(xf, sl.transpose(1, 2)).sum(0)
            return dw
        if ctx.needs_input_grad[1]:
            from . import side
            wp = getattr(ctx, "weight_p", None)
            if _DEBUG_CT_WGRAD == "clone_main" and in_norm is not None:
                ctx._dbg_clones = (x.clone(), dout.clone(), in_norm.clone())
            # (kind "pcr": deferred into the graphed segment's second graph, never on the eager weight-gradient stream; callers that
            # compose this backward - heads._UpsampleLevelFn - strip the DEFERRED marker)
            dw = side.run(wp, wgrad, x, dout, kind="pcr") if wp is not None else wgrad()
        if ctx.has_bias and ctx.needs_input_grad[2]:
            if dout_sum is not None:
                db = dout_sum
            elif d16:
                db = dout.float().sum(dim=(0, 2, 3, 4))
            elif dout[0, 0].numel() % 4 == 0:   # per-channel sums of a planar tensor: the first half of the BN3d statistics pass
                db = _bncm_reduce("s2d_bncm_stats_f32", (_ptr(dout),), n, cout, dout[0, 0].numel(), x.device)[:cout]
            else:
                db = dout.sum(dim=(0, 2, 3, 4))
        if not isinstance(ctx, torch.autograd.function.FunctionCtx):   # composed by heads._UpsampleLevelFn: it strips the marker itself
            return dx, dw, db, None, None, None
        from . import side as _side
        return dx, _side.undefer(dw), db, None, None, None


def _hip_ok(x):
    return x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()


class PointwiseConv3d(nn.Conv3d):
    """nn.Conv3d with kernel_size 1 (stride 1, no padding).  `bf16_compute` (set by the detector in its bf16 mode): the weight
    gradient contracts bf16-rounded operands on the matrix cores; forward and data gradient stay fp32 streams."""

    bf16_compute = False

    def forward(self, x):
        if _hip_ok(x) and self.kernel_size == (1, 1, 1) and self.stride == (1, 1, 1) and self.padding == (0, 0, 0) \
                and self.groups == 1:
            return _PwConvFn.apply(x, self.weight, self.bias, self.bf16_compute)
        return super().forward(x)


class ConvTranspose3dK4S2(nn.ConvTranspose3d):
    """nn.ConvTranspose3d(cin, cout, 4, 2, 1).  `bf16_compute` (set by the detector in its bf16 mode) selects the matrix-core
    kernels: operands rounded to bf16, fp32 accumulate, fp32 NCDHW tensors."""

    bf16_compute = False
    emit_bn_stats = False   # set by the owner when a batch norm consumes the output: y._s2d_bn_stats = [sum | sum of squares] per channel
    out_bf16 = False        # set by the owner for ONE forward whose consumer reads bf16 (the fused PCR level): the raw output is stored in bf16

    def forward(self, x, output_size=None):
        if _hip_ok(x) and self.kernel_size == (4, 4, 4) and self.stride == (2, 2, 2) and self.padding == (1, 1, 1) \
                and self.output_padding == (0, 0, 0) and self.groups == 1 and self.dilation == (1, 1, 1):
            if self.emit_bn_stats and self.training and torch.is_grad_enabled():
                y, stats = _ConvT3dFn.apply(x, self.weight, self.bias, self.bf16_compute, True, self.out_bf16)
                y._s2d_bn_stats = stats
                return y
            return _ConvT3dFn.apply(x, self.weight, self.bias, self.bf16_compute)
        return super().forward(x, output_size)


# --------------------------------------------------------------------------------------------------
# channel-major BatchNorm (NC[D]HW, few channels, ~1e7 positions per plane)
# --------------------------------------------------------------------------------------------------
def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _bncm_reduce(fn_name, args_front, n, c, pos, device):
    lib = _lib.load()
    out = torch.empty((2 * c,), dtype=torch.float32, device=device)
    ws = _ws(lib.s2d_bncm_workspace_bytes(n, c, pos), device)
    check(getattr(lib, fn_name)(*args_front, n, c, pos, _ptr(out), _ptr(ws), ws.numel(), _stream()), fn_name)
    return out


def bncm_finalize_fwd(x, gamma, beta, eps, sync, module, training, stats=None):
    """statistics (reduced here unless the producer handed them over) -> SyncBN exchange -> finalisation (running statistics updated):
    (mean, invstd, scale, shift, count) of a channel-major batch norm over x[n][c][...]"""
    from . import hip_ops as H
    n, c = x.shape[0], x.shape[1]
    pos = x[0, 0].numel()
    dev = x.device
    if training:
        if stats is None:   # (else: the producing kernel's epilogue already reduced them)
            stats = _bncm_reduce("s2d_bncm_stats_f32", (_ptr(x),), n, c, pos, dev)
        count = torch.full((1,), float(n * pos), device=dev)
        if sync:
            packed = torch.cat([stats, count])
            _collective.allreduce_sum_(packed)
            stats, count = packed[:-1].contiguous(), packed[-1:].contiguous()
        track = module.track_running_stats
        fin = H.bn1d_finalize_fwd(stats, count, gamma, beta, eps, module.momentum if track else 0.0,
                                  module.running_mean if track else None, module.running_var if track else None,
                                  module.num_batches_tracked if track else None)
        return fin[0], fin[1], fin[2], fin[3], count
    invstd = torch.rsqrt(module.running_var + eps)
    mean = module.running_mean
    scale = gamma * invstd
    return mean, invstd, scale, beta - mean * scale, torch.full((1,), float(n * pos), device=dev)


def bncm_backward(dy, x, gamma, mean, invstd, count, scale, shift, relu, sync, training, need_dx=True):
    """backward of y = [relu](x * scale + shift) with batch (training) or running statistics: (dx, dgamma, dbeta); the ReLU mask is
    re-derived from x (fma(x, scale, shift) > 0: the forward's expression)"""
    from . import hip_ops as H
    lib = _lib.load()
    dy = dy.contiguous()
    n, c = x.shape[0], x.shape[1]
    pos = x[0, 0].numel()
    x16, g16 = x.dtype == torch.bfloat16, dy.dtype == torch.bfloat16
    typed = bool(relu and (x16 or g16))   # r06: bf16-stored x / dy (and then dx): the typed passes (storage flags per tensor)
    assert (not x16 or g16) and (typed or not (x16 or g16)), "bf16-stored operands: ReLU form, dy in bf16 whenever x is"
    if typed:
        sums = _bncm_reduce("s2d_bncm_bwd_reduce_x_typed", (_ptr(dy), int(g16), _ptr(x), int(x16), _ptr(scale), _ptr(shift)), n, c, pos, x.device)
    elif relu:
        sums = _bncm_reduce("s2d_bncm_bwd_reduce_x_f32", (_ptr(dy), _ptr(x), _ptr(scale), _ptr(shift)), n, c, pos, x.device)
    else:
        sums = _bncm_reduce("s2d_bncm_bwd_reduce_f32", (_ptr(dy), None, _ptr(x), 0), n, c, pos, x.device)
    if training:
        sums_all = sums
        if sync:
            sums_all = sums.clone()
            _collective.allreduce_sum_(sums_all)
        fin = H.bn1d_finalize_bwd(sums, sums_all, count, gamma, mean, invstd)
        dgamma, dbeta, a, b, d = fin[0], fin[1], fin[2], fin[3], fin[4]
    else:
        dbeta = sums[:c]
        dgamma = invstd * (sums[c:] - mean * sums[:c])
        a, b, d = scale.contiguous(), torch.zeros_like(scale), torch.zeros_like(scale)
    dx = None
    if need_dx:
        dx = torch.empty_like(x)
        if typed:
            check(lib.s2d_bncm_bwd_apply_x_typed(_ptr(dy), int(g16), _ptr(x), int(x16), _ptr(scale), _ptr(shift), _ptr(a), _ptr(b), _ptr(d), n, c, pos,
                                                 _ptr(dx), int(x16), _stream()), "s2d_bncm_bwd_apply_x_typed")
        elif relu:
            check(lib.s2d_bncm_bwd_apply_x_f32(_ptr(dy), _ptr(x), _ptr(scale), _ptr(shift), _ptr(a), _ptr(b), _ptr(d), n, c, pos, _ptr(dx), _stream()),
                  "s2d_bncm_bwd_apply_x_f32")
        else:
            check(lib.s2d_bncm_bwd_apply_f32(_ptr(dy), None, _ptr(x), _ptr(a), _ptr(b), _ptr(d), 0, n, c, pos, _ptr(dx), _stream()),
                  "s2d_bncm_bwd_apply_f32")
    return dx, dgamma, dbeta


class _BNChannelMajorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, relu, eps, sync, module, training, stats=None):
        lib = _lib.load()
        x = x.contiguous()
        n, c = x.shape[0], x.shape[1]
        pos = x[0, 0].numel()
        mean, invstd, scale, shift, count = bncm_finalize_fwd(x, gamma, beta, eps, sync, module, training, stats)
        y = torch.empty_like(x)
        check(lib.s2d_bncm_apply_f32(_ptr(x), _ptr(scale.contiguous()), _ptr(shift.contiguous()), int(relu), n, c, pos,
                                     _ptr(y), _stream()), "s2d_bncm_apply_f32")
        # the backward re-derives the ReLU mask from x (fma(x, scale, shift) > 0: the expression the apply kernel evaluated) - y is not kept
        ctx.save_for_backward(x, gamma, mean, invstd, count, scale.contiguous(), shift.contiguous())
        ctx.relu, ctx.sync, ctx.training = relu, sync, training
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, invstd, count, scale, shift = ctx.saved_tensors
        dx, dgamma, dbeta = bncm_backward(dy, x, gamma, mean, invstd, count, scale, shift, ctx.relu, ctx.sync, ctx.training, ctx.needs_input_grad[0])
        return dx, dgamma, dbeta, None, None, None, None, None, None


class FastBatchNorm3d(nn.BatchNorm3d):
    """nn.BatchNorm3d (same parameters / buffers) with an optional fused ReLU; CUDA fp32 inputs whose
    plane size is a multiple of 4 run the channel-major HIP kernels and synchronise their statistics
    across ranks themselves when torch.distributed is initialised."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, fused_relu=False):
        super().__init__(num_features, eps, momentum, affine, track_running_stats)
        self.fused_relu = fused_relu

    def forward(self, x):
        training = self.training or not self.track_running_stats
        sync = training and _collective.sync_on()
        if _hip_ok(x) and x.dim() >= 3 and x[0, 0].numel() % 4 == 0 and self.affine and x.shape[0] * x.shape[1] <= 65535 \
                and self.momentum is not None:
            stats = getattr(x, "_s2d_bn_stats", None) if training else None
            if stats is not None and stats.numel() != 2 * self.num_features:
                stats = None
            return _BNChannelMajorFn.apply(x, self.weight, self.bias, self.fused_relu, self.eps, sync, self, training, stats)
        if sync and x.is_cuda:   # inputs the channel-major kernels do not cover still synchronise (as FastBatchNorm2d does)
            import torch.distributed as dist
            from torch.nn.modules._functions import SyncBatchNorm as _SyncFn
            if self.track_running_stats:
                self.num_batches_tracked += 1
            y = _SyncFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps, self.momentum,
                              dist.group.WORLD, dist.get_world_size())
        else:
            y = super().forward(x)
        return F.relu(y) if self.fused_relu else y
