"""CenterHead and its losses under the reference's registry keys.

  HEADS["CenterHead"], SepHead     /root/reference/det3d/models/bbox_heads/center_head.py:65-110,167-291
  FastFocalLoss, RegLoss           /root/reference/det3d/models/losses/centernet_loss.py:6-54
  _transpose_and_gather_feat       /root/reference/det3d/core/utils/center_utils.py:66-80
  distillation losses              /root/reference/det3d/torchie/trainer/trainer.py:38-76,783-805
  mask_offset_loss                 /root/reference/det3d/models/detectors/voxelnet.py:171-185

`CenterHead.predict` (decode + rotated NMS) is inference post-processing and out of scope of the
training hot path.
"""
import copy
import logging
import os
from collections import defaultdict

import torch
import torch.nn.functional as F
from torch import nn

from .dense2d import Conv3x3, FastBatchNorm2d, SmallConv3x3, fuse_bn_relu
from .registry import HEADS, LOSSES


def _gather_feat(feat, ind):
    return feat.gather(1, ind.unsqueeze(2).expand(ind.size(0), ind.size(1), feat.size(2)))


def _transpose_and_gather_feat(feat, ind):
    b, c = feat.size(0), feat.size(1)
    return _gather_feat(feat.permute(0, 2, 3, 1).reshape(b, -1, c), ind)


@LOSSES.register_module
class RegLoss(nn.Module):
    """L1 of the gathered regression vector at <=max_objs centre indices, per output channel."""

    def forward(self, output, mask, ind, target):
        if _loss_maps_ok(output, ind) and target.is_cuda and target.shape[-1] == output.shape[1]:
            return _RegLossFn.apply(output, mask, ind, target)
        pred = _transpose_and_gather_feat(output, ind)
        m = mask.float().unsqueeze(2)
        loss = F.l1_loss(pred * m, target * m, reduction="none") / (m.sum() + 1e-4)
        return loss.transpose(2, 0).sum(dim=2).sum(dim=1)


def _u8(mask):
    return mask.view(torch.uint8) if mask.dtype == torch.bool else (mask if mask.dtype == torch.uint8 else (mask != 0).view(torch.uint8))


def _loss_maps_ok(feat, ind, *others):
    return (feat.is_cuda and feat.dtype == torch.float32 and feat.dim() == 4 and feat.is_contiguous() and ind.dtype == torch.int64
            and all(o.dtype == torch.int64 for o in others))


class _FocalFn(torch.autograd.Function):
    """fast_focal_loss in two launches per direction (csrc/center_loss.hip)"""

    @staticmethod
    def forward(ctx, out, target, ind, mask, cat):
        from . import _lib
        from .dense2d import _ptr, _stream, _ws
        lib = _lib.load()
        b, c, h, w = out.shape
        target = target.float().contiguous()
        ind, cat, mask = ind.contiguous(), cat.contiguous(), _u8(mask).contiguous()
        res = torch.empty(4, dtype=torch.float32, device=out.device)
        ws = _ws(lib.s2d_focal_workspace_bytes(), out.device)
        _lib.check(lib.s2d_focal_fwd(_ptr(out), _ptr(target), _ptr(ind), _ptr(mask), _ptr(cat), b, c, h * w, ind.shape[1], _ptr(res), _ptr(ws),
                                     ws.numel(), _stream()), "s2d_focal_fwd")
        ctx.save_for_backward(out, target, ind, mask, cat, res)
        return res[0]

    @staticmethod
    def backward(ctx, go):
        from . import _lib
        from .dense2d import _ptr, _stream
        out, target, ind, mask, cat, res = ctx.saved_tensors
        b, c, h, w = out.shape
        dout = torch.empty_like(out)
        _lib.check(_lib.load().s2d_focal_bwd(_ptr(out), _ptr(target), _ptr(ind), _ptr(mask), _ptr(cat), b, c, h * w, ind.shape[1], _ptr(res),
                                             _ptr(go.float().reshape(1).contiguous()), _ptr(dout), _stream()), "s2d_focal_bwd")
        return dout, None, None, None, None


class _RegLossFn(torch.autograd.Function):
    """RegLoss in one launch forward, memset + scatter backward (csrc/center_loss.hip)"""

    @staticmethod
    def forward(ctx, output, mask, ind, target):
        from . import _lib
        from .dense2d import _ptr, _stream
        b, c, h, w = output.shape
        target = target.float().contiguous()
        ind, mask = ind.contiguous(), _u8(mask).contiguous()
        res = torch.empty(c + 1, dtype=torch.float32, device=output.device)
        _lib.check(_lib.load().s2d_regloss_fwd(_ptr(output), _ptr(ind), _ptr(mask), _ptr(target), b, c, h * w, ind.shape[1], _ptr(res), _stream()),
                   "s2d_regloss_fwd")
        ctx.save_for_backward(output, mask, ind, target, res)
        return res[:c]

    @staticmethod
    def backward(ctx, go):
        from . import _lib
        from .dense2d import _ptr, _stream
        output, mask, ind, target, res = ctx.saved_tensors
        b, c, h, w = output.shape
        dfeat = torch.empty_like(output)
        _lib.check(_lib.load().s2d_regloss_bwd(_ptr(output), _ptr(ind), _ptr(mask), _ptr(target), b, c, h * w, ind.shape[1], _ptr(res),
                                               _ptr(go.float().contiguous()), _ptr(dfeat), _stream()), "s2d_regloss_bwd")
        return dfeat, None, None, None


def fast_focal_loss(out, target, ind, mask, cat):
    """CornerNet focal loss with gathered positives (centernet_loss.py:33-54)."""
    if _loss_maps_ok(out, ind, cat) and target.shape == out.shape and target.is_cuda:
        return _FocalFn.apply(out, target, ind, mask, cat)
    mask = mask.float()
    neg = (torch.log(1 - out) * out.pow(2) * (1 - target).pow(4)).sum()
    pos_pix = _transpose_and_gather_feat(out, ind)           # B x M x C
    pos_pred = pos_pix.gather(2, cat.unsqueeze(2))          # B x M x 1
    num_pos = mask.sum()
    pos = (torch.log(pos_pred) * (1 - pos_pred).pow(2) * mask.unsqueeze(2)).sum()
    # num_pos == 0  =>  pos == 0 and the reference returns -neg: same value, without its host sync
    return -(pos + neg) / num_pos.clamp(min=1.0)


@LOSSES.register_module
class FastFocalLoss(nn.Module):
    def forward(self, out, target, ind, mask, cat):
        return fast_focal_loss(out, target, ind, mask, cat)


def distill_reg_loss(output, target, mask, ind):
    """MSE between student and teacher box maps at the GT centres (trainer.py:68-76)."""
    pred = _transpose_and_gather_feat(output, ind)
    gt = _transpose_and_gather_feat(target, ind)
    m = mask.float().unsqueeze(2)
    loss = F.mse_loss(pred * m, gt * m, reduction="none") / (m.sum() + 1e-4)
    return loss.transpose(2, 0).sum(dim=2).sum(dim=1)


class _MaskedMseFn(torch.autograd.Function):
    """w_pos*MSE over teacher>0 + w_neg*MSE over the rest in one pass per direction (csrc/losses.hip)"""

    @staticmethod
    def forward(ctx, student, teacher, w_pos, w_neg):
        from . import _lib
        from .dense2d import _ptr, _stream, _ws
        lib = _lib.load()
        out = torch.empty(4, dtype=torch.float32, device=student.device)
        ws = _ws(lib.s2d_masked_mse_workspace_bytes(), student.device)
        _lib.check(lib.s2d_masked_mse_fwd(student.data_ptr(), int(student.dtype == torch.bfloat16), teacher.data_ptr(),
                                          int(teacher.dtype == torch.bfloat16), student.numel(), float(w_pos), float(w_neg), _ptr(out), _ptr(ws),
                                          ws.numel(), _stream()), "s2d_masked_mse_fwd")
        ctx.save_for_backward(student, teacher, out)
        return out[0]

    @staticmethod
    def backward(ctx, go):
        from . import _lib
        from .dense2d import _ptr, _stream
        student, teacher, out = ctx.saved_tensors
        ds = torch.empty_like(student)   # same strides
        _lib.check(_lib.load().s2d_masked_mse_bwd(student.data_ptr(), int(student.dtype == torch.bfloat16), teacher.data_ptr(),
                                                  int(teacher.dtype == torch.bfloat16), student.numel(), _ptr(out),
                                                  _ptr(go.float().reshape(1).contiguous()), ds.data_ptr(), _stream()), "s2d_masked_mse_bwd")
        return ds, None, None, None


def _dense_same_layout(a, b):
    ok = lambda t: t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))
    return a.shape == b.shape and a.stride() == b.stride() and ok(a) and ok(b)


def masked_mse_pair(student, teacher, w_pos, w_neg):
    """w_pos*MSE over teacher>0 + w_neg*MSE over the rest (trainer.py:783-789).  CUDA bf16 / fp32 maps of identical layout take the
    fused kernels; otherwise two masked sums and two counts (no boolean-indexed copies)."""
    if (student.is_cuda and student.dtype in (torch.bfloat16, torch.float32) and teacher.dtype in (torch.bfloat16, torch.float32)
            and student.numel() % 8 == 0 and student.shape == teacher.shape):
        if not _dense_same_layout(student, teacher):   # bring the (gradient-free) teacher map into the student's memory order once
            teacher = torch.empty_like(student, dtype=teacher.dtype).copy_(teacher.detach())
        if _dense_same_layout(student, teacher):
            return _MaskedMseFn.apply(student, teacher.detach(), w_pos, w_neg)
    if student.dtype != torch.float64:   # bf16 feature maps: differences and sums in fp32 (a float64 run keeps its precision)
        student, teacher = student.float(), teacher.float()
    pos = teacher > 0
    d2 = (student - teacher) ** 2
    n_pos = pos.sum()
    n_neg = pos.numel() - n_pos
    s_pos = (d2 * pos).sum()
    s_neg = d2.sum() - s_pos
    return w_pos * s_pos / n_pos + w_neg * s_neg / n_neg


def sparse2dense_loss(F_S_a, F_D_a, F_S_b, F_D_b):
    """CenterPoint branch weights 10/20 on (a), 5/20 on (b) (trainer.py:783-789)."""
    return masked_mse_pair(F_S_a, F_D_a, 10.0, 20.0) + masked_mse_pair(F_S_b, F_D_b, 5.0, 20.0)


DEBUG_SUMS = []


def mask_offset_loss(gen_offset, gen_mask, gt, grid):
    """PCR losses (voxelnet.py:171-185): BCE-with-logits on occupancy with pos_weight=neg/pos and L1
    on offsets at the non-zero GT entries."""
    gt_mask = gt.sum(1) != 0
    count_pos = gt_mask.sum()
    count_neg = (~gt_mask).sum()
    beta = count_neg / count_pos
    loss = F.binary_cross_entropy_with_logits(gen_mask[:, 0], gt_mask.float(), pos_weight=beta)
    tgt = gt[:, :3] - grid * gt_mask[:, None]
    sel = tgt != 0
    n_sel = sel.sum()
    com = ((gen_offset - tgt).abs() * sel).sum() / n_sel   # == F.l1_loss(gen_offset[sel], tgt[sel])
    return loss, com


class _PcrLossFn(torch.autograd.Function):
    """mask_offset_loss from the sparse recon voxels (csrc/losses.hip): no dense target, no metric grid tensor"""

    @staticmethod
    def forward(ctx, gen_offset, gen_mask, coors, feats):
        from . import _lib
        from .dense2d import _ptr, _stream, _ws
        lib = _lib.load()
        gen_offset, gen_mask = gen_offset.contiguous(), gen_mask.contiguous()
        coors, feats = coors.contiguous(), feats.contiguous()
        b, _, d, h, w = gen_offset.shape
        assert gen_mask.shape == (b, 1, d, h, w) and gen_offset.shape[1] == 3 and feats.shape[1] == 5
        out = torch.empty(8, dtype=torch.float32, device=gen_offset.device)
        ws = _ws(lib.s2d_pcr_loss_workspace_bytes(), gen_offset.device)
        _lib.check(lib.s2d_pcr_loss_fwd_f32(_ptr(gen_offset), _ptr(gen_mask), _ptr(coors), _ptr(feats), coors.shape[0], b, d, h, w,
                                            _ptr(out), _ptr(ws), ws.numel(), _stream()), "s2d_pcr_loss_fwd_f32")
        ctx.save_for_backward(gen_offset, gen_mask, coors, feats, out)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, go_mask, go_off):
        from . import _lib
        from .dense2d import _ptr, _stream
        lib = _lib.load()
        gen_offset, gen_mask, coors, feats, out = ctx.saved_tensors
        b, _, d, h, w = gen_offset.shape
        g_mask = torch.empty_like(gen_mask) if ctx.needs_input_grad[1] else None
        g_off = torch.zeros_like(gen_offset) if ctx.needs_input_grad[0] else None
        _lib.check(lib.s2d_pcr_loss_bwd_f32(_ptr(gen_offset), _ptr(gen_mask), _ptr(coors), _ptr(feats), coors.shape[0], b, d, h, w,
                                            _ptr(out), _ptr(go_mask.float().contiguous()), _ptr(go_off.float().contiguous()),
                                            _ptr(g_mask), _ptr(g_off), _stream()), "s2d_pcr_loss_bwd_f32")
        return g_off, g_mask, None, None


def mask_offset_loss_sparse(gen_offset, gen_mask, coors, feats):
    """`mask_offset_loss(gen_offset, gen_mask, SparseConvTensor(feats, coors, [D,H,W]).dense(), metric_grid(...))`
    (voxelnet.py:171-185,203-249) without the dense target: CUDA fp32 only."""
    return _PcrLossFn.apply(gen_offset, gen_mask, coors if coors.dtype == torch.int32 else coors.int(), feats.float())


class _PcrLevelFn(torch.autograd.Function):
    """One PCR level: the mask / offset 1x1x1 heads and their two losses straight from the level's feature volume, plus
    (optionally) the level's next 1x1x1 conv g -> z whose data gradient joins the same backward pass (csrc/losses.hip,
    "Fused PCR level heads").  Returns (mask_loss, offset_loss, z | None)."""

    @staticmethod
    def forward(ctx, g, w_mask, b_mask, w_off, b_off, coors, feats, w2, b2, bf16=False):
        from . import _lib
        from .dense2d import _ptr, _stream, _ws
        from .dense3d import pointwise_conv
        lib = _lib.load()
        g = g.contiguous()
        b, c, d, h, w = g.shape
        hp = torch.cat([w_mask.reshape(-1), w_off.reshape(-1), b_mask.reshape(-1), b_off.reshape(-1)]).float().contiguous()
        coors, feats = coors.contiguous(), feats.contiguous()
        out = torch.empty(8, dtype=torch.float32, device=g.device)
        ws = _ws(lib.s2d_pcr_heads_workspace_bytes(c), g.device)
        _lib.check(lib.s2d_pcr_heads_fwd_f32(_ptr(g), _ptr(hp), _ptr(coors), _ptr(feats), coors.shape[0], b, c, d, h, w, _ptr(out), _ptr(ws),
                                             ws.numel(), _stream()), "s2d_pcr_heads_fwd_f32")
        z = w2d = None
        if w2 is not None:
            w2d = w2.reshape(w2.shape[0], w2.shape[1]).contiguous()
            z = pointwise_conv(g, w2d, b2)
        ctx.save_for_backward(g, hp, coors, feats, out, w2d)
        ctx.shapes = (w_mask.shape, w_off.shape, None if w2 is None else w2.shape, b2 is not None)
        ctx.bf16 = bool(bf16)
        return out[0], out[1], z

    @staticmethod
    def backward(ctx, go_mask, go_off, dz):
        from . import _lib
        from .dense2d import _ptr, _stream, _ws
        lib = _lib.load()
        g, hp, coors, feats, out, w2d = ctx.saved_tensors
        b, c, d, h, w = g.shape
        dev = g.device
        zero = lambda: torch.zeros(1, dtype=torch.float32, device=dev)
        go_mask = zero() if go_mask is None else go_mask.float().reshape(1).contiguous()
        go_off = zero() if go_off is None else go_off.float().reshape(1).contiguous()
        co = 0
        if w2d is not None:
            co = w2d.shape[0]
            dz = torch.zeros((b, co, d, h, w), dtype=torch.float32, device=dev) if dz is None else dz.contiguous()
        dg = torch.empty_like(g)
        grads = torch.empty(4 * c + 4, dtype=torch.float32, device=dev)   # dw_mask | dw_off | db_mask | db_off
        ws = _ws(lib.s2d_pcr_heads_workspace_bytes(c), dev)
        base = grads.data_ptr()
        _lib.check(lib.s2d_pcr_heads_bwd_f32(_ptr(g), _ptr(hp), _ptr(coors), _ptr(feats), coors.shape[0], b, c, d, h, w, _ptr(out), _ptr(go_mask),
                                             _ptr(go_off), _ptr(dz) if co else None, _ptr(w2d) if co else None, co, _ptr(dg), base,
                                             base + 4 * 4 * c, base + 4 * c, base + 4 * (4 * c + 1), _ptr(ws), ws.numel(), _stream()),
                   "s2d_pcr_heads_bwd_f32")
        wm_shape, wo_shape, w2_shape, has_b2 = ctx.shapes
        dw2 = db2 = None
        if co:
            from .dense3d import pointwise_conv_wgrad
            dw2, db2 = pointwise_conv_wgrad(g, dz, has_b2, ctx.bf16)
            dw2 = dw2.reshape(w2_shape)
        return (dg, grads[:c].reshape(wm_shape), grads[4 * c:4 * c + 1], grads[c:4 * c].reshape(wo_shape), grads[4 * c + 1:], None, None,
                dw2, db2, None)


class _PcrLevelNormFn(torch.autograd.Function):
    """A PCR level together with the BatchNorm3d + ReLU in front of it (csrc/losses.hip "PCR level with the preceding BatchNorm3d"):
    input = the RAW ConvTranspose3d output; the normalised volume, its gradient and the batch norm's masked gradient are never
    written.  The batch-norm statistics / finalisation (running stats, SyncBN all-reduces) are the FastBatchNorm3d ones."""

    @staticmethod
    def forward(ctx, y, gamma, beta, w_mask, b_mask, w_off, b_off, coors, feats, w2, b2, bn, bf16, stats=None, z_stats_out=None, z16=False):
        """z16 (r06, direct calls from _UpsampleLevelFn only): z = next_conv(relu(bn(y))) is STORED in bf16 (c = 32 -> co = 16 behind a bf16 y) - its
        reader is the next up-sampler node in its x16 form, and the gradient that comes back for it is bf16 as well"""
        from . import _lib, collective as _collective, hip_ops as H
        from .dense2d import _ptr, _stream, _ws
        from .dense3d import _bncm_reduce
        lib = _lib.load()
        y = y.contiguous()
        b, c, d, h, w = y.shape
        pos, dev = d * h * w, y.device
        sync = _collective.sync_on()
        y16 = y.dtype == torch.bfloat16   # r04: bf16-stored raw up-sampler output (dense3d._ConvT3dFn, out_bf16)
        if stats is None or stats.numel() != 2 * c:   # (else: reduced in the epilogue of the kernel that produced y)
            if y16:
                y, y16 = y.float(), False
            stats = _bncm_reduce("s2d_bncm_stats_f32", (_ptr(y),), b, c, pos, dev)
        count = torch.full((1,), float(b * pos), device=dev)
        ctx.local_ysum, ctx.local_rows = stats[:c], float(b * pos)   # this rank's sum of y per channel (for the up-sampler's bias gradient)
        if sync:
            packed = torch.cat([stats, count])
            _collective.allreduce_sum_(packed)
            stats, count = packed[:-1].contiguous(), packed[-1:].contiguous()
        track = bn.track_running_stats
        fin = H.bn1d_finalize_fwd(stats, count, gamma, beta, bn.eps, bn.momentum if track else 0.0, bn.running_mean if track else None,
                                  bn.running_var if track else None, bn.num_batches_tracked if track else None)
        mean, invstd = fin[0], fin[1]
        norm = torch.cat([fin[2].reshape(-1), fin[3].reshape(-1)]).contiguous()   # scale | shift
        hp = torch.cat([w_mask.reshape(-1), w_off.reshape(-1), b_mask.reshape(-1), b_off.reshape(-1)]).float().contiguous()
        coors, feats = coors.contiguous(), feats.contiguous()
        out = torch.empty(8, dtype=torch.float32, device=dev)
        co, z, w2d = 0, None, None
        if w2 is not None:
            co = w2.shape[0]
            w2d = w2.reshape(co, c).contiguous()
            z16 = bool(z16 and y16 and c == 32 and co == 16)
            z = torch.empty((b, co, d, h, w), dtype=torch.bfloat16 if z16 else torch.float32, device=dev)
        else:
            z16 = False
        ctx.z16 = z16
        ws = _ws(lib.s2d_pcr_level_workspace_bytes(c), dev)
        # the kernel that writes z also reduces its per-channel (sum, sum of squares): the statistics of the BatchNorm3d behind the conv
        zst = torch.empty(2 * co, dtype=torch.float32, device=dev) if (z_stats_out is not None and c == 32 and co == 16) else None
        # r06 site cache: the forward's per-voxel pass keeps every recon voxel's c raw values as one row; the backward's two per-voxel passes read the
        # row instead of c scattered loads per voxel (S2D_PCR_SITE_CACHE=0: off)
        m_sites = coors.shape[0]
        ysite = None
        if m_sites > 0 and os.environ.get("S2D_PCR_SITE_CACHE", "1") != "0":
            ysite = torch.empty((m_sites, 4 if c == 3 else c), dtype=y.dtype, device=dev)
            lib.s2d_pcr_level_site_cache(_ptr(ysite), ysite.numel() * ysite.element_size())
        ctx.ysite = ysite
        _lib.check((lib.s2d_pcr_level_fwd_y16_z16 if z16 else lib.s2d_pcr_level_fwd_y16 if y16 else lib.s2d_pcr_level_fwd_f32)(
            _ptr(y), _ptr(norm), _ptr(hp), _ptr(w2d), _ptr(b2), _ptr(coors), _ptr(feats), coors.shape[0], b, c, co, d, h, w, _ptr(z), _ptr(zst), _ptr(out),
            _ptr(ws), ws.numel(), _stream()), "s2d_pcr_level_fwd")
        if zst is not None:
            z_stats_out.append(zst)
        ctx.save_for_backward(y, norm, hp, coors, feats, out, w2d, gamma, mean, invstd, count)
        ctx.shapes = (w_mask.shape, w_off.shape, None if w2 is None else w2.shape, b2 is not None)
        ctx.bf16, ctx.sync = bool(bf16), sync
        ctx.w2_p, ctx.b2_p = w2, b2   # (the parameters themselves: side.run hands deferred gradients to them)
        return out[0], out[1], z

    @staticmethod
    def backward(ctx, go_mask, go_off, dz):
        from . import _lib, collective as _collective, hip_ops as H
        from .dense2d import _ptr, _stream, _ws
        from .dense3d import pointwise_conv_wgrad
        lib = _lib.load()
        y, norm, hp, coors, feats, out, w2d, gamma, mean, invstd, count = ctx.saved_tensors
        b, c, d, h, w = y.shape
        dev = y.device
        zero = lambda: torch.zeros(1, dtype=torch.float32, device=dev)
        go_mask = zero() if go_mask is None else go_mask.float().reshape(1).contiguous()
        go_off = zero() if go_off is None else go_off.float().reshape(1).contiguous()
        co = 0
        y16 = y.dtype == torch.bfloat16
        dy16 = y16 and getattr(ctx, "dy16", False)   # set by _UpsampleLevelFn: dy never leaves the node and its readers take bf16
        z16 = False
        if w2d is not None:
            co = w2d.shape[0]
            # r06: the bf16-stored z's gradient arrives in bf16 and is read as such (the level's bf16-y / bf16-dy kernels); anything else is widened
            z16 = bool(getattr(ctx, "z16", False) and dy16 and (dz is None or dz.dtype == torch.bfloat16))
            if dz is None:
                dz = torch.zeros((b, co, d, h, w), dtype=torch.bfloat16 if z16 else torch.float32, device=dev)
            else:
                dz = dz.contiguous() if (z16 or dz.dtype == torch.float32) else dz.float().contiguous()
        grads = torch.empty(4 * c + 4, dtype=torch.float32, device=dev)   # dw_mask | dw_off | db_mask | db_off
        sums = torch.empty(2 * c, dtype=torch.float32, device=dev)
        ws = _ws(lib.s2d_pcr_level_workspace_bytes(c), dev)
        args = (_ptr(y), _ptr(norm), _ptr(hp), _ptr(coors), _ptr(feats), coors.shape[0], b, c, d, h, w, _ptr(out), _ptr(go_mask), _ptr(go_off),
                _ptr(dz) if co else None, _ptr(w2d) if co else None, co)
        ysite = getattr(ctx, "ysite", None)
        site = (lambda: lib.s2d_pcr_level_site_cache(_ptr(ysite), ysite.numel() * ysite.element_size())) if ysite is not None else (lambda: None)
        site()
        _lib.check((lib.s2d_pcr_level_bwd_sums_y16_z16 if z16 else lib.s2d_pcr_level_bwd_sums_y16 if y16 else lib.s2d_pcr_level_bwd_sums_f32)(
            *args, _ptr(grads), _ptr(sums), _ptr(ws), ws.numel(), _stream()), "s2d_pcr_level_bwd_sums")
        if os.environ.get("S2D_DEBUG_SUMS2") == "1":   # debugging aid (tools/side_stress.py): the same launch again, results kept for a comparison after the pass
            grads2, sums2 = torch.empty_like(grads), torch.empty_like(sums)
            site()
            _lib.check((lib.s2d_pcr_level_bwd_sums_y16_z16 if z16 else lib.s2d_pcr_level_bwd_sums_y16 if y16 else lib.s2d_pcr_level_bwd_sums_f32)(
                *args, _ptr(grads2), _ptr(sums2), _ptr(ws), ws.numel(), _stream()), "s2d_pcr_level_bwd_sums")
            DEBUG_SUMS.append((c, co, sums, sums2, grads, grads2))
        sums_all = sums
        if ctx.sync:
            sums_all = sums.clone()
            _collective.allreduce_sum_(sums_all)
        fin = H.bn1d_finalize_bwd(sums, sums_all, count, gamma, mean, invstd)
        dgamma, dbeta = fin[0], fin[1]
        abd = torch.cat([fin[2].reshape(-1), fin[3].reshape(-1), fin[4].reshape(-1)]).contiguous()
        # sum over this rank's rows of dy = a g + b y + d, per channel, from the sums at hand (= the bias gradient of the conv that produced
        # y; mathematically zero behind a training-mode batch norm, fp32 rounding noise here as in the reference) - no pass over dy
        ctx.dy_sum = fin[2].reshape(-1) * sums[:c] + fin[3].reshape(-1) * ctx.local_ysum + fin[4].reshape(-1) * ctx.local_rows
        dy = torch.empty(y.shape, dtype=torch.bfloat16 if dy16 else torch.float32, device=dev)
        apply = (lib.s2d_pcr_level_bwd_apply_y16_d16_z16 if z16 else lib.s2d_pcr_level_bwd_apply_y16_d16 if dy16
                 else (lib.s2d_pcr_level_bwd_apply_y16 if y16 else lib.s2d_pcr_level_bwd_apply_f32))
        site()
        _lib.check(apply(*args, _ptr(abd), _ptr(dy), _stream()), "s2d_pcr_level_bwd_apply")
        wm_shape, wo_shape, w2_shape, has_b2 = ctx.shapes
        dw2 = db2 = None
        if co:
            from . import side

            def wgrad():
                dwf, dbf = pointwise_conv_wgrad(y, dz, has_b2, ctx.bf16, norm=norm)
                return dwf.reshape(w2_shape), dbf
            # kind "pcr": part of the graphed segment's second graph (side.GRAPH_DEFER), never on the eager weight-gradient stream
            dw2, db2 = side.run(ctx.w2_p, wgrad, y, dz, norm, kind="pcr", bias=ctx.b2_p if has_b2 else None, pair=True)
            if isinstance(ctx, torch.autograd.function.FunctionCtx):   # (composed by _UpsampleLevelFn otherwise: it strips the marker)
                dw2, db2 = side.undefer(dw2), side.undefer(db2)
        return (dy, dgamma, dbeta, grads[:c].reshape(wm_shape), grads[4 * c:4 * c + 1], grads[c:4 * c].reshape(wo_shape), grads[4 * c + 1:],
                None, None, dw2, db2, None, None, None, None, None)


def pcr_level_norm(y, bn, mask_conv, offset_conv, coors, feats, next_conv=None):
    """`pcr_level(relu(bn(y)), ...)` with the training-mode BatchNorm3d + ReLU folded into the level's kernels: y is the raw output of
    the level's ConvTranspose3d.  Same returns as pcr_level."""
    assert bn.training and bn.affine and getattr(bn, "fused_relu", False) and bn.momentum is not None
    assert mask_conv.bias is not None and offset_conv.bias is not None
    coors = coors if coors.dtype == torch.int32 else coors.int()
    holder = []
    out = _PcrLevelNormFn.apply(y, bn.weight, bn.bias, mask_conv.weight, mask_conv.bias, offset_conv.weight, offset_conv.bias, coors,
                                feats.float(), None if next_conv is None else next_conv.weight, None if next_conv is None else next_conv.bias,
                                bn, bool(getattr(next_conv, "bf16_compute", False)), getattr(y, "_s2d_bn_stats", None), holder)
    if holder and out[2] is not None:
        out[2]._s2d_bn_stats = holder[0]   # read by the FastBatchNorm3d that follows the 1x1x1 conv (dense3d.FastBatchNorm3d.forward)
    return out


class _SubCtx:
    """the slice of an autograd context that the two composed Functions below use"""

    def __init__(self, needs):
        self.needs_input_grad = tuple(needs)
        self.saved_tensors = ()

    def save_for_backward(self, *ts):
        self.saved_tensors = ts

    def mark_non_differentiable(self, *ts):
        pass

    def set_materialize_grads(self, flag):
        pass


class _UpsampleLevelFn(torch.autograd.Function):
    """ConvTranspose3d(4,2,1) + the fused PCR level behind it as ONE autograd node (r04): the raw up-sampler output y lives only inside
    the node, stored in bf16 (half the bytes for the four level passes that read it), and its gradient goes from the level's backward
    to the up-sampler's in fp32 without crossing an autograd edge - autograd casts a gradient to the dtype of the tensor it belongs
    to, which put three 0.1 ms conversion passes over the 362-724 MB gradients into the step when y was a bf16 graph tensor.
    forward = dense3d._ConvT3dFn.forward -> _PcrLevelNormFn.forward, backward the reverse; same kernels, same results."""

    @staticmethod
    def forward(ctx, x, ct_w, ct_b, gamma, beta, w_mask, b_mask, w_off, b_off, coors, feats, w2, b2, bn, bf16_next, z_stats_out, y16,
                pre_gamma=None, pre_beta=None, pre_bn=None, pre_stats=None, z16=False):
        from . import _lib, collective as _collective
        from .dense3d import _ConvT3dFn, bncm_finalize_fwd
        c1 = _SubCtx((ctx.needs_input_grad[0], ctx.needs_input_grad[1], ct_b is not None and ctx.needs_input_grad[2], False, False, False))
        # pre_bn: the BatchNorm3d + ReLU in FRONT of the up-sampler is part of the node too - x is its raw input, the up-sampler's kernels
        # normalise it on load (forward and weight gradient) and the normalised tensor is never written or read
        in_norm = None
        if x.dtype == torch.bfloat16 and (pre_bn is None or pre_stats is None
                                          or not _lib.load().s2d_convt3d_mfma_x16_supported(x.shape[1], ct_w.shape[1], *x.shape[2:])):
            x = x.float()   # (a bf16-stored input is read by the x16 kernels of the pre-norm form only; never on the configured path)
        if pre_bn is not None:
            x = x.contiguous()
            sync = _collective.sync_on()
            mean, invstd, scale, shift, count = bncm_finalize_fwd(x, pre_gamma, pre_beta, pre_bn.eps, sync, pre_bn, True, pre_stats)
            in_norm = torch.cat([scale.reshape(-1), shift.reshape(-1)]).contiguous()
            ctx.pre = (pre_gamma, mean, invstd, count, scale.contiguous(), shift.contiguous(), sync)
        y, stats = _ConvT3dFn.forward(c1, x, ct_w, ct_b, True, True, y16, in_norm)
        c2 = _SubCtx((True,) * 3 + (False,) * 12)
        ml, ol, z = _PcrLevelNormFn.forward(c2, y, gamma, beta, w_mask, b_mask, w_off, b_off, coors, feats, w2, b2, bn, bf16_next, stats, z_stats_out,
                                            bool(z16 and y16))
        # the gradient of y goes from the level's backward straight into the up-sampler's: stored in bf16 as well when the up-sampler's
        # matrix-core kernels read that (they round it to bf16 on load in any case; S2D_PCR_DY16=0 keeps it fp32)
        c2.dy16 = bool(y16 and os.environ.get("S2D_PCR_DY16", "1") != "0"
                       and _lib.load().s2d_convt3d_mfma_d16_supported(x.shape[1], ct_w.shape[1], *x.shape[2:]))
        assert in_norm is None or c2.dy16
        ctx.c1, ctx.c2 = c1, c2
        ctx.has_z = z is not None
        return (ml, ol, z) if z is not None else (ml, ol)

    @staticmethod
    def backward(ctx, go_mask, go_off, dz=None):
        from .dense3d import _ConvT3dFn, bncm_backward
        g2 = _PcrLevelNormFn.backward(ctx.c2, go_mask, go_off, dz)
        dy, dgamma, dbeta, dwm, dbm, dwo, dbo, _, _, dw2, db2 = g2[:11]
        dx, dw, db = _ConvT3dFn.backward(ctx.c1, dy, dout_sum=ctx.c2.dy_sum)[:3]
        dpg = dpb = None
        if getattr(ctx, "pre", None) is not None:
            pre_gamma, mean, invstd, count, scale, shift, sync = ctx.pre
            x = ctx.c1.saved_tensors[0]
            dx, dpg, dpb = bncm_backward(dx, x, pre_gamma, mean, invstd, count, scale, shift, True, sync, True, ctx.needs_input_grad[0])
        from . import side
        u = side.undefer
        return dx, u(dw), db, dgamma, dbeta, dwm, dbm, dwo, dbo, None, None, u(dw2), u(db2), None, None, None, None, dpg, dpb, None, None, None


def upsample_level(ct, x, bn, mask_conv, offset_conv, coors, feats, next_conv=None, y16=True, pre_bn=None, z16=False):
    """`pcr_level_norm(ct(x), bn, ...)` as one node (see _UpsampleLevelFn); ct = dense3d.ConvTranspose3dK4S2 in its bf16-compute mode.
    pre_bn (a training-mode FastBatchNorm3d with fused ReLU, y16 only): `pcr_level_norm(ct(pre_bn(x)), bn, ...)` - the batch norm in front
    of the up-sampler joins the node and its output is never materialised.
    z16 (r06): the returned z is stored in bf16 - pass it only to an `upsample_level(..., pre_bn=...)` whose layer is
    `upsample_level_x16_supported` (it reads a bf16 x and returns a bf16 gradient for it)."""
    assert bn.training and bn.affine and getattr(bn, "fused_relu", False) and bn.momentum is not None
    coors = coors if coors.dtype == torch.int32 else coors.int()
    holder = []
    pre = (None, None, None, None)
    if pre_bn is not None:
        assert y16 and pre_bn.training and pre_bn.affine and getattr(pre_bn, "fused_relu", False) and pre_bn.momentum is not None
        stats = getattr(x, "_s2d_bn_stats", None)
        pre = (pre_bn.weight, pre_bn.bias, pre_bn, stats if stats is not None and stats.numel() == 2 * pre_bn.num_features else None)
    out = _UpsampleLevelFn.apply(x, ct.weight, ct.bias, bn.weight, bn.bias, mask_conv.weight, mask_conv.bias, offset_conv.weight, offset_conv.bias,
                                 coors, feats.float(), None if next_conv is None else next_conv.weight, None if next_conv is None else next_conv.bias,
                                 bn, bool(getattr(next_conv, "bf16_compute", False)), holder, bool(y16), *pre, bool(z16))
    z = out[2] if len(out) > 2 else None
    if holder and z is not None:
        z._s2d_bn_stats = holder[0]
    return out[0], out[1], z


def upsample_level_pre_bn_supported(ct, in_dhw, pre_bn):
    """the batch norm in front of the up-sampler can join the node: a training-mode FastBatchNorm3d + ReLU of the layer's input channels
    on a layer shape where the fold pays (narrow outputs: the 16 -> 3 up-sampler; S2D_PCR_PRE_BN=0 keeps it a node of its own)"""
    from . import _lib
    from .dense3d import FastBatchNorm3d
    if os.environ.get("S2D_PCR_PRE_BN", "1") == "0" or os.environ.get("S2D_PCR_DY16", "1") == "0":
        return False
    if not (isinstance(pre_bn, FastBatchNorm3d) and pre_bn.training and pre_bn.affine and pre_bn.fused_relu and pre_bn.momentum is not None
            and pre_bn.num_features == ct.weight.shape[0] and (int(in_dhw[0]) * int(in_dhw[1]) * int(in_dhw[2])) % 4 == 0):
        return False
    return bool(_lib.load().s2d_convt3d_mfma_norm_supported(ct.weight.shape[0], ct.weight.shape[1], *[int(v) for v in in_dhw]))


def upsample_level_x16_supported(ct, in_dhw, pre_bn):
    """the up-sampler node can take a bf16-STORED raw input (and return its gradient in bf16): the pre-norm form on the narrow 16-channel
    layer (S2D_PCR_Z16=0 keeps the volume between the two levels in fp32)"""
    from . import _lib
    if os.environ.get("S2D_PCR_Z16", "1") == "0" or not upsample_level_pre_bn_supported(ct, in_dhw, pre_bn):
        return False
    return bool(_lib.load().s2d_convt3d_mfma_x16_supported(ct.weight.shape[0], ct.weight.shape[1], *[int(v) for v in in_dhw]))


def upsample_level_supported(ct, in_dhw, next_conv=None):
    """the one-node form needs the matrix-core up-sampler (bf16 compute mode, 32 -> 32 / 16 -> 3 channels) and a level shape the fused
    kernels cover; in_dhw = the (D, H, W) extent of the up-sampler's INPUT"""
    from . import _lib
    from .dense3d import ConvTranspose3dK4S2
    if not (isinstance(ct, ConvTranspose3dK4S2) and ct.bf16_compute and ct.weight.is_cuda and ct.kernel_size == (4, 4, 4)
            and ct.stride == (2, 2, 2) and ct.padding == (1, 1, 1) and ct.output_padding == (0, 0, 0)):
        return False
    lib = _lib.load()
    cin, cout = ct.weight.shape[0], ct.weight.shape[1]
    co = 0 if next_conv is None else next_conv.weight.shape[0]
    cells = 8 * int(in_dhw[0]) * int(in_dhw[1]) * int(in_dhw[2])
    return bool(lib.s2d_convt3d_mfma_supported(cin, cout) and lib.s2d_pcr_heads_supported(cout, co, cells))


def pcr_level_supported(g, next_conv=None):
    """the fused level (heads + losses [+ next 1x1x1 conv]) runs on CUDA fp32 volumes of 32 or 3 channels"""
    if not (torch.is_tensor(g) and g.is_cuda and g.dtype in (torch.float32, torch.bfloat16) and g.dim() == 5):
        return False
    from . import _lib
    co = 0 if next_conv is None else next_conv.weight.shape[0]
    return bool(_lib.load().s2d_pcr_heads_supported(g.shape[1], co, g.shape[2] * g.shape[3] * g.shape[4]))


def pcr_level(g, mask_conv, offset_conv, coors, feats, next_conv=None):
    """(mask_loss, offset_loss, next_conv(g) | None) of one PCR level: `mask_offset_loss(offset_conv(g), mask_conv(g), gt, grid)` with
    the sparse recon voxels (coors i32[M,4], feats f32[M,5]) standing in for the dense gt (voxelnet.py:171-185,203-249)."""
    assert mask_conv.bias is not None and offset_conv.bias is not None
    coors = coors if coors.dtype == torch.int32 else coors.int()
    return _PcrLevelFn.apply(g, mask_conv.weight, mask_conv.bias, offset_conv.weight, offset_conv.bias, coors, feats.float(),
                             None if next_conv is None else next_conv.weight, None if next_conv is None else next_conv.bias,
                             bool(getattr(next_conv, "bf16_compute", False)))


def metric_grid(n, d, h, w, like):
    """Cell-centre metric coordinates (x,y,z) of a [D,H,W] grid over the Waymo range
    (voxelnet.py:232-236; the x step reuses 150.4/H exactly as the reference does)."""
    zs, ys, xs = torch.meshgrid(torch.arange(d, device=like.device), torch.arange(h, device=like.device),
                                torch.arange(w, device=like.device), indexing="ij")
    ys = ys * (150.4 / h) - 75.2 + (150.4 / h) / 2
    xs = xs * (150.4 / w) - 75.2 + (150.4 / h) / 2
    zs = zs * (6 / d) - 2 + (6 / d) / 2
    return torch.stack([xs, ys, zs], 0)[None].repeat(n, 1, 1, 1, 1).to(like)


def _kaiming_normal_conv(m):
    nn.init.kaiming_normal_(m.weight, a=0, mode="fan_out", nonlinearity="relu")
    if m.bias is not None:
        nn.init.constant_(m.bias, 0)


class SepHead(nn.Module):
    def __init__(self, in_channels, heads, head_conv=64, final_kernel=1, bn=False, init_bias=-2.19, **kwargs):
        super().__init__(**kwargs)
        self.heads = heads
        for head, (classes, num_conv) in heads.items():
            layers = []
            for _ in range(num_conv - 1):
                layers.append(Conv3x3(in_channels, head_conv, final_kernel, 1, final_kernel // 2, bias=True))
                if bn:
                    layers.append(FastBatchNorm2d(head_conv))
                layers.append(nn.ReLU())
            last = SmallConv3x3 if final_kernel == 3 and classes <= 4 else nn.Conv2d
            layers.append(last(head_conv, classes, final_kernel, 1, final_kernel // 2, bias=True))
            fc = nn.Sequential(*fuse_bn_relu(layers))
            if "hm" in head:
                fc[-1].bias.data.fill_(init_bias)
            else:
                for m in fc.modules():
                    if isinstance(m, nn.Conv2d):
                        _kaiming_normal_conv(m)
            setattr(self, head, fc)

    def forward(self, x):
        return {head: getattr(self, head)(x) for head in self.heads}


@HEADS.register_module
class CenterHead(nn.Module):
    def __init__(self, in_channels=[128, ], tasks=[], dataset="nuscenes", weight=0.25, code_weights=[],
                 common_heads=dict(), logger=None, init_bias=-2.19, share_conv_channel=64, num_hm_conv=2,
                 dcn_head=False):
        super().__init__()
        if dcn_head:
            raise NotImplementedError("dcn_head=True (nuScenes DCN configs) is outside the Waymo hot path")
        num_classes = [len(t["class_names"]) for t in tasks]
        self.class_names = [t["class_names"] for t in tasks]
        self.code_weights = code_weights
        self.weight = weight
        self.dataset = dataset
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.crit = FastFocalLoss()
        self.crit_reg = RegLoss()
        self.box_n_dim = 9 if "vel" in common_heads else 7
        self.use_direction_classifier = False
        self.logger = logger or logging.getLogger("CenterHead")
        self.shared_conv = nn.Sequential(*fuse_bn_relu([Conv3x3(in_channels, share_conv_channel, 3, padding=1, bias=True),
                                                        FastBatchNorm2d(share_conv_channel), nn.ReLU(inplace=True)]))
        self.tasks = nn.ModuleList()
        for num_cls in num_classes:
            heads = copy.deepcopy(dict(common_heads))
            heads.update(dict(hm=(num_cls, num_hm_conv)))
            self.tasks.append(SepHead(share_conv_channel, heads, bn=True, init_bias=init_bias, final_kernel=3))

    def forward(self, x, *kwargs):
        x = self.shared_conv(x)
        return [task(x) for task in self.tasks]

    @staticmethod
    def _sigmoid(x):
        return torch.clamp(x.sigmoid_(), min=1e-4, max=1 - 1e-4)

    def loss(self, example, preds_dicts, **kwargs):
        merged = defaultdict(list)
        for task_id, preds in enumerate(preds_dicts):
            preds["hm"] = self._sigmoid(preds["hm"])
            hm_loss = self.crit(preds["hm"], example["hm"][task_id], example["ind"][task_id], example["mask"][task_id],
                                example["cat"][task_id])
            target_box = example["anno_box"][task_id]
            if self.dataset not in ("waymo", "nuscenes"):
                raise NotImplementedError()
            if "vel" in preds:
                preds["anno_box"] = torch.cat((preds["reg"], preds["height"], preds["dim"], preds["vel"], preds["rot"]), 1)
            else:
                preds["anno_box"] = torch.cat((preds["reg"], preds["height"], preds["dim"], preds["rot"]), 1)
                # drop the velocity target: columns [0..5, -2, -1] (two slices - an index list would be an H2D copy per step, which a
                # HIP-graph capture also refuses)
                target_box = torch.cat((target_box[..., :6], target_box[..., -2:]), -1)
            box_loss = self.crit_reg(preds["anno_box"], example["mask"][task_id], example["ind"][task_id], target_box)
            cw = getattr(self, "_code_w", None)   # device copy of the code weights, made once (an H2D copy per step is also not capturable)
            if cw is None or cw.device != box_loss.device or cw.dtype != box_loss.dtype or cw.numel() != len(self.code_weights):
                cw = self._code_w = box_loss.new_tensor(self.code_weights)
            loc_loss = (box_loss * cw).sum()
            loss = hm_loss + self.weight * loc_loss
            # NB: the reference copies the logging scalars to the host here (.detach().cpu(), four
            # blocking syncs per step, center_head.py:283); we keep them on the device.
            ret = {"loss": loss, "hm_loss": hm_loss.detach(), "loc_loss": loc_loss, "loc_loss_elem": box_loss.detach(),
                   "num_positive": example["mask"][task_id].float().sum()}
            for k, v in ret.items():
                merged[k].append(v)
        return merged

    @torch.no_grad()
    def predict(self, example, preds_dicts, test_cfg, **kwargs):
        """Decode + score / range filter + NMS (center_head.py:293-448,452-507).
        Returns one dict per sample: box3d_lidar [n, 7 or 9], scores [n], label_preds [n], metadata.
        test_cfg.double_flip (r06): the batch holds every sample four times - original, y-flipped, x-flipped, both (center_head.py:318-333) -
        and the four decoded maps are flipped back and averaged (hm, height, dim, reg with the flipped offsets mirrored, the rotation's
        sin / cos and the velocities with the flipped components negated) before the boxes are built (center_head.py:343-381,402-412).
        test_cfg.circular_nms (r06): centre-distance NMS with min_radius[task] instead of the rotated-IoU NMS (center_head.py:476-479).
        per_class_nms: the reference's branch is `pass` (center_head.py:417-418: no result is produced) - not supported."""
        get = (lambda k, d=None: test_cfg.get(k, d)) if hasattr(test_cfg, "get") else (lambda k, d=None: getattr(test_cfg, k, d))
        if get("per_class_nms", False):
            raise NotImplementedError("CenterHead.predict: per_class_nms produces no result in the reference either (center_head.py:417-418)")
        double_flip, circular = bool(get("double_flip", False)), bool(get("circular_nms", False))
        nms_cfg = get("nms")
        nget = (lambda k: nms_cfg[k]) if isinstance(nms_cfg, dict) else (lambda k: getattr(nms_cfg, k))
        hm0 = preds_dicts[0]["hm"]
        pcr = get("post_center_limit_range")
        pcr = torch.tensor(pcr, dtype=torch.float32, device=hm0.device) if len(pcr) > 0 else None
        factor, vs, pc0 = get("out_size_factor"), get("voxel_size"), get("pc_range")
        rets = []
        for task_id, preds in enumerate(preds_dicts):
            p = {k: v.float().permute(0, 2, 3, 1).contiguous() for k, v in preds.items()}   # N C H W -> N H W C
            batch, h, w, num_cls = p["hm"].shape
            if double_flip:
                assert batch % 4 == 0, batch
                batch //= 4
                for k in p:   # back to the unflipped frame: [:, 1] was flipped along H (y = -y), [:, 2] along W (x = -x), [:, 3] both
                    v = p[k].reshape(batch, 4, h, w, -1)
                    p[k] = torch.stack([v[:, 0], torch.flip(v[:, 1], dims=[1]), torch.flip(v[:, 2], dims=[2]), torch.flip(v[:, 3], dims=[1, 2])], 1)
            hm, dim = torch.sigmoid(p["hm"]), torch.exp(p["dim"])
            rots, rotc = p["rot"][..., 0:1], p["rot"][..., 1:2]
            reg, hei = p["reg"], p["height"]
            vel = p.get("vel")
            if double_flip:
                hm, hei, dim = hm.mean(1), hei.mean(1), dim.mean(1)
                reg = reg.clone()
                reg[:, 1, ..., 1] = 1 - reg[:, 1, ..., 1]      # y = -y: the offset inside the cell mirrors
                reg[:, 2, ..., 0] = 1 - reg[:, 2, ..., 0]
                reg[:, 3, ..., 0] = 1 - reg[:, 3, ..., 0]
                reg[:, 3, ..., 1] = 1 - reg[:, 3, ..., 1]
                reg = reg.mean(1)
                rots, rotc = rots.clone(), rotc.clone()
                rotc[:, 1] *= -1                                 # y-flip: theta -> pi - theta
                rots[:, 2] *= -1                                 # x-flip: theta -> 2 pi - theta
                rots[:, 3] *= -1
                rotc[:, 3] *= -1
                rots, rotc = rots.mean(1), rotc.mean(1)
                if vel is not None:
                    vel = vel.clone()
                    vel[:, 1, ..., 1] *= -1
                    vel[:, 2, ..., 0] *= -1
                    vel[:, 3] *= -1
                    vel = vel.mean(1)
            hm = hm.reshape(batch, h * w, num_cls)
            dim = dim.reshape(batch, h * w, 3)
            rot = torch.atan2(rots, rotc).reshape(batch, h * w, 1)
            reg = reg.reshape(batch, h * w, 2)
            hei = hei.reshape(batch, h * w, 1)
            ys, xs = torch.meshgrid(torch.arange(0, h, device=hm.device), torch.arange(0, w, device=hm.device), indexing="ij")
            xs = xs.reshape(1, -1, 1).to(hm) + reg[:, :, 0:1]
            ys = ys.reshape(1, -1, 1).to(hm) + reg[:, :, 1:2]
            xs = xs * factor * vs[0] + pc0[0]
            ys = ys * factor * vs[1] + pc0[1]
            parts = [xs, ys, hei, dim] + ([vel.reshape(batch, h * w, 2)] if vel is not None else []) + [rot]
            boxes = torch.cat(parts, dim=2)
            radius = get("min_radius")[task_id] if circular else None
            rets.append(self.post_processing(boxes, hm, get("score_threshold"), pcr, nget("nms_iou_threshold"),
                                             nget("nms_pre_max_size"), nget("nms_post_max_size"), circle_radius=radius))
        meta = example.get("metadata") if isinstance(example, dict) else None
        if meta and double_flip:
            meta = meta[:4 * len(rets[0]):4]
        out = []
        for i in range(len(rets[0])):
            ret, flag = {}, 0
            ret["box3d_lidar"] = torch.cat([r[i]["box3d_lidar"] for r in rets])
            ret["scores"] = torch.cat([r[i]["scores"] for r in rets])
            labels = []
            for j, num_class in enumerate(self.num_classes):   # label offsets of the later tasks (center_head.py:431-437)
                labels.append(rets[j][i]["label_preds"] + flag)
                flag += num_class
            ret["label_preds"] = torch.cat(labels)
            ret["metadata"] = meta[i] if meta else None
            out.append(ret)
        return out

    @torch.no_grad()
    def post_processing(self, batch_box_preds, batch_hm, score_threshold, post_center_range, iou_threshold, pre_max, post_max, circle_radius=None):
        """center_head.py:452-495 per sample: max over classes, score + centre-range mask, rotate_nms_pcdet - or, with circle_radius
        (test_cfg.circular_nms), `_circle_nms` on the centres (center_head.py:476-479,499-507)"""
        from .nms import circle_nms, rotate_nms
        res = []
        for box_preds, hm_preds in zip(batch_box_preds, batch_hm):
            scores, labels = torch.max(hm_preds, dim=-1)
            mask = scores > score_threshold
            if post_center_range is not None:
                mask &= (box_preds[..., :3] >= post_center_range[:3]).all(1) & (box_preds[..., :3] <= post_center_range[3:]).all(1)
            box_preds, scores, labels = box_preds[mask], scores[mask], labels[mask]
            if circle_radius is not None:
                sel = circle_nms(box_preds[:, [0, 1]].float(), scores.float(), circle_radius, post_max)
            else:
                sel = rotate_nms(box_preds[:, [0, 1, 2, 3, 4, 5, -1]].float(), scores.float(), iou_threshold, pre_max, post_max)
            res.append(dict(box3d_lidar=box_preds[sel], scores=scores[sel], label_preds=labels[sel]))
        return res


# ----------------------------------------------------------------------------------------------
# SECOND anchor head — forward only (BASELINE config 1)
# ----------------------------------------------------------------------------------------------
@HEADS.register_module
class Head(nn.Module):
    """three 1x1 convs + NHWC permute (/root/reference/det3d/models/bbox_heads/mg_head.py:199-232)"""

    def __init__(self, num_input, num_pred, num_cls, use_dir=False, num_dir=0, header=True, name="",
                 focal_loss_init=False, **kwargs):
        super().__init__(**kwargs)
        self.use_dir = use_dir
        self.conv_box = nn.Conv2d(num_input, num_pred, 1)
        self.conv_cls = nn.Conv2d(num_input, num_cls, 1)
        if self.use_dir:
            self.conv_dir = nn.Conv2d(num_input, num_dir, 1)

    def forward(self, x):
        ret = {"box_preds": self.conv_box(x).permute(0, 2, 3, 1).contiguous(),
               "cls_preds": self.conv_cls(x).permute(0, 2, 3, 1).contiguous()}
        if self.use_dir:
            ret["dir_cls_preds"] = self.conv_dir(x).permute(0, 2, 3, 1).contiguous()
        return ret


@HEADS.register_module
class MultiGroupHead(nn.Module):
    """Constructor signature and parameter names of the reference head
    (/root/reference/det3d/models/bbox_heads/mg_head.py:386-533); `forward` is on the hot path of
    config 1, the anchor loss / target assignment / NMS are out of scope."""

    def __init__(self, mode="3d", in_channels=[128, ], norm_cfg=None, tasks=[], weights=[], num_classes=[1, ],
                 box_coder=None, with_cls=True, with_reg=True, reg_class_agnostic=False, encode_background_as_zeros=True,
                 loss_norm=None, loss_cls=None, use_sigmoid_score=True, loss_bbox=None, encode_rad_error_by_sin=True,
                 loss_aux=None, direction_offset=0.0, name="rpn", logger=None):
        super().__init__()
        assert with_cls or with_reg
        num_classes = [len(t["class_names"]) for t in tasks]
        self.class_names = [t["class_names"] for t in tasks]
        self.num_anchor_per_locs = [2 * n for n in num_classes]
        self.box_coder = box_coder
        code_size = box_coder["code_size"] if isinstance(box_coder, dict) else box_coder.code_size
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.encode_background_as_zeros = encode_background_as_zeros
        self.use_sigmoid_score = use_sigmoid_score
        self.box_n_dim = code_size
        self.use_direction_classifier = loss_aux is not None
        self.direction_offset = direction_offset
        self.bev_only = mode == "bev"
        self.tasks = nn.ModuleList()
        for num_c, num_a in zip(num_classes, self.num_anchor_per_locs):
            num_cls = num_a * num_c if encode_background_as_zeros else num_a * (num_c + 1)
            num_pred = num_a * (code_size - 2) if self.bev_only else num_a * code_size
            self.tasks.append(Head(in_channels, num_pred, num_cls, use_dir=self.use_direction_classifier,
                                   num_dir=num_a * 2 if self.use_direction_classifier else None, header=False))

    def forward(self, x):
        return [task(x) for task in self.tasks]

    def loss(self, example, preds_dicts, **kwargs):
        raise NotImplementedError("MultiGroupHead.loss (anchor targets / box coders) is out of scope of the hot path")

    def predict(self, example, preds_dicts, test_cfg, **kwargs):
        raise NotImplementedError("MultiGroupHead.predict (anchor decode + NMS) is out of scope of the hot path")
