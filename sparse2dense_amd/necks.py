"""BEV necks under the reference's registry keys.

  NECKS["RPN"]      /root/reference/det3d/models/necks/rpn.py:24-162   (teacher / plain CenterPoint)
  NECKS["S2D_RPN"]  /root/reference/det3d/models/necks/rpn.py:164-337  (S2D densify module + PCR head
                                                                        + the RPN trunk on F_S_a)
Module names / Sequential indices are the reference's, so state_dicts are interchangeable
(`blocks.0.1.weight` — index 0 is the ZeroPad2d; `encoder_1.0.weight`; `generator_2.3.weight` ...).
Under bf16 autocast on NHWC inputs the layers run on the hand-written kernels behind `dense2d` / `dense3d`
(3x3, 1x1 and the 2x2-stride-2 convs, the ConvTranspose2d(2,2) deblock, depth-wise 7x7, batch norms with the ReLU / GELU fused, the
whole-map LayerNorm, the PCR head with its fused levels, ConvTranspose2d(4,2,1) and the backward of the stride-2 3x3 convs as
parity-class convs); fp32 runs and CPU inputs take the stock torch layer.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .backbones import build_norm_layer
from .dense2d import Conv1x1, Conv2x2S2, Conv3x3, ConvT2x2S2, ConvT4x4S2, DepthwiseConv7, FastBatchNorm2d, WideLayerNorm, fuse_bn_relu
from .dense3d import ConvTranspose3dK4S2, FastBatchNorm3d, PointwiseConv3d
from .heads import (pcr_level, pcr_level_norm, pcr_level_supported, upsample_level, upsample_level_pre_bn_supported, upsample_level_x16_supported,
                    upsample_level_supported)
from .registry import NECKS


def _xavier_uniform_convs(module):
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)


@NECKS.register_module
class RPN(nn.Module):
    def __init__(self, layer_nums, ds_layer_strides, ds_num_filters, us_layer_strides, us_num_filters,
                 num_input_features, norm_cfg=None, name="rpn", logger=None, **kwargs):
        super().__init__()
        self._layer_strides = ds_layer_strides
        self._num_filters = ds_num_filters
        self._layer_nums = layer_nums
        self._upsample_strides = us_layer_strides
        self._num_upsample_filters = us_num_filters
        self._num_input_features = num_input_features
        self._norm_cfg = norm_cfg if norm_cfg is not None else dict(type="BN", eps=1e-3, momentum=0.01)
        self.trunk_channels_last = False   # set by the detector: run blocks/deblocks on NHWC activations
        assert len(ds_layer_strides) == len(layer_nums) == len(ds_num_filters)
        assert len(us_num_filters) == len(us_layer_strides)
        self._upsample_start_idx = len(layer_nums) - len(us_layer_strides)
        ratios = [us_layer_strides[i] / np.prod(ds_layer_strides[: i + self._upsample_start_idx + 1])
                  for i in range(len(us_layer_strides))]
        assert all(r == ratios[0] for r in ratios)

        in_filters = [num_input_features, *ds_num_filters[:-1]]
        blocks, deblocks = [], []
        for i, depth in enumerate(layer_nums):
            blocks.append(self._make_layer(in_filters[i], ds_num_filters[i], depth, stride=ds_layer_strides[i]))
            j = i - self._upsample_start_idx
            if j < 0:
                continue
            up = us_layer_strides[j]
            if up > 1:
                conv = (ConvT2x2S2 if up == 2 else nn.ConvTranspose2d)(ds_num_filters[i], us_num_filters[j], up, stride=up, bias=False)
            else:
                down = int(np.round(1 / up))
                conv = (Conv1x1 if down == 1 else nn.Conv2d)(ds_num_filters[i], us_num_filters[j], down, stride=down, bias=False)
            deblocks.append(nn.Sequential(*fuse_bn_relu([conv, build_norm_layer(self._norm_cfg, us_num_filters[j])[1], nn.ReLU()])))
        self.blocks = nn.ModuleList(blocks)
        self.deblocks = nn.ModuleList(deblocks)
        if logger is not None:
            logger.info("Finish RPN Initialization")

    @property
    def downsample_factor(self):
        factor = np.prod(self._layer_strides)
        if len(self._upsample_strides) > 0:
            factor /= self._upsample_strides[-1]
        return factor

    def _make_layer(self, inplanes, planes, num_blocks, stride=1):
        layers = [nn.ZeroPad2d(1), Conv3x3(inplanes, planes, 3, stride=stride, bias=False),
                  build_norm_layer(self._norm_cfg, planes)[1], nn.ReLU()]
        for j in range(num_blocks):
            layers += [Conv3x3(planes, planes, 3, padding=1, bias=False), build_norm_layer(self._norm_cfg, planes)[1]]
            if j < num_blocks - 1:  # the last conv+BN of a block has no ReLU of its own (rpn.py:142-143)
                layers.append(nn.ReLU())
        return nn.Sequential(*fuse_bn_relu(layers))

    def init_weights(self):
        _xavier_uniform_convs(self)

    def _trunk(self, x, relu_between):
        if self.trunk_channels_last and x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)
        ups = []
        slices = self._slices_possible(x)
        for i, blk in enumerate(self.blocks):
            if relu_between and isinstance(blk[-1], FastBatchNorm2d):   # F.relu(block(x)) with the ReLU fused into the last BN
                for layer in blk[:-1]:
                    x = layer(x)
                x = blk[-1](x, relu=True)
            else:
                x = blk(x)
                if relu_between:
                    x = F.relu(x)
            if i - self._upsample_start_idx >= 0:
                ups.append(self._deblock(i - self._upsample_start_idx, x, ups, slices))
        if ups and all(isinstance(u, tuple) for u in ups):   # every branch wrote its slice of the concatenated tensor (see _deblock)
            from .dense2d import _CatSlicesFn
            return _CatSlicesFn.apply(ups[0][1], *[u[0] for u in ups])
        # (a branch that could not write its slice - other spatial size, a layer off the HIP path: plain concatenation of what there is; a slice
        # that was written is an ordinary tensor aliasing its part of the buffer)
        return torch.cat([u[0] if isinstance(u, tuple) else u for u in ups], dim=1) if ups else x

    def _slices_possible(self, x):
        """decided once per forward, for all deblocks: the batch norms of the up-sampling branches write channel slices of one buffer (= the
        concatenated tensor) when every branch ends in a FastBatchNorm2d that will see an NHWC bf16 map and all widths are multiples of 8"""
        if (os.environ.get("S2D_RPN_CAT", "slices") == "torch" or len(self.deblocks) < 2 or not torch.is_grad_enabled()
                or not (x.is_cuda and x.dim() == 4 and torch.is_autocast_enabled() and torch.get_autocast_gpu_dtype() == torch.bfloat16)):
            return False
        widths = []
        for d in self.deblocks:
            mods = [m for m in d if not isinstance(m, nn.Identity)]
            if not mods or not isinstance(mods[-1], FastBatchNorm2d) or not hasattr(mods[0], "out_channels"):
                return False
            widths.append(int(mods[0].out_channels))
        return sum(widths) % 8 == 0 and all(wd % 8 == 0 for wd in widths)

    def _deblock(self, j, x, ups, slices):
        """deblock j of the trunk.  r04: when every deblock ends in a FastBatchNorm2d on the HIP path (`slices`, decided up front by
        `_slices_possible`) and the branch outputs have one spatial size, the batch norms write their outputs as channel slices of ONE buffer
        = the concatenated tensor of rpn.py:171 (no torch.cat pass, no contiguous copies of the gradient slices in the backward); returns
        (slice, buffer) then, else the plain output (the trunk then concatenates with torch.cat).  S2D_RPN_CAT=torch keeps the concatenation."""
        blk = self.deblocks[j]
        if not slices:
            return blk(x)
        mods = [m for m in blk if not isinstance(m, nn.Identity)]
        for layer in mods[:-1]:
            x = layer(x)
        bn = mods[-1]
        widths = [int([m for m in d if not isinstance(m, nn.Identity)][0].out_channels) for d in self.deblocks]
        prev = [u for u in ups if isinstance(u, tuple)]
        ok = bn._hip_ok(x) and len(prev) == len(ups)
        if ok and prev:
            buf = prev[0][1]
            ok = buf.shape[0] == x.shape[0] and buf.shape[2:] == x.shape[2:]
        elif ok:
            buf = torch.empty((x.shape[0], sum(widths), x.shape[2], x.shape[3]), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
        if not ok:
            return bn(x)
        off = sum(widths[:j])
        return bn(x, out=buf[:, off:off + widths[j]]), buf

    def forward(self, x):
        return self._trunk(x, relu_between=True)  # rpn.py:156


_PCR_STREAMS = {}


def _pcr_side_stream(x):
    """second stream for the PCR branch (S2D_PCR_STREAM=1), one per device; None = run it in line (default).  r03 (B=4 S2D student step):
    23.7 / 23.7 ms in line, 23.8 / 23.3 ms on its own stream - within the noise of separate runs.  r04, in-process A/B with
    tools/ab_step.py (blocks of 40 steps, +-0.03 ms): 20.05 -> 19.78 ms - the branch's HBM-bound streams run beside the trunk's
    matrix-core convs in both passes.  NOT the default: a stress run (tools/side_stress.py: 4-step training runs compared bit for bit
    with the in-line run) showed one run in ~25 whose backbone gradients differed at step 3 with the branch on its own stream - an
    unordered access somewhere in the branch's backward that has not been found; 1.3 % is not worth a wrong gradient."""
    import os
    if os.environ.get("S2D_PCR_STREAM", "0") != "1" or not x.is_cuda or torch.cuda.is_current_stream_capturing():
        return None
    st = _PCR_STREAMS.get(x.device.index)
    if st is None:
        st = _PCR_STREAMS[x.device.index] = torch.cuda.Stream(x.device)
    return st


def _tensors_in(obj):
    if torch.is_tensor(obj):
        return [obj] if obj.is_cuda else []
    if isinstance(obj, dict):
        obj = list(obj.values())
    if isinstance(obj, (list, tuple)):
        return [t for o in obj for t in _tensors_in(o)]
    return []


class _ToNhwcBf16(torch.autograd.Function):
    """contiguous (NCHW) fp32 map -> NHWC bf16 in ONE cast+layout pass (csrc/layout.hip), gradient back to contiguous fp32 in one pass:
    the inverse hand-over of _ToPlanarF32 (the pillar canvas entering the bf16 S2D module, readers/pillar_encoder.py:337-394)"""

    @staticmethod
    def forward(ctx, x):
        n, c, h, w = x.shape
        ctx.hip = bool(x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and c % 8 == 0 and (h * w) % 4 == 0)
        if ctx.hip:
            from . import _lib
            from .dense2d import _stream
            y = torch.empty((n, c, h, w), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
            _lib.check(_lib.load().s2d_nchw_f32_to_nhwc_bf16(x.data_ptr(), n, c, h * w, y.data_ptr(), _stream()), "s2d_nchw_f32_to_nhwc_bf16")
            return y
        return x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    @staticmethod
    def backward(ctx, g):
        n, c, h, w = g.shape
        if ctx.hip and g.dtype == torch.bfloat16 and g.is_contiguous(memory_format=torch.channels_last):
            from . import _lib
            from .dense2d import _stream
            dx = torch.empty((n, c, h, w), dtype=torch.float32, device=g.device)
            _lib.check(_lib.load().s2d_nhwc_bf16_to_nchw_f32(g.data_ptr(), n, c, h * w, dx.data_ptr(), _stream()), "s2d_nhwc_bf16_to_nchw_f32")
            return dx
        return g.to(dtype=torch.float32, memory_format=torch.contiguous_format)


class _ToPlanarF32(torch.autograd.Function):
    """NHWC bf16 map -> contiguous (NCHW) fp32 in ONE cast+layout pass, and the gradient back to NHWC bf16 in one pass: the
    layers on either side then see the layout they are written for (a plain `.float().contiguous()` left the gradient in NCHW
    strides, which sent the GELU / batch-norm backward of the producing layer down strided, non-vectorised kernels)."""

    @staticmethod
    def forward(ctx, x):
        ctx.src_dtype = x.dtype
        ctx.channels_last = x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
        n, c, h, w = x.shape
        ctx.hip = bool(ctx.channels_last and x.is_cuda and x.dtype == torch.bfloat16 and c % 8 == 0 and (h * w) % 4 == 0)
        if ctx.hip:   # tiled transpose (csrc/layout.hip)
            from . import _lib
            from .dense2d import _stream
            y = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
            _lib.check(_lib.load().s2d_nhwc_bf16_to_nchw_f32(x.data_ptr(), n, c, h * w, y.data_ptr(), _stream()), "s2d_nhwc_bf16_to_nchw_f32")
            return y
        return x.to(dtype=torch.float32, memory_format=torch.contiguous_format)

    @staticmethod
    def backward(ctx, g):
        if ctx.hip and g.dtype == torch.float32:
            from . import _lib
            from .dense2d import _stream
            g = g.contiguous()
            n, c, h, w = g.shape
            dx = torch.empty((n, c, h, w), dtype=torch.bfloat16, device=g.device, memory_format=torch.channels_last)
            _lib.check(_lib.load().s2d_nchw_f32_to_nhwc_bf16(g.data_ptr(), n, c, h * w, dx.data_ptr(), _stream()), "s2d_nchw_f32_to_nhwc_bf16")
            return dx
        return g.to(dtype=ctx.src_dtype, memory_format=torch.channels_last if ctx.channels_last else torch.contiguous_format)


def _cbg(*convs_and_channels):
    """conv -> BatchNorm2d -> GELU groups of the S2D module (rpn.py:186-253); the drop-in layer classes take the HIP
    kernels on NHWC bf16 inputs and are the stock layers otherwise"""
    layers = []
    for conv, c in convs_and_channels:
        layers += [conv, FastBatchNorm2d(c), nn.GELU()]
    return nn.Sequential(*fuse_bn_relu(layers))


def _convnext(c, hw):
    return nn.Sequential(DepthwiseConv7(c, c, kernel_size=7, padding=3, groups=c), WideLayerNorm([c, hw, hw], eps=1e-6),
                         Conv1x1(c, 4 * c, 1), nn.GELU(), Conv1x1(4 * c, c, 1))


@NECKS.register_module
class S2D_RPN(RPN):
    def __init__(self, layer_nums, ds_layer_strides, ds_num_filters, us_layer_strides, us_num_filters,
                 num_input_features, norm_cfg=None, name="rpn", logger=None, **kwargs):
        super().__init__(layer_nums, ds_layer_strides, ds_num_filters, us_layer_strides, us_num_filters,
                         num_input_features, norm_cfg, name, logger)
        c = num_input_features
        # ---- S2D module (rpn.py:186-253): 188 -> 94 -> 47 -> 3x ConvNeXt -> 94 -> 188 ----
        self.encoder_1 = _cbg((Conv2x2S2(c, 256, 2, 2), 256), (Conv3x3(256, 256, 3, 1, 1), 256))
        self.encoder_2 = _cbg((Conv3x3(256, 256, 3, 2, 1), 256), (Conv3x3(256, 256, 3, 1, 1), 256))
        self.convnext_block_1 = _convnext(256, 47)
        self.convnext_block_2 = _convnext(256, 47)
        self.convnext_block_3 = _convnext(256, 47)
        self.decoder_1 = _cbg((ConvT4x4S2(256, 256, 4, 2, 1), 256))
        self.decoder_2 = _cbg((Conv3x3(512, 256, 3, 1, 1), 256), (ConvT4x4S2(256, c, 4, 2, 1), c))
        self.fusion_sparse = _cbg((Conv1x1(c, c, 1, 1, 0), c))
        self.fusion_dense = _cbg((Conv1x1(c, c, 1, 1, 0), c))
        self.out_conv = _cbg((Conv1x1(c, 640, 1, 1, 0), 640))
        # ---- PCR point-cloud-reconstruction head (rpn.py:263-296) ----
        # (nn.Conv3d / nn.ConvTranspose3d subclasses: same parameters, HIP streaming kernels on CUDA fp32)
        # BN+ReLU pairs are fused in FastBatchNorm3d; an nn.Identity keeps the reference's Sequential indices
        def bnr(c):
            return FastBatchNorm3d(c, fused_relu=True)

        self.generator_1 = nn.Sequential(PointwiseConv3d(128, 32, 1, 1, 0), bnr(32), nn.Identity(),
                                         ConvTranspose3dK4S2(32, 32, 4, 2, 1), bnr(32), nn.Identity())
        self.gen_out_4 = nn.Sequential(PointwiseConv3d(32, 3, 1, 1, 0))
        self.gen_mask_4 = nn.Sequential(PointwiseConv3d(32, 1, 1, 1, 0))
        self.generator_2 = nn.Sequential(PointwiseConv3d(32, 16, 1, 1, 0), bnr(16), nn.Identity(),
                                         ConvTranspose3dK4S2(16, 3, 4, 2, 1), bnr(3), nn.Identity())
        self.gen_out_2 = nn.Sequential(PointwiseConv3d(3, 3, 1, 1, 0))
        self.gen_mask_2 = nn.Sequential(PointwiseConv3d(3, 1, 1, 1, 0))
        self.generator_1[3].emit_bn_stats = self.generator_2[3].emit_bn_stats = True   # up-samplers: BN statistics from the epilogue
        # {4: (coors, feats), 2: (coors, feats)} of the recon voxels, set by KD_VoxelNet for ONE forward: the PCR levels then return
        # their losses (0-dim tensors in the gen_mask_* / gen_offset_* slots) instead of the dense logits / offsets
        self.pcr_targets = None

    def _pcr_head(self, x, F_S_b):
        """the PCR point-cloud-reconstruction branch behind F_S_b (rpn.py:263-296, 316-325): out_conv -> generator_1 -> level-4 heads ->
        generator_2 -> level-2 heads; returns (gen_offset_2, gen_mask_2, gen_offset_4, gen_mask_4) - losses when the detector handed the
        recon voxels in (pcr_targets), dense volumes otherwise"""
        n, _, h, w = x.shape
        gen = self.out_conv(F_S_b)
        # PCR head in fp32 / standard layout: its 3-D convs are memory-bound, and MIOpen's
        # BatchNorm3d segfaults on bf16 5-D inputs under autocast (ROCm 7.2)
        with torch.autocast("cuda", enabled=False):
            # r05: when the first 1x1x1 conv can read the NHWC bf16 map directly (dense3d._PwConvNhwcFn) the fp32 planar copy of the
            # 640-channel map (362 MB at B = 4) and its gradient are never made
            from .dense3d import pw_conv_from_nhwc, pw_conv_from_nhwc_supported
            gen_nhwc = gen if (self.pcr_targets is not None and pw_conv_from_nhwc_supported(gen, self.generator_1[0], 5)) else None
            if gen_nhwc is None:
                if gen.dtype in (torch.bfloat16, torch.float16):
                    gen = _ToPlanarF32.apply(gen)
                gen = gen.contiguous().view(n, 128, 5, h, w)
            tg, self.pcr_targets = self.pcr_targets, None
            bn1, bn2 = self.generator_1[4], self.generator_2[4]
            fold = (tg is not None and gen.is_cuda and isinstance(bn1, FastBatchNorm3d) and isinstance(bn2, FastBatchNorm3d)
                    and bn1.training and bn2.training and bn1.fused_relu and bn2.fused_relu)
            first = (lambda: pw_conv_from_nhwc(gen_nhwc, self.generator_1[0], 5)) if gen_nhwc is not None else (lambda: self.generator_1[0](gen))
            # r04: on the fused path each up-sampler + its level is ONE autograd node whose raw output is stored in bf16
            # (heads.upsample_level; S2D_PCR_Y16=0 keeps the r03 two-node fp32 form)
            one_node = (fold and os.environ.get("S2D_PCR_Y16", "1") != "0"
                        and upsample_level_supported(self.generator_1[3], (5, h, w), self.generator_2[0])
                        and upsample_level_supported(self.generator_2[3], (10, 2 * h, 2 * w)))
            if one_node:
                # the BatchNorm3d + ReLU in front of each up-sampler joins its node when the kernels cover the shape: the normalised
                # 90 / 362 MB tensors are never written (heads.upsample_level, pre_bn)
                pre1 = self.generator_1[1] if upsample_level_pre_bn_supported(self.generator_1[3], (5, h, w), self.generator_1[1]) else None
                pre2 = self.generator_2[1] if upsample_level_pre_bn_supported(self.generator_2[3], (10, 2 * h, 2 * w), self.generator_2[1]) else None
                mid = first() if pre1 is not None else self.generator_1[1:3](first())
                # r06: the 16-channel volume between the two levels (362 MB in fp32 at B = 4), its gradient and the second up-sampler's input gradient
                # are stored in bf16 when that node reads one (heads.upsample_level_x16_supported; S2D_PCR_Z16=0: fp32)
                z16 = pre2 is not None and upsample_level_x16_supported(self.generator_2[3], (10, 2 * h, 2 * w), pre2)
                gen_mask_4, gen_offset_4, z = upsample_level(self.generator_1[3], mid, bn1, self.gen_mask_4[0], self.gen_out_4[0], *tg[4],
                                                              next_conv=self.generator_2[0], pre_bn=pre1, z16=z16)
                mid2 = z if pre2 is not None else self.generator_2[1:3](z)
                gen_mask_2, gen_offset_2, _ = upsample_level(self.generator_2[3], mid2, bn2, self.gen_mask_2[0], self.gen_out_2[0], *tg[2], pre_bn=pre2)
                return gen_offset_2, gen_mask_2, gen_offset_4, gen_mask_4
            if gen_nhwc is not None:   # (the branches below read the planar tensor)
                gen = self.generator_1[1:3](first())
                gen_from = 3
            else:
                gen_from = 0
            if fold:
                raw = self.generator_1[gen_from:4](gen)   # ... up to the RAW output of the first up-sampler
                fold = pcr_level_supported(raw, self.generator_2[0])
                gen = raw if fold else self.generator_1[4:](raw)
            else:
                gen = self.generator_1[gen_from:](gen)
            if fold:
                # the detector handed the recon voxels in and the levels' batch norms are ours: BatchNorm3d + ReLU + mask / offset heads +
                # losses (+ the next 1x1x1 conv) run from the raw up-sampler outputs (heads.pcr_level_norm); the gen_* slots carry
                # the 0-dim losses
                gen_mask_4, gen_offset_4, z = pcr_level_norm(gen, bn1, self.gen_mask_4[0], self.gen_out_4[0], *tg[4], next_conv=self.generator_2[0])
                raw2 = self.generator_2[1:4](z)
                if pcr_level_supported(raw2):
                    gen_mask_2, gen_offset_2, _ = pcr_level_norm(raw2, bn2, self.gen_mask_2[0], self.gen_out_2[0], *tg[2])
                else:
                    gen = self.generator_2[4:](raw2)
                    from .heads import mask_offset_loss_sparse
                    gen_mask_2, gen_offset_2 = mask_offset_loss_sparse(self.gen_out_2(gen), self.gen_mask_2(gen), *tg[2])
            elif tg is not None and pcr_level_supported(gen, self.generator_2[0]):
                # the detector handed the recon voxels in: each level's mask / offset heads and losses are evaluated without
                # writing the logits / offset volumes (heads.pcr_level); the gen_* slots carry the 0-dim losses instead
                gen_mask_4, gen_offset_4, z = pcr_level(gen, self.gen_mask_4[0], self.gen_out_4[0], *tg[4], next_conv=self.generator_2[0])
                gen = self.generator_2[1:](z)
                if pcr_level_supported(gen):
                    gen_mask_2, gen_offset_2, _ = pcr_level(gen, self.gen_mask_2[0], self.gen_out_2[0], *tg[2])
                else:
                    from .heads import mask_offset_loss_sparse
                    gen_mask_2, gen_offset_2 = mask_offset_loss_sparse(self.gen_out_2(gen), self.gen_mask_2(gen), *tg[2])
            else:
                gen_offset_4 = self.gen_out_4(gen)
                gen_mask_4 = self.gen_mask_4(gen)
                gen = self.generator_2(gen)
                gen_mask_2 = self.gen_mask_2(gen)
                gen_offset_2 = self.gen_out_2(gen)
        return gen_offset_2, gen_mask_2, gen_offset_4, gen_mask_4

    def forward_s2d(self, x):
        """the S2D module + the PCR head (rpn.py:186-296,300-325): everything of `forward` in front of the RPN trunk ->
        (gen_offset_2, gen_mask_2, gen_offset_4, gen_mask_4, F_S_a, F_S_b).  `forward` = forward_s2d + forward_trunk; the detector's graphed
        path replays the two halves as separate HIP-graph segments (detectors.KD_VoxelNet._dense_call)."""
        out = self.forward(x, _stop_before_trunk=True)
        return out[1:]

    def forward_trunk(self, F_S_a):
        """the RPN trunk WITHOUT the outer ReLU of RPN.forward (rpn.py:327-331 vs :156)"""
        if self.trunk_channels_last and F_S_a.is_cuda:
            F_S_a = F_S_a.contiguous(memory_format=torch.channels_last)
        return self._trunk(F_S_a, relu_between=False)

    def forward(self, x, _stop_before_trunk=False):
        if self.trunk_channels_last and x.is_cuda:   # NHWC end to end: conv, batch norm and GELU all keep the layout
            x = x.contiguous(memory_format=torch.channels_last)
        y_1 = self.encoder_1(x)
        y_2 = self.encoder_2(y_1)
        att = self.convnext_block_1(y_2) + y_2
        att = self.convnext_block_2(att) + att
        att = F.gelu(self.convnext_block_3(att) + att)
        y_3 = torch.cat([self.decoder_1(att), y_1], 1)
        F_S_b = self.decoder_2(y_3)
        F_S_a = self.fusion_dense(F_S_b) + self.fusion_sparse(x)
        side = None
        if self.training:
            side = _pcr_side_stream(x)
            if side is None:
                gen_offset_2, gen_mask_2, gen_offset_4, gen_mask_4 = self._pcr_head(x, F_S_b)
            else:
                # S2D_PCR_STREAM=1: the PCR branch (HBM-bound streaming kernels over 0.4-0.7 GB volumes) runs on a second stream beside the
                # trunk / CenterHead (matrix-core convs) - forward here, and in the backward too: autograd replays every node on the stream
                # of its forward.  The branch joins the main stream after the trunk; what it reads from the main stream is marked.
                cur = torch.cuda.current_stream(x.device)
                side.wait_stream(cur)
                for t in _tensors_in((F_S_b, self.pcr_targets)):
                    t.record_stream(side)
                with torch.cuda.stream(side):
                    gen_offset_2, gen_mask_2, gen_offset_4, gen_mask_4 = self._pcr_head(x, F_S_b)
        else:
            gen_offset_2 = gen_mask_2 = gen_offset_4 = gen_mask_4 = None
        # the trunk WITHOUT the outer ReLU of RPN.forward (rpn.py:327-331 vs :156)
        out = None if _stop_before_trunk else self._trunk(F_S_a, relu_between=False)
        if side is not None:
            cur.wait_stream(side)
            for t in _tensors_in((gen_offset_2, gen_mask_2, gen_offset_4, gen_mask_4)):
                t.record_stream(cur)
        return out, gen_offset_2, gen_mask_2, gen_offset_4, gen_mask_4, F_S_a, F_S_b
