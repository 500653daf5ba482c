"""Readers and 3-D sparse backbones under the reference's registry keys.

  READERS["VoxelFeatureExtractorV3"]   /root/reference/det3d/models/readers/voxel_encoder.py:8-24
  BACKBONES["SpMiddleResNetFHD"]       /root/reference/det3d/models/backbones/scn.py:88-185
  BACKBONES["SpMiddleFHD"]             /root/reference/det3d/models/backbones/scn.py:187-289
  SparseBasicBlock                     /root/reference/det3d/models/backbones/scn.py:42-85

Parameter names, Sequential indices and the spconv weight layout [kD,kH,kW,Cin,Cout] are those of
the reference, so its checkpoints load (SURVEY.md §8(b) state_dict contract).  All geometry
(4 SubM + 4 strided rulebooks) is planned before the first feature kernel so that the four host
reads of N_out happen while the device is otherwise idle.
"""
import numpy as np
import torch
from torch import nn

from . import hip_ops as H
from .registry import BACKBONES, READERS
from .dense2d import FastBatchNorm2d
from .spconv import (FeatureBatchNorm1d, SparseConv3d, SparseConvTensor, SparseModule, SparseSequential, SubMConv3d)

NORM_LAYERS = {"BN": ("bn", FastBatchNorm2d), "BN1d": ("bn1d", FeatureBatchNorm1d), "GN": ("gn", nn.GroupNorm)}


def build_norm_layer(cfg, num_features, postfix=""):
    """(name, layer) like det3d/models/utils/norm.py:67-108; BN1d maps to the HIP-backed drop-in."""
    cfg = dict(cfg)
    kind = cfg.pop("type")
    if kind not in NORM_LAYERS:
        raise KeyError(f"Unrecognized norm type {kind}")
    abbr, cls = NORM_LAYERS[kind]
    requires_grad = cfg.pop("requires_grad", True)
    cfg.setdefault("eps", 1e-5)
    layer = cls(num_channels=num_features, **cfg) if kind == "GN" else cls(num_features, **cfg)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return abbr + str(postfix), layer


@READERS.register_module
class VoxelFeatureExtractorV3(nn.Module):
    def __init__(self, num_input_features=4, norm_cfg=None, name="VoxelFeatureExtractorV3"):
        super().__init__()
        self.name = name
        self.num_input_features = num_input_features

    def forward(self, features, num_voxels, coors=None):
        assert self.num_input_features == features.shape[-1]
        total = features[:, :, : self.num_input_features].sum(dim=1)
        return (total / num_voxels.type_as(features).view(-1, 1)).contiguous()


def conv3x3(in_planes, out_planes, stride=1, indice_key=None, bias=True):
    return SubMConv3d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=bias, indice_key=indice_key)


class SparseBasicBlock(SparseModule):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, norm_cfg=None, downsample=None, indice_key=None):
        super().__init__()
        if norm_cfg is None:
            norm_cfg = dict(type="BN1d", eps=1e-3, momentum=0.01)
        bias = norm_cfg is not None  # evaluates True, as in the reference (scn.py:59)
        self.conv1 = conv3x3(inplanes, planes, stride, indice_key=indice_key, bias=bias)
        self.bn1 = build_norm_layer(norm_cfg, planes)[1]
        self.relu = nn.ReLU()
        self.conv2 = conv3x3(planes, planes, indice_key=indice_key, bias=bias)
        self.bn2 = build_norm_layer(norm_cfg, planes)[1]
        self.downsample = downsample
        self.stride = stride
        self.conv1.emit_bn_stats = self.conv2.emit_bn_stats = True   # bn1 / bn2 take their statistics from the conv epilogues

    def forward(self, x):
        out = self.conv1(x)
        out.features = self.bn1(out.features, relu=True)
        out = self.conv2(out)
        identity = x if self.downsample is None else self.downsample(x)
        out.features = self.bn2(out.features, residual=identity.features, relu=True)  # bn2 + identity, then ReLU
        return out


def build_geometry(coors, batch_size, shape, strided_specs, subm_keys):
    """Every rulebook of one backbone pass, from coordinates alone.  `strided_specs[i]` =
    (plan_key, ksize, stride, padding); `subm_keys[i]` = (indice_key, ksize) at the resolution
    before strided conv i.  Returns {indice_key: Rulebook, ("conv", plan_key): Rulebook}."""
    plan = {}
    specs3 = [(tuple(int(v) for v in k), tuple(int(v) for v in s), tuple(int(v) for v in p)) for _, k, s, p in strided_specs]
    if (coors.is_cuda and coors.shape[0] > 0 and all(sk is None or tuple(sk[1]) == (3, 3, 3) for sk in subm_keys)
            and H.rulebook_chain_supported(batch_size, shape, specs3)):
        # r04: one launch chain + one host read for the whole pass (csrc/rulebook_chain.hip)
        want = [i < len(subm_keys) and subm_keys[i] is not None for i in range(len(strided_specs) + 1)]
        subm, conv = H.build_rulebook_chain(coors, batch_size, shape, specs3, want)
        for i, rb in enumerate(subm):
            if rb is not None:
                plan[subm_keys[i][0]] = rb
        for (pk, _, _, _), rb in zip(strided_specs, conv):
            plan[("conv", pk)] = rb
        return plan
    for i in range(len(strided_specs) + 1):
        if i < len(subm_keys) and subm_keys[i] is not None:
            key, ksize = subm_keys[i]
            plan[key] = H.build_subm_rulebook(coors, batch_size, shape, ksize)
        if i == len(strided_specs):
            break
        pk, ksize, stride, padding = strided_specs[i]
        rb = H.build_conv_rulebook(coors, batch_size, shape, ksize, stride, padding)
        plan[("conv", pk)] = rb
        coors, shape = rb.out_coors, rb.out_shape
    return plan


def plan_geometry(x: SparseConvTensor, strided_specs, subm_keys):
    """Attach the geometry plan to the tensor's indice_dict — built before the first feature kernel, so that the
    host reads of N_out happen while the device is otherwise idle.  A plan that the data pipeline already built for these
    coordinates (data.attach_geometry: `indices._s2d_geometry`) is taken as is."""
    pre = getattr(x.indices, "_s2d_geometry", None)
    if pre is not None and pre[0] == tuple(int(s) for s in x.spatial_shape) and pre[1] == x.batch_size:
        x.indice_dict.update(pre[2])
        return
    x.indice_dict.update(build_geometry(x.indices, x.batch_size, x.spatial_shape, strided_specs, subm_keys))


class _PlannedBackbone(nn.Module):
    """shared geometry planning of the two sparse middle extractors"""
    SUBM_PREFIX = "res"

    def _strided_convs(self):
        raise NotImplementedError

    def _specs(self):
        convs = self._strided_convs()
        for i, c in enumerate(convs):
            c.plan_key = f"down{i}"
        strided = [(c.plan_key, c.kernel_size, c.stride, c.padding) for c in convs]
        subm = [(f"{self.SUBM_PREFIX}{i}", (3, 3, 3)) for i in range(4)]
        return strided, subm


@BACKBONES.register_module
class SpMiddleResNetFHD(_PlannedBackbone):
    def __init__(self, num_input_features=128, norm_cfg=None, name="SpMiddleResNetFHD", is_student=False, **kwargs):
        super().__init__()
        self.name = name
        self.dcn = None
        self.zero_init_residual = False
        self.is_student = is_student
        if norm_cfg is None:
            norm_cfg = dict(type="BN1d", eps=1e-3, momentum=0.01)

        def norm(c):
            return build_norm_layer(norm_cfg, c)[1]

        def block(c, key):
            return SparseBasicBlock(c, c, norm_cfg=norm_cfg, indice_key=key)

        self.conv_input = SparseSequential(
            SubMConv3d(num_input_features, 16, 3, bias=False, indice_key="res0"), norm(16), nn.ReLU(inplace=True))
        self.conv1 = SparseSequential(block(16, "res0"), block(16, "res0"))
        self.conv2 = SparseSequential(
            SparseConv3d(16, 32, 3, 2, padding=1, bias=False), norm(32), nn.ReLU(inplace=True),
            block(32, "res1"), block(32, "res1"))
        self.conv3 = SparseSequential(
            SparseConv3d(32, 64, 3, 2, padding=1, bias=False), norm(64), nn.ReLU(inplace=True),
            block(64, "res2"), block(64, "res2"))
        self.conv4 = SparseSequential(
            SparseConv3d(64, 128, 3, 2, padding=[0, 1, 1], bias=False), norm(128), nn.ReLU(inplace=True),
            block(128, "res3"), block(128, "res3"))
        self.extra_conv = SparseSequential(
            SparseConv3d(128, 128, (3, 1, 1), (2, 1, 1), bias=False), norm(128), nn.ReLU())

    def _strided_convs(self):
        return [self.conv2[0], self.conv3[0], self.conv4[0], self.extra_conv[0]]

    def forward(self, voxel_features, coors, batch_size, input_shape, bev_nhwc_bf16=False):
        sparse_shape = np.array(input_shape[::-1]) + [1, 0, 0]  # (z+1, y, x), scn.py:159
        ret = SparseConvTensor(voxel_features, coors if coors.dtype == torch.int32 else coors.int(), sparse_shape, batch_size)
        plan_geometry(ret, *self._specs())
        x = self.conv_input(ret)
        x_conv1 = self.conv1(x)
        x_conv2 = self.conv2(x_conv1)
        x_conv3 = self.conv3(x_conv2)
        x_conv4 = self.conv4(x_conv3)
        ret = self.extra_conv(x_conv4).dense_bev(nhwc_bf16=bev_nhwc_bf16)
        return ret, {"conv1": x_conv1, "conv2": x_conv2, "conv3": x_conv3, "conv4": x_conv4}


@BACKBONES.register_module
class SpMiddleFHD(_PlannedBackbone):
    SUBM_PREFIX = "subm"

    """SECOND's plain (non-residual) middle extractor."""

    def __init__(self, num_input_features=128, norm_cfg=None, name="SpMiddleFHD", **kwargs):
        super().__init__()
        self.name = name
        self.dcn = None
        self.zero_init_residual = False
        if norm_cfg is None:
            norm_cfg = dict(type="BN1d", eps=1e-3, momentum=0.01)
        layers = []

        def subm(ci, co, key):
            layers.extend([SubMConv3d(ci, co, 3, indice_key=key, bias=False), build_norm_layer(norm_cfg, co)[1], nn.ReLU()])

        def down(ci, co, pad):
            layers.extend([SparseConv3d(ci, co, 3, 2, padding=pad, bias=False), build_norm_layer(norm_cfg, co)[1], nn.ReLU()])

        subm(num_input_features, 16, "subm0"); subm(16, 16, "subm0")
        down(16, 32, 1); subm(32, 32, "subm1"); subm(32, 32, "subm1")
        down(32, 64, 1); subm(64, 64, "subm2"); subm(64, 64, "subm2"); subm(64, 64, "subm2")
        down(64, 64, [0, 1, 1]); subm(64, 64, "subm3"); subm(64, 64, "subm3"); subm(64, 64, "subm3")
        self.middle_conv = SparseSequential(*layers)
        self.extra_conv = SparseSequential(
            SparseConv3d(64, 64, (3, 1, 1), (2, 1, 1), bias=False), build_norm_layer(norm_cfg, 64)[1], nn.ReLU())

    def _strided_convs(self):
        return [m for m in self.middle_conv._modules.values() if isinstance(m, SparseConv3d)] + [self.extra_conv[0]]

    def forward(self, voxel_features, coors, batch_size, input_shape, bev_nhwc_bf16=False):
        sparse_shape = np.array(input_shape[::-1]) + [1, 0, 0]
        ret = SparseConvTensor(voxel_features, coors if coors.dtype == torch.int32 else coors.int(), sparse_shape, batch_size)
        plan_geometry(ret, *self._specs())
        conv_4 = self.middle_conv(ret)
        return self.extra_conv(conv_4).dense_bev(nhwc_bf16=bev_nhwc_bf16), conv_4
