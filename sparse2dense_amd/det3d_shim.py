"""Makes the reference's own import paths resolve to this package, so its unmodified config files
and call sites load our implementation as a drop-in for the hot path:

    import sparse2dense_amd.det3d_shim as shim; shim.install()
    from det3d.torchie import Config                 # det3d/torchie/utils/config.py:77-100
    from det3d.models import build_detector          # det3d/models/builder.py:49
    cfg = Config.fromfile("configs/waymo/voxelnet/waymo_centerpoint_voxelnet_3x_distill_interval_5.py")
    student = build_detector(cfg.S_model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)

Provided module paths (everything else of det3d is out of scope and absent on purpose):
  det3d.models{,.registry,.builder}            registries + build_* (same keys)
  det3d.ops.point_cloud.point_cloud_ops        points_to_voxel
  det3d.core.input.voxel_generator             VoxelGenerator
  det3d.utils.config_tool                      get_downsample_factor  (config_tool.py:39-53; imported by configs)
  det3d.builder                                build_box_coder stub   (imported by the SECOND configs)
  det3d.torchie                                Config, ConfigDict
  spconv                                       SparseConvTensor, SubMConv3d, SparseConv3d, SparseSequential, SparseModule
"""
import importlib.util
import os
import sys
import types

import numpy as np


class ConfigDict(dict):
    """attribute-access dict (the reference uses addict.Dict, det3d/torchie/utils/config.py:12-33)"""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'ConfigDict' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        self[name] = value


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, (list, tuple)):
        return type(v)(_wrap(x) for x in v)
    return v


class Config:
    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, "_cfg_dict", _wrap(cfg_dict or {}))
        object.__setattr__(self, "_filename", filename)

    @staticmethod
    def fromfile(filename):
        filename = os.path.abspath(os.path.expanduser(filename))
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        install()
        spec = importlib.util.spec_from_file_location("_s2d_cfg_" + str(abs(hash(filename))), filename)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        cfg = {k: v for k, v in mod.__dict__.items() if not k.startswith("__") and not isinstance(v, types.ModuleType)
               and not callable(v)}
        return Config(cfg, filename)

    @property
    def filename(self):
        return self._filename

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, name, default=None):
        return self._cfg_dict.get(name, default)


def get_downsample_factor(model_config):
    """neck ds strides / last us stride * backbone ds_factor (det3d/utils/config_tool.py:39-53)."""
    try:
        neck_cfg = model_config["neck"]
    except Exception:
        model_config = model_config["first_stage_cfg"]
        neck_cfg = model_config["neck"]
    factor = np.prod(neck_cfg.get("ds_layer_strides", [1]))
    if len(neck_cfg.get("us_layer_strides", [])) > 0:
        factor /= neck_cfg.get("us_layer_strides", [])[-1]
    factor *= model_config["backbone"]["ds_factor"]
    factor = int(factor)
    assert factor > 0
    return factor


def _module(name, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        m.__path__ = []  # behave like a package so that dotted imports continue
        sys.modules[name] = m
        parent, _, leaf = name.rpartition(".")
        if parent:
            setattr(_module(parent), leaf, m)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


_installed = False


def install():
    """Idempotent.  Refuses to shadow a real `det3d` that is already imported."""
    global _installed
    if _installed:
        return
    if "det3d" in sys.modules and not getattr(sys.modules["det3d"], "__s2d_shim__", False):
        raise RuntimeError("a real det3d package is already imported; the shim would shadow it")
    from . import backbones, detectors, heads, necks, pillars, registry, spconv, voxel_ops  # noqa: F401 (registers keys)

    _module("det3d", __s2d_shim__=True)
    reg_attrs = {k: getattr(registry, k) for k in ["READERS", "BACKBONES", "NECKS", "HEADS", "LOSSES", "DETECTORS",
                                                   "SECOND_STAGE", "ROI_HEAD"]}
    build_attrs = {k: getattr(registry, k) for k in ["build_reader", "build_backbone", "build_neck", "build_head",
                                                     "build_loss", "build_detector", "build"]}
    _module("det3d.models", **reg_attrs, **build_attrs)
    _module("det3d.models.registry", **reg_attrs)
    _module("det3d.models.builder", **build_attrs)
    _module("det3d.utils", Registry=registry.Registry, build_from_cfg=registry.build_from_cfg)
    _module("det3d.utils.registry", Registry=registry.Registry, build_from_cfg=registry.build_from_cfg)
    _module("det3d.utils.config_tool", get_downsample_factor=get_downsample_factor)
    _module("det3d.builder", build_box_coder=lambda cfg, **kw: ConfigDict(
        dict(cfg, code_size=cfg.get("n_dim", 7) + (1 if cfg.get("encode_angle_vector", False) else 0))))
    _module("det3d.torchie", Config=Config, ConfigDict=ConfigDict)
    _module("det3d.ops")
    _module("det3d.ops.point_cloud")
    _module("det3d.ops.point_cloud.point_cloud_ops", points_to_voxel=voxel_ops.points_to_voxel)
    _module("det3d.core")
    _module("det3d.core.input")
    _module("det3d.core.input.voxel_generator", VoxelGenerator=voxel_ops.VoxelGenerator)
    if "spconv" not in sys.modules:
        _module("spconv", SparseConvTensor=spconv.SparseConvTensor, SubMConv3d=spconv.SubMConv3d,
                SparseConv3d=spconv.SparseConv3d, SparseSequential=spconv.SparseSequential,
                SparseModule=spconv.SparseModule)
    _installed = True
