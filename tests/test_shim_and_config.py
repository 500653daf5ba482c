"""Drop-in boundary on the host side: the reference's unmodified config files import through the
det3d shim and `build_detector` returns our modules under the same registry keys with the
reference's parameter names.  (Skipped where /root/reference does not exist, e.g. the GPU box.)"""
import os

import pytest
import torch

import sparse2dense_amd.det3d_shim as shim
from sparse2dense_amd import waymo_configs

REF = "/root/reference"
need_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


def test_registry_keys_present():
    shim.install()
    from det3d.models import BACKBONES, DETECTORS, HEADS, NECKS, READERS
    assert {"SpMiddleResNetFHD", "SpMiddleFHD"} <= set(BACKBONES.module_dict)
    assert {"RPN", "S2D_RPN"} <= set(NECKS.module_dict)
    assert "CenterHead" in HEADS.module_dict and "VoxelFeatureExtractorV3" in READERS.module_dict
    assert {"VoxelNet", "KD_VoxelNet"} <= set(DETECTORS.module_dict)
    import spconv
    from det3d.core.input.voxel_generator import VoxelGenerator
    from det3d.ops.point_cloud.point_cloud_ops import points_to_voxel
    assert callable(points_to_voxel) and hasattr(spconv, "SubMConv3d")
    vg = VoxelGenerator([0.1, 0.1, 0.15], [-75.2, -75.2, -2, 75.2, 75.2, 4], 5, 150000)
    assert list(vg.grid_size) == [1504, 1504, 40]


@need_ref
@pytest.mark.parametrize("rel", [
    "configs/waymo/voxelnet/waymo_centerpoint_voxelnet_3x_distill_interval_5.py",
    "configs/waymo/voxelnet/waymo_centerpoint_voxelnet_3x_interval_5.py",
])
def test_reference_config_loads_and_builds(rel):
    shim.install()
    from det3d.models import build_detector
    from det3d.torchie import Config
    cfg = Config.fromfile(os.path.join(REF, rel))
    assert cfg.assigner.out_size_factor == 8
    mcfg = cfg.model if "model" in cfg else cfg.S_model   # plain configs only define S_model (tools/train.py:121)
    model = build_detector(mcfg, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    keys = set(model.state_dict().keys())
    for k in ["backbone.conv_input.0.weight", "backbone.conv_input.1.running_mean", "backbone.conv1.0.conv1.bias",
              "backbone.conv2.0.weight", "backbone.conv4.4.bn2.weight", "backbone.extra_conv.0.weight",
              "neck.blocks.0.1.weight", "neck.deblocks.1.0.weight", "bbox_head.shared_conv.0.weight",
              "bbox_head.tasks.0.hm.3.bias"]:
        assert k in keys, k
    sd = model.state_dict()
    assert tuple(sd["backbone.conv_input.0.weight"].shape) == (3, 3, 3, 5, 16)     # [kD,kH,kW,Cin,Cout]
    assert tuple(sd["backbone.extra_conv.0.weight"].shape) == (3, 1, 1, 128, 128)
    assert "backbone.conv2.0.bias" not in keys and "backbone.conv1.1.conv2.bias" in keys
    if cfg.get("distillation", False):
        student = build_detector(cfg.S_model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
        skeys = set(student.state_dict().keys())
        assert "neck.encoder_1.0.weight" in skeys and "neck.generator_2.3.weight" in skeys
        n_params = sum(p.numel() for p in student.neck.parameters())
        assert abs(n_params - 15.27e6) < 0.05e6   # S2D_RPN 15.27 M params
        # same dictionaries as the in-repo literals used by bench.py
        lit = waymo_configs.s2d_student()
        for part in ["reader", "backbone", "bbox_head"]:
            a = {k: v for k, v in dict(cfg.S_model[part]).items() if k != "logger"}
            b = {k: v for k, v in lit[part].items() if k != "logger"}
            assert a == b, part


@need_ref
def test_second_config_file_imports_and_backbone_builds():
    """BASELINE config 1 (SECOND): the file imports det3d.builder.build_box_coder at import time;
    reader/backbone/neck are on the path, the anchor head and its targets are out of scope."""
    shim.install()
    from det3d.models import build_backbone, build_neck
    from det3d.torchie import Config
    cfg = Config.fromfile(os.path.join(REF, "configs/waymo/voxelnet/waymo_second_3x_interval_5.py"))
    bb = build_backbone(cfg.S_model.backbone)
    assert type(bb).__name__ == "SpMiddleFHD"
    assert "middle_conv.0.weight" in bb.state_dict() and "extra_conv.1.running_var" in bb.state_dict()
    neck = build_neck(cfg.S_model.neck)
    assert type(neck).__name__ == "RPN"
    from det3d.models import build_detector
    det = build_detector(cfg.S_model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    sd = det.state_dict()
    assert tuple(sd["bbox_head.tasks.0.conv_box.weight"].shape) == (42, 128, 1, 1)     # 6 anchors x 7 (mg_head.py:460-478)
    assert tuple(sd["bbox_head.tasks.0.conv_cls.weight"].shape) == (18, 128, 1, 1)
    assert tuple(sd["bbox_head.tasks.0.conv_dir.weight"].shape) == (12, 128, 1, 1)


@need_ref
def test_pointpillars_config_file_loads_and_builds():
    """BASELINE config 5: configs/waymo/pp/...distill...: teacher PointPillars + student KD_PointPillars."""
    shim.install()
    from det3d.models import build_detector
    from det3d.torchie import Config
    cfg = Config.fromfile(os.path.join(REF, "configs/waymo/pp/waymo_centerpoint_pp_two_pfn_stride1_3x_distill_interval_5.py"))
    t = build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    s = build_detector(cfg.S_model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    assert type(t).__name__ == "PointPillars" and type(s).__name__ == "KD_PointPillars"
    for k in ["reader.pfn_layers.0.linear.weight", "reader.pfn_layers.1.norm.running_var", "neck.blocks.2.1.weight",
              "bbox_head.shared_conv.0.weight"]:
        assert k in t.state_dict(), k
    assert "backbone.convnext_block_3.1.weight" in s.state_dict() and "backbone.gen_mask.3.bias" in s.state_dict()
    assert tuple(s.state_dict()["backbone.convnext_block_1.1.weight"].shape) == (256, 59, 59)
    lit = waymo_configs.pillar_s2d_student()
    assert {k: v for k, v in dict(cfg.S_model.reader).items()} == lit["reader"]
    assert cfg.assigner.out_size_factor == 1


def test_param_counts_of_literal_configs():
    shim.install()
    from sparse2dense_amd.registry import build_detector
    m = build_detector(waymo_configs.centerpoint_voxelnet())
    nb = sum(p.numel() for p in m.backbone.parameters())
    nh = sum(p.numel() for p in m.bbox_head.parameters())
    assert abs(nb - 2.7e6) < 0.1e6 and abs(nh - 0.49e6) < 0.02e6


def test_hot_path_has_no_cpu_fallback():
    shim.install()
    from sparse2dense_amd.registry import build_backbone
    bb = build_backbone(dict(type="SpMiddleResNetFHD", num_input_features=5))
    with pytest.raises(Exception):
        bb(torch.zeros(4, 5), torch.zeros(4, 4, dtype=torch.int32), 1, [1504, 1504, 40])
