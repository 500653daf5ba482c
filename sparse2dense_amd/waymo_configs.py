"""Model dictionaries of the BASELINE.json configurations, with the same keys and values as the
reference config files (the drop-in contract is that those files load unchanged through
det3d_shim.Config.fromfile; these literals exist so that bench.py and the GPU tests do not need
/root/reference at run time):

  centerpoint_voxelnet()   configs/waymo/voxelnet/waymo_centerpoint_voxelnet_3x_distill_interval_5.py:18-46  (`model`, teacher / plain)
  s2d_student()            same file :48-76 (`S_model`)
  second_voxelnet_parts()  configs/waymo/voxelnet/waymo_second_3x_interval_5.py (reader/backbone/neck of config 1)
"""
import logging

TASKS = [dict(num_class=3, class_names=["VEHICLE", "PEDESTRIAN", "CYCLIST"])]


def _head():
    return dict(type="CenterHead", in_channels=sum([256, 256]), tasks=TASKS, dataset="waymo", weight=2,
                code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0],
                common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2)})


def _neck(kind):
    return dict(type=kind, layer_nums=[5, 5], ds_layer_strides=[1, 2], ds_num_filters=[128, 256],
                us_layer_strides=[1, 2], us_num_filters=[256, 256], num_input_features=256,
                logger=logging.getLogger(kind))


def centerpoint_voxelnet():
    return dict(type="VoxelNet", pretrained=None,
                reader=dict(type="VoxelFeatureExtractorV3", num_input_features=5),
                backbone=dict(type="SpMiddleResNetFHD", num_input_features=5, ds_factor=8),
                neck=_neck("RPN"), bbox_head=_head())


def s2d_student():
    return dict(type="KD_VoxelNet", pretrained=None,
                reader=dict(type="VoxelFeatureExtractorV3", num_input_features=5),
                backbone=dict(type="SpMiddleResNetFHD", num_input_features=5, ds_factor=8),
                neck=_neck("S2D_RPN"), bbox_head=_head())


def second_voxelnet():
    """configs/waymo/voxelnet/waymo_second_3x_interval_5.py:58-104 (BASELINE config 1); loss dictionaries of the
    anchor head are omitted (forward only)."""
    parts = second_voxelnet_parts()
    return dict(type="VoxelNet", pretrained=None, **parts,
                bbox_head=dict(type="MultiGroupHead", mode="3d", in_channels=sum([128, ]), tasks=TASKS, weights=[1, ],
                               box_coder=dict(type="ground_box3d_coder", n_dim=7, linear_dim=False,
                                              encode_angle_vector=False, code_size=7),
                               encode_background_as_zeros=True, use_sigmoid_score=True, encode_rad_error_by_sin=True,
                               loss_aux=dict(type="WeightedSoftmaxClassificationLoss", name="direction_classifier",
                                             loss_weight=0.2), direction_offset=0.0))


def second_voxelnet_parts():
    return dict(reader=dict(type="VoxelFeatureExtractorV3", num_input_features=5),
                backbone=dict(type="SpMiddleFHD", num_input_features=5, ds_factor=8),
                neck=dict(type="RPN", layer_nums=[5], ds_layer_strides=[1], ds_num_filters=[128], us_layer_strides=[1],
                          us_num_filters=[128], num_input_features=128, logger=logging.getLogger("RPN")))


def _pp_reader():
    return dict(type="PillarFeatureNet", num_filters=[64, 64], num_input_features=5, with_distance=False,
                voxel_size=(0.32, 0.32, 6.0), pc_range=(-74.88, -74.88, -2, 74.88, 74.88, 4.0))


def _pp_neck():
    return dict(type="RPN", layer_nums=[3, 5, 5], ds_layer_strides=[1, 2, 2], ds_num_filters=[64, 128, 256],
                us_layer_strides=[1, 2, 4], us_num_filters=[128, 128, 128], num_input_features=64,
                logger=logging.getLogger("RPN"))


def _pp_head():
    h = _head()
    h["in_channels"] = 128 * 3
    return h


def centerpoint_pillar():
    """configs/waymo/pp/waymo_centerpoint_pp_two_pfn_stride1_3x_distill_interval_5.py:18-51 (`model`)"""
    return dict(type="PointPillars", pretrained=None, reader=_pp_reader(),
                backbone=dict(type="PointPillarsScatter", ds_factor=1), neck=_pp_neck(), bbox_head=_pp_head())


def pillar_s2d_student():
    """same file :55-88 (`S_model`) — BASELINE config 5"""
    return dict(type="KD_PointPillars", pretrained=None, reader=_pp_reader(),
                backbone=dict(type="PointPillarsScatter_S2D", ds_factor=1), neck=_pp_neck(), bbox_head=_pp_head())
