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


def second_voxelnet_parts():
    return dict(reader=dict(type="VoxelFeatureExtractorV3", num_input_features=5),
                backbone=dict(type="SpMiddleFHD", num_input_features=5, ds_factor=8),
                neck=dict(type="RPN", layer_nums=[5], ds_layer_strides=[1], ds_num_filters=[128], us_layer_strides=[1],
                          us_num_filters=[128], num_input_features=128, logger=logging.getLogger("RPN")))
