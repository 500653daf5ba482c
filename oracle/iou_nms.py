"""ORACLE (test infrastructure only) - ctypes front-end of oracle/iou_nms.c (rotated BEV IoU + greedy NMS, see its header for
the reference lines it follows) and `rotate_nms_pcdet` / `CenterHead.post_processing` restated with numpy
(/root/reference/det3d/core/bbox/box_torch_ops.py:449-464, det3d/models/bbox_heads/center_head.py:452-495)."""
import ctypes
import os

import numpy as np

from . import voxelize as _v


def _lib():
    lib = _v._lib()
    if not hasattr(lib, "_iou_ready"):
        lib.s2d_oracle_bev_iou_matrix.restype = None
        lib.s2d_oracle_bev_iou_matrix.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        lib.s2d_oracle_circle_nms.restype = ctypes.c_int
        lib.s2d_oracle_circle_nms.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
        lib.s2d_oracle_nms.restype = ctypes.c_int
        lib.s2d_oracle_nms.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
        lib._iou_ready = True
    return lib


def bev_iou(a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    if a.shape[0] and b.shape[0]:
        _lib().s2d_oracle_bev_iou_matrix(a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], out.ctypes.data)
    return out


def rotate_nms(boxes7, scores, thresh, pre_maxsize=None, post_max_size=None):
    """indices (into the unsorted input) kept by rotate_nms_pcdet"""
    boxes7 = np.ascontiguousarray(boxes7, np.float32)
    order = np.argsort(-np.asarray(scores, np.float32), kind="stable")
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    b = np.ascontiguousarray(boxes7[order])
    keep = np.zeros((max(len(order), 1),), np.int64)
    n = _lib().s2d_oracle_nms(b.ctypes.data, len(order), float(thresh), len(order), keep.ctypes.data) if len(order) else 0
    sel = order[keep[:n]]
    return sel[:post_max_size] if post_max_size is not None else sel


def circle_nms(centers_xy, scores, min_radius, post_max_size=83):
    """indices (into the unsorted input) kept by CenterPoint's circle NMS (center_head.py:499-507, circle_nms_jit.py:4-31)"""
    xy = np.ascontiguousarray(centers_xy, np.float32)
    order = np.argsort(-np.asarray(scores, np.float32), kind="stable")
    b = np.ascontiguousarray(xy[order])
    keep = np.zeros((max(len(order), 1),), np.int64)
    n = _lib().s2d_oracle_circle_nms(b.ctypes.data, len(order), float(min_radius), keep.ctypes.data) if len(order) else 0
    sel = order[keep[:n]]
    return sel[:post_max_size] if post_max_size is not None else sel
