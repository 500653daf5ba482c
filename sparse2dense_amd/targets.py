"""Device-side CenterPoint target assignment: `AssignLabel`
(/root/reference/det3d/datasets/pipelines/preprocess.py:478-653) for the one-task Waymo head, produced on the GPU from padded
ground-truth boxes so that no per-frame numpy work is left on the host.  Output fields, dtypes and shapes are those of the
reference's `example` after `collate_kitti` (lists with one entry per task): hm f32[B,3,H,W], anno_box f32[B,500,10],
ind i64[B,500], mask u8[B,500], cat i64[B,500] (+ gt_boxes_and_cls f32[B,500,10] for the two-stage code).
`scene.assign_targets` is the CPU restatement the tests compare against."""
import ctypes

import numpy as np
import torch

from . import _lib, scene


def pad_boxes(boxes_list, classes_list, device):
    """per-frame numpy boxes [K_b,9] / classes [K_b] -> padded device tensors f32[B,Kmax,9], i32[B,Kmax] (0 = padding)"""
    kmax = max([len(b) for b in boxes_list] + [1])
    boxes = np.zeros((len(boxes_list), kmax, 9), np.float32)
    classes = np.zeros((len(boxes_list), kmax), np.int32)
    for i, (b, c) in enumerate(zip(boxes_list, classes_list)):
        boxes[i, :len(b)] = b
        classes[i, :len(b)] = c
    return torch.from_numpy(boxes).to(device), torch.from_numpy(classes).to(device)


def assign_label(gt_boxes, gt_classes, pc_range=scene.WAYMO_RANGE, voxel_size=scene.WAYMO_VOXEL, out_size_factor=8, num_classes=3,
                 max_objs=500, gaussian_overlap=0.1, min_radius=2, grid_xy=(1504, 1504), with_boxes_and_cls=False):
    """gt_boxes f32[B,K,9] cuda, gt_classes i32[B,K] cuda -> dict of per-task lists (one task)."""
    if not gt_boxes.is_cuda:
        raise _lib.S2DError("assign_label: CUDA tensors expected (scene.assign_targets is the host restatement)")
    lib = _lib.load()
    gt_boxes = gt_boxes.float().contiguous()
    gt_classes = gt_classes.int().contiguous()
    b, k = gt_classes.shape
    fw, fh = grid_xy[0] // out_size_factor, grid_xy[1] // out_size_factor
    dev = gt_boxes.device
    hm = torch.zeros((b, num_classes, fh, fw), dtype=torch.float32, device=dev)
    anno = torch.empty((b, max_objs, 10), dtype=torch.float32, device=dev)
    ind = torch.empty((b, max_objs), dtype=torch.int64, device=dev)
    mask = torch.empty((b, max_objs), dtype=torch.uint8, device=dev)
    cat = torch.empty((b, max_objs), dtype=torch.int64, device=dev)
    bc = torch.empty((b, max_objs, 10), dtype=torch.float32, device=dev) if with_boxes_and_cls else None
    f2 = lambda v: (ctypes.c_float * 2)(float(np.float32(v[0])), float(np.float32(v[1])))
    p = lambda t: None if t is None else t.data_ptr()
    _lib.check(lib.s2d_assign_label(p(gt_boxes), p(gt_classes), b, k, f2(pc_range[:2]), f2(voxel_size[:2]), int(out_size_factor), fw, fh,
                                    int(num_classes), int(max_objs), float(gaussian_overlap), int(min_radius), p(hm), p(anno), p(ind), p(mask),
                                    p(cat), p(bc), torch._C._cuda_getCurrentRawStream(dev.index)), "s2d_assign_label")
    out = dict(hm=[hm], anno_box=[anno], ind=[ind], mask=[mask], cat=[cat])
    if with_boxes_and_cls:
        out["gt_boxes_and_cls"] = bc
    return out
