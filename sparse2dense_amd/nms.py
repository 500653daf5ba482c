"""Rotated BEV NMS / IoU on the device (csrc/nms.hip): `box_torch_ops.rotate_nms_pcdet` and `iou3d_nms_cuda.boxes_iou_bev_gpu`
(/root/reference/det3d/core/bbox/box_torch_ops.py:449-464, det3d/ops/iou3d_nms/src/iou3d_nms_api.cpp:11-16)."""
import torch

from . import _lib


def _stream(dev):
    return torch._C._cuda_getCurrentRawStream(dev.index)


def boxes_iou_bev(boxes_a, boxes_b):
    """[N,7] x [M,7] (x,y,z,dx,dy,dz,heading) cuda fp32 -> IoU matrix [N,M]"""
    if not boxes_a.is_cuda:
        raise _lib.S2DError("boxes_iou_bev: CUDA tensors expected (no CPU fallback)")
    lib = _lib.load()
    a, b = boxes_a.float().contiguous(), boxes_b.float().contiguous()
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    _lib.check(lib.s2d_bev_iou_f32(a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], out.data_ptr(), _stream(a.device)), "s2d_bev_iou_f32")
    return out


def rotate_nms(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
    """indices into `boxes` kept by the greedy rotated NMS, in descending-score order (rotate_nms_pcdet).  The only host read is
    the number of kept boxes (the result tensor's size - the reference API exposes it the same way)."""
    if not boxes.is_cuda:
        raise _lib.S2DError("rotate_nms: CUDA tensors expected (no CPU fallback)")
    lib = _lib.load()
    order = torch.sort(scores, dim=0, descending=True, stable=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    n = int(order.shape[0])
    if n == 0:
        return order
    b = boxes[order].float().contiguous()
    keep = torch.empty(n, dtype=torch.int64, device=b.device)
    n_keep = torch.empty(1, dtype=torch.int32, device=b.device)
    ws = torch.empty(lib.s2d_nms_workspace_bytes(n), dtype=torch.uint8, device=b.device)
    max_keep = n if post_max_size is None else min(n, int(post_max_size))
    _lib.check(lib.s2d_nms_rotated_bev(b.data_ptr(), n, float(thresh), max_keep, keep.data_ptr(), n_keep.data_ptr(), ws.data_ptr(), ws.numel(),
                                       _stream(b.device)), "s2d_nms_rotated_bev")
    return order[keep[:int(n_keep.item())]]


def circle_nms(centers_xy, scores, min_radius, post_max_size=83):
    """CenterPoint's `_circle_nms` (/root/reference/det3d/models/bbox_heads/center_head.py:499-507 over core/utils/circle_nms_jit.py:4-31):
    indices into the input kept by the greedy centre-distance suppression, in descending-score order, at most post_max_size of them.
    (The reference orders equal scores by numpy's reversed argsort; here the sort is stable descending - float scores do not tie.)"""
    if not centers_xy.is_cuda:
        raise _lib.S2DError("circle_nms: CUDA tensors expected (no CPU fallback)")
    lib = _lib.load()
    order = torch.sort(scores, dim=0, descending=True, stable=True)[1]
    n = int(order.shape[0])
    if n == 0:
        return order
    xy = centers_xy[order].float().contiguous()
    keep = torch.empty(n, dtype=torch.int64, device=xy.device)
    n_keep = torch.empty(1, dtype=torch.int32, device=xy.device)
    ws = torch.empty(lib.s2d_nms_workspace_bytes(n), dtype=torch.uint8, device=xy.device)
    max_keep = n if post_max_size is None else min(n, int(post_max_size))
    _lib.check(lib.s2d_nms_circle(xy.data_ptr(), n, float(min_radius), max_keep, keep.data_ptr(), n_keep.data_ptr(), ws.data_ptr(), ws.numel(),
                                  _stream(xy.device)), "s2d_nms_circle")
    return order[keep[:int(n_keep.item())]]
