"""Two-stage CenterPoint head: what sits on top of the hot path's first stage in the reference's
`two_stage` configs (first stage frozen, configs/waymo/voxelnet/two_stage/*: num_point=5, NMS_POST_MAXSIZE=500).

  DETECTORS["TwoStageDetector"]        /root/reference/det3d/models/detectors/two_stage.py:8-199
  SECOND_STAGE["BEVFeatureExtractor"]  /root/reference/det3d/models/second_stage/bird_eye_view.py:9-41
                                       (bilinear_interpolate_torch, det3d/core/utils/center_utils.py:93-122)
  ROI_HEAD["RoIHead"]                  /root/reference/det3d/models/roi_heads/roi_head.py:16-106,
                                       roi_head_template.py:27-41,153-183 (make_fc_layers, generate_predicted_boxes)
  VoxelNet/KD_VoxelNet.forward_two_stage   /root/reference/det3d/models/detectors/voxelnet.py:107-141,266-301

  ProposalTargetLayer                  /root/reference/det3d/models/roi_heads/target_assigner/proposal_target_layer.py:14-237
  RoIHead.assign_targets / losses      /root/reference/det3d/models/roi_heads/roi_head_template.py:43-151
  boxes_iou3d                          /root/reference/det3d/ops/iou3d_nms/iou3d_nms_utils.py:28-70

Module / parameter names follow the reference (`single_det.*`, `roi_head.shared_fc_layer.0.weight` ...), so two-stage
checkpoints load through checkpoint.load_state_dict.  Inference path (`return_loss=False`): first-stage decode + rotated NMS
(heads.CenterHead.predict on the HIP kernels) -> 5 BEV feature samples per box -> RoI MLP -> refined boxes and scores.
Training path (`return_loss=True`, two_stage.py:154-199): the same first stage with its loss, then the RoIs are matched to the
ground truth by 3-D IoU (rotated BEV overlap from csrc/nms.hip x height overlap), sampled into ROI_PER_IMAGE foreground /
hard- / easy-background boxes with the reference's numpy / torch random draws (same seeds -> same samples), residual targets are
encoded in each RoI's frame, and the RoI MLP is trained with the IoU-scaled BCE + masked L1 losses."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import registry
from .registry import DETECTORS, ROI_HEAD, SECOND_STAGE


def bilinear_interpolate(im, x, y):
    """im [H, W, C]; x, y [N] in feature-map cells -> [N, C] (center_utils.py:93-122: neighbours clamped to the map)"""
    x0 = torch.floor(x).long()
    y0 = torch.floor(y).long()
    x1, y1 = x0 + 1, y0 + 1
    x0c, x1c = x0.clamp(0, im.shape[1] - 1), x1.clamp(0, im.shape[1] - 1)
    y0c, y1c = y0.clamp(0, im.shape[0] - 1), y1.clamp(0, im.shape[0] - 1)
    wa = (x1c.type_as(x) - x) * (y1c.type_as(y) - y)
    wb = (x1c.type_as(x) - x) * (y - y0c.type_as(y))
    wc = (x - x0c.type_as(x)) * (y1c.type_as(y) - y)
    wd = (x - x0c.type_as(x)) * (y - y0c.type_as(y))
    return (im[y0c, x0c] * wa[:, None] + im[y1c, x0c] * wb[:, None] + im[y0c, x1c] * wc[:, None] + im[y1c, x1c] * wd[:, None])


@SECOND_STAGE.register_module
class BEVFeatureExtractor(nn.Module):
    def __init__(self, pc_start, voxel_size, out_stride):
        super().__init__()
        self.pc_start, self.voxel_size, self.out_stride = pc_start, voxel_size, out_stride

    def absl_to_relative(self, absolute):
        a1 = (absolute[..., 0] - self.pc_start[0]) / self.voxel_size[0] / self.out_stride
        a2 = (absolute[..., 1] - self.pc_start[1]) / self.voxel_size[1] / self.out_stride
        return a1, a2

    def forward(self, example, batch_centers, num_point):
        ret = []
        for bev, centers in zip(example["bev_feature"], batch_centers):
            xs, ys = self.absl_to_relative(centers)
            feat = bilinear_interpolate(bev, xs, ys)
            if num_point > 1:   # the num_point sample sets are stacked along dim 0: concatenate them per box
                sec = len(feat) // num_point
                feat = torch.cat([feat[i * sec:(i + 1) * sec] for i in range(num_point)], dim=1)
            ret.append(feat)
        return ret


def _cfg_get(cfg, key, *default):
    if isinstance(cfg, dict):
        return cfg.get(key, *default) if default else cfg[key]
    return getattr(cfg, key, *default)


def limit_period(val, offset=0.5, period=np.pi):
    return val - torch.floor(val / period + offset) * period


def rotate_points_along_z(points, angle):
    """points [B, N, 3+C] as ROW vectors times [[c, -s, 0], [s, c, 0], [0, 0, 1]] per batch entry (box_torch_ops.py:326-344)"""
    cosa, sina = torch.cos(angle), torch.sin(angle)
    x = points[:, :, 0] * cosa[:, None] + points[:, :, 1] * sina[:, None]
    y = -points[:, :, 0] * sina[:, None] + points[:, :, 1] * cosa[:, None]
    return torch.cat([x[..., None], y[..., None], points[:, :, 2:]], dim=-1)


def _to_pcdet(boxes):
    """(x, y, z, w, l, h, yaw) -> OpenPCDet's (x, y, z, dx, dy, dz, heading) (iou3d_nms_utils.py:22-26)"""
    b = boxes[:, [0, 1, 2, 4, 3, 5, -1]].clone()
    b[:, -1] = -b[:, -1] - np.pi / 2
    return b


def boxes_iou3d(boxes_a, boxes_b, bev_iou=None):
    """3-D IoU matrix [N, M] of (x, y, z, w, l, h, yaw) boxes (iou3d_nms_utils.boxes_iou3d_gpu): rotated BEV overlap x height overlap
    over the union of the volumes.  The BEV overlap comes from the device IoU kernel (IoU = o / (A + B - o) -> o = IoU (A + B) /
    (1 + IoU)); `bev_iou` (tests: the CPU oracle) replaces it, otherwise CUDA tensors are required - no CPU fallback."""
    a, b = _to_pcdet(boxes_a), _to_pcdet(boxes_b)
    if bev_iou is None:
        from . import nms
        bev_iou = nms.boxes_iou_bev
    iou = bev_iou(a.contiguous(), b.contiguous()).to(a.dtype)
    area_a, area_b = (a[:, 3] * a[:, 4]).view(-1, 1), (b[:, 3] * b[:, 4]).view(1, -1)
    overlaps_bev = iou * (area_a + area_b) / (1.0 + iou)
    a_max, a_min = (a[:, 2] + a[:, 5] / 2).view(-1, 1), (a[:, 2] - a[:, 5] / 2).view(-1, 1)
    b_max, b_min = (b[:, 2] + b[:, 5] / 2).view(1, -1), (b[:, 2] - b[:, 5] / 2).view(1, -1)
    overlaps_h = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a, vol_b = (a[:, 3] * a[:, 4] * a[:, 5]).view(-1, 1), (b[:, 3] * b[:, 4] * b[:, 5]).view(1, -1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)


class ProposalTargetLayer(nn.Module):
    """IoU-based sampling of the first stage's RoIs and their classification / regression labels
    (proposal_target_layer.py).  The random draws are the reference's (np.random.permutation / np.random.rand on the host,
    torch.randint on the CPU generator), so a seeded run samples the same RoIs."""

    def __init__(self, roi_sampler_cfg, iou_fn=None):
        super().__init__()
        self.cfg = roi_sampler_cfg
        self.iou_fn = iou_fn or boxes_iou3d

    def _c(self, key, default=None):
        return _cfg_get(self.cfg, key, default)

    def forward(self, batch_dict):
        rois, gt_of_rois, ious, scores, labels, feats = self.sample_rois_for_rcnn(batch_dict)
        reg_valid_mask = (ious > self._c("REG_FG_THRESH")).long()
        kind = self._c("CLS_SCORE_TYPE")
        if kind == "cls":
            cls_labels = (ious > self._c("CLS_FG_THRESH")).long()
            ignore = (ious > self._c("CLS_BG_THRESH")) & (ious < self._c("CLS_FG_THRESH"))
            cls_labels[ignore > 0] = -1
        elif kind == "roi_iou":
            bg, fg = self._c("CLS_BG_THRESH"), self._c("CLS_FG_THRESH")
            fg_mask, bg_mask = ious > fg, ious < bg
            interval = (fg_mask == 0) & (bg_mask == 0)
            cls_labels = (fg_mask > 0).float()
            cls_labels[interval] = (ious[interval] - bg) / (fg - bg)
        else:
            raise NotImplementedError(kind)
        return dict(rois=rois, gt_of_rois=gt_of_rois, gt_iou_of_rois=ious, roi_scores=scores, roi_labels=labels, roi_features=feats,
                    reg_valid_mask=reg_valid_mask, rcnn_cls_labels=cls_labels)

    def sample_rois_for_rcnn(self, batch_dict):
        bs, per = batch_dict["batch_size"], self._c("ROI_PER_IMAGE")
        rois, roi_scores, roi_labels = batch_dict["rois"], batch_dict["roi_scores"], batch_dict["roi_labels"]
        gt_boxes, roi_features = batch_dict["gt_boxes_and_cls"], batch_dict["roi_features"]
        code = rois.shape[-1]
        out_rois, out_gt = rois.new_zeros(bs, per, code), rois.new_zeros(bs, per, code + 1)
        out_iou, out_scores = rois.new_zeros(bs, per), rois.new_zeros(bs, per)
        out_labels = rois.new_zeros((bs, per), dtype=torch.long)
        out_feats = roi_features.new_zeros(bs, per, roi_features.shape[-1])
        for i in range(bs):
            cur_gt = gt_boxes[i]
            k = len(cur_gt) - 1
            rowsum = cur_gt.sum(-1).tolist()                    # the reference walks back over the zero padding rows (sum == 0)
            while k > 0 and rowsum[k] == 0:
                k -= 1
            cur_gt = cur_gt[:k + 1]
            cur_gt = cur_gt.new_zeros((1, cur_gt.shape[1])) if len(cur_gt) == 0 else cur_gt
            if self._c("SAMPLE_ROI_BY_EACH_CLASS", False):
                max_overlaps, assignment = self.get_max_iou_with_same_class(rois[i][:, :7], roi_labels[i], cur_gt[:, 0:7], cur_gt[:, -1].long())
            else:
                max_overlaps, assignment = torch.max(self.iou_fn(rois[i], cur_gt[:, 0:7]), dim=1)
            idx = self.subsample_rois(max_overlaps)
            out_rois[i], out_labels[i], out_iou[i], out_scores[i] = rois[i][idx], roi_labels[i][idx], max_overlaps[idx], roi_scores[i][idx]
            out_gt[i], out_feats[i] = cur_gt[assignment[idx]], roi_features[i][idx]
        return out_rois, out_gt, out_iou, out_scores, out_labels, out_feats

    def subsample_rois(self, max_overlaps):
        per = self._c("ROI_PER_IMAGE")
        fg_per = int(np.round(self._c("FG_RATIO") * per))
        fg_thresh = min(self._c("REG_FG_THRESH"), self._c("CLS_FG_THRESH"))
        fg = (max_overlaps >= fg_thresh).nonzero().view(-1)
        easy = (max_overlaps < self._c("CLS_BG_THRESH_LO")).nonzero().view(-1)
        hard = ((max_overlaps < self._c("REG_FG_THRESH")) & (max_overlaps >= self._c("CLS_BG_THRESH_LO"))).nonzero().view(-1)
        n_fg, n_bg = fg.numel(), hard.numel() + easy.numel()
        if n_fg > 0 and n_bg > 0:
            take = min(fg_per, n_fg)
            perm = torch.from_numpy(np.random.permutation(n_fg)).to(max_overlaps.device).long()
            fg = fg[perm[:take]]
            bg = self.sample_bg_inds(hard, easy, per - take, self._c("HARD_BG_RATIO"))
        elif n_fg > 0:
            draw = torch.from_numpy(np.floor(np.random.rand(per) * n_fg)).to(max_overlaps.device).long()
            fg = fg[draw]
            bg = fg.new_zeros(0)
        elif n_bg > 0:
            bg = self.sample_bg_inds(hard, easy, per, self._c("HARD_BG_RATIO"))
        else:
            raise NotImplementedError(f"ProposalTargetLayer: no RoI to sample (max IoU in [{float(max_overlaps.min())}, {float(max_overlaps.max())}])")
        return torch.cat((fg, bg), dim=0)

    @staticmethod
    def sample_bg_inds(hard, easy, count, hard_ratio):
        draw = lambda n, k: torch.randint(low=0, high=n, size=(k,)).long()    # CPU generator, as in the reference
        if hard.numel() > 0 and easy.numel() > 0:
            n_hard = min(int(count * hard_ratio), len(hard))
            h = hard[draw(hard.numel(), n_hard).to(hard.device)]
            e = easy[draw(easy.numel(), count - n_hard).to(easy.device)]
            return torch.cat([h, e], dim=0)
        if hard.numel() > 0:
            return hard[draw(hard.numel(), count).to(hard.device)]
        if easy.numel() > 0:
            return easy[draw(easy.numel(), count).to(easy.device)]
        raise NotImplementedError

    def get_max_iou_with_same_class(self, rois, roi_labels, gt_boxes, gt_labels):
        max_overlaps = rois.new_zeros(rois.shape[0])
        assignment = roi_labels.new_zeros(roi_labels.shape[0])
        for k in range(int(gt_labels.min()), int(gt_labels.max()) + 1):
            rm, gm = roi_labels == k, gt_labels == k
            if rm.sum() > 0 and gm.sum() > 0:
                orig = gm.nonzero().view(-1)
                cur_max, cur_arg = torch.max(self.iou_fn(rois[rm], gt_boxes[gm]), dim=1)
                max_overlaps[rm] = cur_max
                assignment[rm] = orig[cur_arg]
        return max_overlaps, assignment


@ROI_HEAD.register_module
class RoIHead(nn.Module):
    def __init__(self, input_channels, model_cfg, num_class=1, code_size=7, test_cfg=None):
        super().__init__()
        self.model_cfg, self.num_class, self.code_size, self.test_cfg = model_cfg, num_class, code_size, test_cfg
        dp = _cfg_get(model_cfg, "DP_RATIO")
        shared, pre = [], input_channels
        fcs = list(_cfg_get(model_cfg, "SHARED_FC"))
        for k, c in enumerate(fcs):
            shared += [nn.Conv1d(pre, c, kernel_size=1, bias=False), nn.BatchNorm1d(c), nn.ReLU()]
            pre = c
            if k != len(fcs) - 1 and dp > 0:
                shared.append(nn.Dropout(dp))
        self.shared_fc_layer = nn.Sequential(*shared)
        self.cls_layers = self.make_fc_layers(pre, self.num_class, list(_cfg_get(model_cfg, "CLS_FC")), dp)
        self.reg_layers = self.make_fc_layers(pre, code_size, list(_cfg_get(model_cfg, "REG_FC")), dp)
        for m in self.modules():   # init_weights('xavier')
            if isinstance(m, (nn.Conv2d, nn.Conv1d)):
                nn.init.xavier_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.normal_(self.reg_layers[-1].weight, mean=0, std=0.001)
        target_cfg = _cfg_get(model_cfg, "TARGET_CONFIG", None)
        self.proposal_target_layer = ProposalTargetLayer(target_cfg) if target_cfg else None   # no parameters: state_dict unchanged
        self.forward_ret_dict = None

    def assign_targets(self, batch_dict):
        """sampled RoIs + their ground-truth boxes encoded in the RoI's frame (roi_head_template.py:43-92)"""
        if self.proposal_target_layer is None:
            raise ValueError("RoIHead: model_cfg.TARGET_CONFIG is required for training")
        bs = batch_dict["batch_size"]
        with torch.no_grad():
            t = self.proposal_target_layer(batch_dict)
        rois, gt = t["rois"], t["gt_of_rois"]
        t["gt_of_rois_src"] = gt.clone().detach()
        roi_ry = limit_period(rois[:, :, 6], offset=0.5, period=np.pi * 2)
        gt[:, :, :6] = gt[:, :, :6] - rois[:, :, :6]
        gt[:, :, 6] = gt[:, :, 6] - roi_ry
        gt = rotate_points_along_z(gt.view(-1, 1, gt.shape[-1]), -roi_ry.view(-1)).view(bs, -1, gt.shape[-1])
        if rois.shape[-1] == 9:
            gt[:, :, 7:-1] = gt[:, :, 7:-1] - rois[:, :, 7:]
        heading = gt[:, :, 6] % (2 * np.pi)                                  # flip when the RoI points the other way
        opposite = (heading > np.pi * 0.5) & (heading < np.pi * 1.5)
        heading[opposite] = (heading[opposite] + np.pi) % (2 * np.pi)
        flag = heading > np.pi
        heading[flag] = heading[flag] - np.pi * 2
        gt[:, :, 6] = torch.clamp(heading, min=-np.pi / 2, max=np.pi / 2)
        t["gt_of_rois"] = gt
        return t

    def get_box_reg_layer_loss(self, ret):
        cfg = _cfg_get(self.model_cfg, "LOSS_CONFIG")
        if _cfg_get(cfg, "REG_LOSS") != "L1":
            raise NotImplementedError(_cfg_get(cfg, "REG_LOSS"))
        weights = _cfg_get(cfg, "LOSS_WEIGHTS")
        code = ret["rcnn_reg"].shape[-1]
        fg = ret["reg_valid_mask"].view(-1) > 0
        target = ret["gt_of_rois"][..., 0:code].reshape(-1, code)
        loss = F.l1_loss(ret["rcnn_reg"].view(target.shape[0], -1), target, reduction="none")
        loss = loss * loss.new_tensor(weights["code_weights"])
        loss = (loss * fg.unsqueeze(-1).float()).sum() / max(int(fg.long().sum().item()), 1)
        loss = loss * weights["rcnn_reg_weight"]
        return loss, {"rcnn_loss_reg": loss.detach()}

    def get_box_cls_layer_loss(self, ret):
        cfg = _cfg_get(self.model_cfg, "LOSS_CONFIG")
        labels = ret["rcnn_cls_labels"].view(-1)
        kind = _cfg_get(cfg, "CLS_LOSS")
        if kind == "BinaryCrossEntropy":
            per = F.binary_cross_entropy(torch.sigmoid(ret["rcnn_cls"].view(-1)), labels.float(), reduction="none")
        elif kind == "CrossEntropy":
            per = F.cross_entropy(ret["rcnn_cls"], labels, reduction="none", ignore_index=-1)
        else:
            raise NotImplementedError(kind)
        valid = (labels >= 0).float()
        loss = (per * valid).sum() / torch.clamp(valid.sum(), min=1.0)
        loss = loss * _cfg_get(cfg, "LOSS_WEIGHTS")["rcnn_cls_weight"]
        return loss, {"rcnn_loss_cls": loss.detach()}

    def get_loss(self, tb_dict=None):
        tb = {} if tb_dict is None else tb_dict
        cls, c = self.get_box_cls_layer_loss(self.forward_ret_dict)
        reg, r = self.get_box_reg_layer_loss(self.forward_ret_dict)
        tb.update(c); tb.update(r)
        total = cls + reg
        tb["rcnn_loss"] = total.item()
        return total, tb

    @staticmethod
    def make_fc_layers(input_channels, output_channels, fc_list, dp_ratio):
        layers, pre = [], input_channels
        for k, c in enumerate(fc_list):
            layers += [nn.Conv1d(pre, c, kernel_size=1, bias=False), nn.BatchNorm1d(c), nn.ReLU()]
            pre = c
            if dp_ratio >= 0 and k == 0:
                layers.append(nn.Dropout(dp_ratio))
        layers.append(nn.Conv1d(pre, output_channels, kernel_size=1, bias=True))
        return nn.Sequential(*layers)

    @staticmethod
    def generate_predicted_boxes(batch_size, rois, cls_preds, box_preds):
        """residuals are predicted in the RoI's frame: add the RoI size / heading, rotate by its yaw, translate to its centre"""
        code_size = box_preds.shape[-1]
        batch_cls = cls_preds.view(batch_size, -1, cls_preds.shape[-1])
        box = box_preds.view(batch_size, -1, code_size)
        ry = rois[:, :, 6].reshape(-1)
        xyz = rois[:, :, 0:3].reshape(-1, 3)
        local = rois.clone().detach()
        local[:, :, 0:3] = 0
        box = (box + local).view(-1, code_size)
        cosa, sina = torch.cos(ry), torch.sin(ry)
        # box_torch_ops.rotate_points_along_z multiplies ROW vectors by [[c, -s, 0], [s, c, 0], [0, 0, 1]]:
        # x' = x*c + y*s, y' = -x*s + y*c
        x = box[:, 0] * cosa + box[:, 1] * sina
        y = -box[:, 0] * sina + box[:, 1] * cosa
        box = torch.cat([x[:, None], y[:, None], box[:, 2:]], dim=1)
        box[:, 0:3] += xyz
        return batch_cls, box.view(batch_size, -1, code_size)

    def forward(self, batch_dict, training=True):
        batch_dict["batch_size"] = len(batch_dict["rois"])
        targets = None
        if training:   # roi_head.py:76-80
            targets = self.assign_targets(batch_dict)
            batch_dict["rois"], batch_dict["roi_labels"], batch_dict["roi_features"] = targets["rois"], targets["roi_labels"], targets["roi_features"]
        pooled = batch_dict["roi_features"].reshape(-1, 1, batch_dict["roi_features"].shape[-1]).permute(0, 2, 1).contiguous()
        shared = self.shared_fc_layer(pooled)
        rcnn_cls = self.cls_layers(shared).transpose(1, 2).contiguous().squeeze(dim=1)
        rcnn_reg = self.reg_layers(shared).transpose(1, 2).contiguous().squeeze(dim=1)
        if training:
            targets["rcnn_cls"], targets["rcnn_reg"] = rcnn_cls, rcnn_reg
            self.forward_ret_dict = targets
            return batch_dict
        cls, box = self.generate_predicted_boxes(batch_dict["batch_size"], batch_dict["rois"], rcnn_cls, rcnn_reg)
        batch_dict["batch_cls_preds"], batch_dict["batch_box_preds"], batch_dict["cls_preds_normalized"] = cls, box, False
        return batch_dict


def box_side_centers(box3d):
    """centre + the four face-centre points of each BEV box (two_stage.py:50-74, box_torch_ops.center_to_corner_box2d):
    [5*n, 3], the five sets stacked along dim 0"""
    center2d, height, dim2d, yaw = box3d[:, :2], box3d[:, 2:3], box3d[:, 3:5], box3d[:, -1]
    # unit corners clockwise from the minimum point, origin 0.5: (-.5,-.5), (-.5,.5), (.5,.5), (.5,-.5)
    unit = box3d.new_tensor([[-0.5, -0.5], [-0.5, 0.5], [0.5, 0.5], [0.5, -0.5]])
    corners = dim2d[:, None, :] * unit[None]                                   # [n, 4, 2]
    s, c = torch.sin(yaw), torch.cos(yaw)
    # rotation_2d: corners @ [[c, -s], [s, c]] per box (einsum "aij,jka->aik" with rot_mat_T = [[c, -s], [s, c]])
    rx = corners[..., 0] * c[:, None] + corners[..., 1] * s[:, None]
    ry = -corners[..., 0] * s[:, None] + corners[..., 1] * c[:, None]
    corners = torch.stack([rx, ry], dim=-1) + center2d[:, None, :]
    mid = lambda a, b: torch.cat([(corners[:, a] + corners[:, b]) / 2, height], dim=-1)
    return torch.cat([box3d[:, :3], mid(0, 1), mid(2, 3), mid(0, 3), mid(1, 2)], dim=0)


@DETECTORS.register_module
class TwoStageDetector(nn.Module):
    def __init__(self, first_stage_cfg, second_stage_modules, roi_head, NMS_POST_MAXSIZE, num_point=1, freeze=False,
                 train_cfg=None, test_cfg=None, pretrained=None, **kwargs):
        super().__init__()
        self.single_det = registry.build_detector(first_stage_cfg, train_cfg=train_cfg, test_cfg=test_cfg)
        self.NMS_POST_MAXSIZE = NMS_POST_MAXSIZE
        if freeze:   # the reference trains in two steps: the first stage is frozen (two_stage.py:24-27)
            for p in self.single_det.parameters():
                p.requires_grad = False
            self.single_det.eval()
        self.freeze = freeze
        self.bbox_head = self.single_det.bbox_head
        self.second_stage = nn.ModuleList([registry.build(m, SECOND_STAGE) for m in second_stage_modules])
        self.roi_head = registry.build(roi_head, ROI_HEAD)
        self.num_point = num_point

    def train(self, mode=True):
        super().train(mode)
        if self.freeze:
            self.single_det.eval()
        return self

    def get_box_center(self, boxes):
        out = []
        for box in boxes:
            b = box["box3d_lidar"]
            if self.num_point == 1 or len(b) == 0:
                out.append(b[:, :3])
            elif self.num_point == 5:
                out.append(box_side_centers(b))
            else:
                raise NotImplementedError()
        return out

    def reorder_first_stage_pred_and_feature(self, first_pred, example, features):
        n, cap = len(first_pred), self.NMS_POST_MAXSIZE
        box_len = first_pred[0]["box3d_lidar"].shape[1]
        flen = sum(f[0].shape[-1] for f in features)
        rois = first_pred[0]["box3d_lidar"].new_zeros((n, cap, box_len))
        scores = first_pred[0]["scores"].new_zeros((n, cap))
        labels = first_pred[0]["label_preds"].new_zeros((n, cap), dtype=torch.long)
        feats = features[0][0].new_zeros((n, cap, flen))
        for i in range(n):
            k = features[0][i].shape[0]
            bp = first_pred[i]["box3d_lidar"]
            if self.roi_head.code_size == 9:   # (x, y, z, w, l, h, yaw, vx, vy)
                bp = bp[:, [0, 1, 2, 3, 4, 5, 8, 6, 7]]
            rois[i, :k], labels[i, :k], scores[i, :k] = bp, first_pred[i]["label_preds"] + 1, first_pred[i]["scores"]
            feats[i, :k] = torch.cat([f[i] for f in features], dim=-1)
        example.update(rois=rois, roi_labels=labels, roi_scores=scores, roi_features=feats, has_class_labels=True)
        return example

    def post_process(self, batch_dict):
        out = []
        meta = batch_dict.get("metadata")
        for i in range(batch_dict["batch_size"]):
            box, cls, lab = batch_dict["batch_box_preds"][i], batch_dict["batch_cls_preds"][i], batch_dict["roi_labels"][i]
            if box.shape[-1] == 9:
                box = box[:, [0, 1, 2, 3, 4, 5, 7, 8, 6]]
            scores = torch.sqrt(torch.sigmoid(cls).reshape(-1) * batch_dict["roi_scores"][i].reshape(-1))
            m = (lab != 0).reshape(-1)
            out.append(dict(box3d_lidar=box[m, :], scores=scores[m], label_preds=lab[m] - 1, metadata=meta[i] if meta else None))
        return out

    @staticmethod
    def combine_loss(one_stage_loss, roi_loss, tb_dict):
        """two_stage.py:40-47: the RoI loss is added to the first task's loss; its two terms are logged per task"""
        one_stage_loss["loss"][0] = one_stage_loss["loss"][0] + roi_loss
        one_stage_loss.setdefault("roi_reg_loss", [])
        one_stage_loss.setdefault("roi_cls_loss", [])
        for _ in range(len(one_stage_loss["loss"])):
            one_stage_loss["roi_reg_loss"].append(tb_dict["rcnn_loss_reg"])
            one_stage_loss["roi_cls_loss"].append(tb_dict["rcnn_loss_cls"])
        return one_stage_loss

    def forward(self, example, return_loss=True, return_feature=False, **kwargs):
        out = self.single_det.forward_two_stage(example, return_loss, **kwargs)
        f_a = f_b = None
        if len(out) == 6:
            one_stage_pred, bev_feature, voxel_feature, one_stage_loss, f_a, f_b = out
        else:
            one_stage_pred, bev_feature, voxel_feature, one_stage_loss = out
        example["voxel_feature"] = voxel_feature
        example["bev_feature"] = bev_feature.float().permute(0, 2, 3, 1).contiguous()   # N C H W -> N H W C
        centers = self.get_box_center(one_stage_pred)
        if self.roi_head.code_size == 7 and return_loss:   # drop the velocity columns (two_stage.py:173-175)
            example["gt_boxes_and_cls"] = example["gt_boxes_and_cls"][:, :, [0, 1, 2, 3, 4, 5, 6, -1]]
        features = [m(example, centers, self.num_point) for m in self.second_stage]
        example = self.reorder_first_stage_pred_and_feature(one_stage_pred, example, features)
        batch_dict = self.roi_head(example, training=return_loss)
        if return_loss:
            roi_loss, tb = self.roi_head.get_loss()
            return self.combine_loss(one_stage_loss, roi_loss, tb)
        res = self.post_process(batch_dict)
        return (res, f_a, f_b) if return_feature else res
