"""Two-stage CenterPoint head: what sits on top of the hot path's first stage in the reference's
`two_stage` configs (first stage frozen, configs/waymo/voxelnet/two_stage/*: num_point=5, NMS_POST_MAXSIZE=500).

  DETECTORS["TwoStageDetector"]        /root/reference/det3d/models/detectors/two_stage.py:8-199
  SECOND_STAGE["BEVFeatureExtractor"]  /root/reference/det3d/models/second_stage/bird_eye_view.py:9-41
                                       (bilinear_interpolate_torch, det3d/core/utils/center_utils.py:93-122)
  ROI_HEAD["RoIHead"]                  /root/reference/det3d/models/roi_heads/roi_head.py:16-106,
                                       roi_head_template.py:27-41,153-183 (make_fc_layers, generate_predicted_boxes)
  VoxelNet/KD_VoxelNet.forward_two_stage   /root/reference/det3d/models/detectors/voxelnet.py:107-141,266-301

Module / parameter names follow the reference (`single_det.*`, `roi_head.shared_fc_layer.0.weight` ...), so two-stage
checkpoints load through checkpoint.load_state_dict.  Inference path (`return_loss=False`): first-stage decode + rotated NMS
(heads.CenterHead.predict on the HIP kernels) -> 5 BEV feature samples per box -> RoI MLP -> refined boxes and scores.
The RoI head's TRAINING targets (ProposalTargetLayer: IoU-sampled RoIs, roi_heads/target_assigner/) are not built here; its
forward raises for training=True."""
import torch
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


def _cfg_get(cfg, key):
    return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)


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
        if training:
            raise NotImplementedError("RoIHead training targets (ProposalTargetLayer) are outside the built path; inference only")
        batch_dict["batch_size"] = len(batch_dict["rois"])
        pooled = batch_dict["roi_features"].reshape(-1, 1, batch_dict["roi_features"].shape[-1]).permute(0, 2, 1).contiguous()
        shared = self.shared_fc_layer(pooled)
        rcnn_cls = self.cls_layers(shared).transpose(1, 2).contiguous().squeeze(dim=1)
        rcnn_reg = self.reg_layers(shared).transpose(1, 2).contiguous().squeeze(dim=1)
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

    def forward(self, example, return_loss=True, return_feature=False, **kwargs):
        if return_loss:
            raise NotImplementedError("TwoStageDetector: RoI-head training (ProposalTargetLayer) is outside the built path")
        out = self.single_det.forward_two_stage(example, return_loss, **kwargs)
        one_stage_pred, bev_feature, voxel_feature, _, f_a, f_b = out
        example["voxel_feature"] = voxel_feature
        example["bev_feature"] = bev_feature.float().permute(0, 2, 3, 1).contiguous()   # N C H W -> N H W C
        centers = self.get_box_center(one_stage_pred)
        features = [m(example, centers, self.num_point) for m in self.second_stage]
        example = self.reorder_first_stage_pred_and_feature(one_stage_pred, example, features)
        batch_dict = self.roi_head(example, training=False)
        res = self.post_process(batch_dict)
        return (res, f_a, f_b) if return_feature else res
