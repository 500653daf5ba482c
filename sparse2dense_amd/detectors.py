"""Detectors under the reference's registry keys.

  DETECTORS["VoxelNet"]      /root/reference/det3d/models/detectors/voxelnet.py:21-105   (teacher,
                             plain CenterPoint single stage, SECOND)
  DETECTORS["KD_VoxelNet"]   /root/reference/det3d/models/detectors/voxelnet.py:144-265 (S2D student)
  SingleStageDetector wiring /root/reference/det3d/models/detectors/single_stage.py:22-31

`forward(example, return_loss=..., return_feature=..., return_recon_feature=...)` keeps the
reference's argument meaning and return tuples.  Two deliberate device-side differences:
  * the reader output may be supplied by the fused voxelizer (`example["voxel_mean"]`), which is
    the same fp32 quantity (tests pin it to <= 2 ulp);
  * KD_VoxelNet builds the reconstruction targets with the densify kernel instead of spconv's
    scatter_nd (`SparseConvTensor(...).dense()`, voxelnet.py:203-215) — same tensor.
"""
import numpy as np
import torch
from torch import nn

from . import registry
from .heads import mask_offset_loss, mask_offset_loss_sparse, metric_grid
from .registry import DETECTORS
from .spconv import SparseConvTensor


class SingleStageDetector(nn.Module):
    def __init__(self, reader, backbone, neck=None, bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__()
        self.reader = registry.build_reader(reader)
        self.backbone = registry.build_backbone(backbone)
        self.neck = registry.build_neck(neck) if neck is not None else None
        self.bbox_head = registry.build_head(bbox_head)
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        self.dense_dtype = torch.float32   # set to torch.bfloat16 to run neck + head under autocast
        self.dense_channels_last = False    # set by use_channels_last()

    @property
    def with_neck(self):
        return self.neck is not None

    def use_channels_last(self):
        """NHWC activations/weights for the whole 2-D neck + head (avoids per-conv layout transposes under bf16).  Every
        BatchNorm2d on that path is a FastBatchNorm2d (own row-major kernels): MIOpen's NHWC bf16 batch norm, which
        segfaults on the 47x47 maps of the S2D module (ROCm 7.2), is never reached.  The PCR head (3-D) stays NCDHW fp32."""
        mods = list(self.bbox_head.modules())
        if self.neck is not None:
            mods += list(self.neck.modules())
            self.neck.trunk_channels_last = True
        if hasattr(self.backbone, "_module_2d"):   # pillar S2D backbone: its 2-D module is dense, same treatment as the neck
            mods += list(self.backbone.modules())
        from .dense2d import Conv1x1, Conv2x2S2, Conv3x3, ConvT2x2S2, ConvT4x4S2, DepthwiseConv7, SmallConv3x3
        from .dense3d import ConvTranspose3dK4S2, PointwiseConv3d
        own = (Conv1x1, Conv2x2S2, Conv3x3, ConvT2x2S2, ConvT4x4S2, DepthwiseConv7, SmallConv3x3)
        for mod in mods:
            # NHWC weights only where the library convs read them: our kernels pack either order, and a contiguous parameter gets its
            # (contiguous) gradient handed over as is - an NHWC one makes the autograd engine re-lay every weight gradient (a copy
            # kernel per parameter and step)
            if isinstance(mod, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)) and not isinstance(mod, own):
                mod.to(memory_format=torch.channels_last)
            if isinstance(mod, (ConvTranspose3dK4S2, PointwiseConv3d)):   # PCR head: bf16 MFMA inputs in the bf16 mode
                mod.bf16_compute = True
        self.dense_channels_last = True
        return self

    def _dense(self, module, x, keep_first=False, keep=()):
        """neck / head call; optionally under bf16 autocast (fp32 master weights, fp32 outputs).  keep_first: the first
        (or only) output stays in the compute dtype — the neck map that only feeds the head.  keep: further tuple positions left
        in the compute dtype (feature maps whose consumers upcast per element: a `.float()` of a [4,256,188,188] map is a 36 ->
        72 MB pass that the student-only step never uses)."""
        if self.dense_channels_last and x.is_cuda and x.dim() == 4 and module is self.bbox_head:
            x = x.contiguous(memory_format=torch.channels_last)   # NHWC: what the bf16 MFMA conv kernels consume
        if self.dense_dtype == torch.float32 or not x.is_cuda:
            return module(x)
        with torch.autocast("cuda", dtype=self.dense_dtype):
            out = module(x)
        f32 = lambda t: t.float() if torch.is_tensor(t) and t.is_floating_point() else t
        if isinstance(out, (tuple, list)):
            kept = set(keep) | ({0} if keep_first and torch.is_tensor(out[0]) else set())   # never upcast (and drop) a kept map
            conv = [o if i in kept else ({k: f32(v) for k, v in o.items()} if isinstance(o, dict) else f32(o)) for i, o in enumerate(out)]
            return type(out)(conv)
        return out if keep_first and torch.is_tensor(out) else f32(out)

    # ---- the shape-static dense segment as HIP graphs (graphed.GraphedSegment) ----------------------------------------------------------
    graph_dense = False   # set by use_hip_graphs()

    def use_hip_graphs(self, on=True):
        """Replay neck + head (+ losses in training) as HIP graphs: forward and backward of everything behind the BEV map are two graph
        launches per step instead of ~600 Python-issued kernel launches (sparse2dense_amd/graphed.py).  Needs CUDA inputs; steps whose
        batch norms synchronise over ranks (SyncBN collectives inside the segment) keep the eager path unless S2D_DENSE_GRAPH_SYNCBN=1."""
        self.graph_dense = bool(on)
        self._segments = {}
        return self

    def _graph_ok(self, x):
        if not (self.graph_dense and torch.is_tensor(x) and x.is_cuda):
            return False
        from . import collective, graphed
        import os
        if collective.sync_on() and os.environ.get("S2D_DENSE_GRAPH_SYNCBN", "0") != "1":
            return False
        return graphed.enabled()

    def _segment(self, name, fn, modules=None):
        from .graphed import GraphedSegment
        segs = self.__dict__.setdefault("_segments", {})
        seg = segs.get(name)
        if seg is None:
            mods = modules if modules is not None else [self.bbox_head] + ([self.neck] if self.neck is not None else [])
            seg = segs[name] = GraphedSegment(fn, mods, name=f"{type(self).__name__}.{name}")
        return seg

    @staticmethod
    def _flat_targets(example, tasks):
        """the CenterHead targets of an example as a flat tensor list (task-major)"""
        return [example[k][t] for t in range(tasks) for k in ("hm", "ind", "mask", "cat", "anno_box")]

    @staticmethod
    def _unflat_targets(flat, tasks):
        keys = ("hm", "ind", "mask", "cat", "anno_box")
        return {k: [flat[t * len(keys) + i] for t in range(tasks)] for i, k in enumerate(keys)}

    @staticmethod
    def _flat_struct(obj, out, spec_path=()):
        """flatten nested dict / list / tuple of tensors (and passthrough constants) into `out`; returns a spec to rebuild it"""
        if torch.is_tensor(obj):
            out.append(obj)
            return ("t", len(out) - 1)
        if isinstance(obj, dict):
            return ("d", type(obj) if type(obj) is dict else dict, [(k, SingleStageDetector._flat_struct(v, out)) for k, v in obj.items()])
        if isinstance(obj, (list, tuple)):
            return ("l", type(obj), [SingleStageDetector._flat_struct(v, out) for v in obj])
        return ("c", obj)

    @staticmethod
    def _unflat_struct(spec, flat):
        kind = spec[0]
        if kind == "t":
            return flat[spec[1]]
        if kind == "d":
            return spec[1]((k, SingleStageDetector._unflat_struct(v, flat)) for k, v in spec[2])
        if kind == "l":
            return spec[1](SingleStageDetector._unflat_struct(v, flat) for v in spec[2])
        return spec[1]

    def _run_segment(self, name, part, x, side_inputs, modules=None):
        """part(x, *side_inputs) -> nested structure of tensors; through the segment's graphs.  The structure is rebuilt around the
        (static) output tensors of the replay."""
        holder = {}

        def fn(x_, *side):
            flat = []
            holder["spec"] = self._flat_struct(part(x_, *side), flat)
            return tuple(flat)
        seg = self._segment(name, fn, modules)
        seg.fn = fn     # (the closure of THIS call: eager warm-up calls and the capture record the output structure through it)
        outs = seg(x, *side_inputs)
        spec = holder.get("spec")
        if spec is None:
            spec = seg.__dict__["_spec"]
        else:
            seg.__dict__["_spec"] = spec
        return self._unflat_struct(spec, list(outs))

    def _padded_list(self, key, coors, feats):
        """a voxel list (coors [M,4] + feats [M,C]) in capacity-sized persistent buffers (rows past the list carry batch index -1, which every
        PCR kernel skips): a static-shaped input for the graph, whatever the cloud's voxel count.  The copies of all lists of a step go out in
        one multi-tensor launch (`_flush_recon`)."""
        coors = coors if coors.dtype == torch.int32 else coors.int()
        feats = feats.float()
        m = int(coors.shape[0])
        store = self.__dict__.setdefault("_recon_pad", {})
        ent = store.get(key)
        if ent is None or ent[0].shape[0] < m or ent[0].device != coors.device or ent[1].shape[1] != feats.shape[1]:
            cap = -(-int(m * 1.5 + 1) // 65536) * 65536
            cb = torch.full((cap, 4), -1, dtype=torch.int32, device=coors.device)
            fb = torch.zeros((cap, feats.shape[1]), dtype=torch.float32, device=coors.device)
            cb._s2d_static = fb._s2d_static = True
            ent = store[key] = [cb, fb, 0]
        cb, fb, prev = ent
        pend = self.__dict__.setdefault("_recon_pending", ([], []))
        pend[0].extend([cb[:m], fb[:m]])
        pend[1].extend([coors, feats])
        if prev > m:
            cb[m:prev].fill_(-1)
        ent[2] = m
        return cb, fb

    def _flush_recon(self):
        pend = self.__dict__.pop("_recon_pending", None)
        if pend and pend[0]:
            from .graphed import _copy_all
            with torch.no_grad():
                _copy_all(pend[0], pend[1])

    def _read(self, example, prefix=""):
        mean_key = prefix + "voxel_mean"
        if mean_key in example:
            return example[mean_key]
        return self.reader(example[prefix + "voxels"], example[prefix + "num_points"])


@DETECTORS.register_module
class VoxelNet(SingleStageDetector):
    graphed_segment = True   # forward() routes everything behind the BEV map through _dense_call (use_hip_graphs)

    def _bev(self, data):
        # the BEV map goes straight to NHWC bf16 in the bf16 mode: the neck reads exactly that, and a caller that asked for the
        # feature (the distillation teacher's F_D_a) gets it in the compute dtype / layout (sparse2dense_loss upcasts per element)
        bev = bool(self.dense_channels_last and self.dense_dtype == torch.bfloat16 and self.with_neck and data["features"].is_cuda)
        return self.backbone(data["features"], data["coors"], data["batch_size"], data["input_shape"], bev_nhwc_bf16=bev)

    def _dense_part(self, x, example, return_loss):
        """everything behind the BEV map: neck -> CenterHead (-> losses).  Shapes depend on the batch size only: this is the segment
        `use_hip_graphs()` replays as HIP graphs."""
        neck = self._dense(self.neck, x, keep_first=True) if self.with_neck else x
        preds = self._dense(self.bbox_head, neck)
        losses = self.bbox_head.loss(example, preds) if return_loss else None
        return neck, preds, losses

    def _dense_call(self, x, example, return_loss):
        if self._graph_ok(x) and self.with_neck:
            tasks = len(self.bbox_head.tasks)
            name = f"{'train' if self.training else 'eval'}:{'loss' if return_loss else 'fwd'}:{int(torch.is_grad_enabled())}"
            if return_loss:
                return self._run_segment(name, lambda x_, *flat: self._dense_part(x_, self._unflat_targets(flat, tasks), True), x,
                                         self._flat_targets(example, tasks))
            return self._run_segment(name, lambda x_: self._dense_part(x_, None, False), x, [])
        return self._dense_part(x, example, return_loss)

    def extract_feat(self, data):
        x, voxel_feature = self._bev(data)
        neck = self._dense(self.neck, x, keep_first=True) if self.with_neck else x
        return neck, voxel_feature, x

    def forward(self, example, return_loss=True, return_feature=False, return_recon_feature=False, **kwargs):
        prefix = "dense_" if "dense_voxels" in example else ""   # teacher sees the dense cloud (voxelnet.py:50-54)
        batch_size = len(example[prefix + "num_voxels"])
        data = dict(features=self._read(example, prefix), coors=example[prefix + "coordinates"], batch_size=batch_size,
                    input_shape=example["shape"][0], bev_private=not return_feature)
        F_D_a, _ = self._bev(data)
        F_D_b = None
        if return_recon_feature:  # second backbone pass on the object-only cloud (voxelnet.py:73-89)
            bev = bool(self.dense_channels_last and self.dense_dtype == torch.bfloat16 and data["features"].is_cuda)
            F_D_b, _ = self.backbone(self._read(example, "reconstruction_"), example["reconstruction_coordinates"],
                                     batch_size, example["shape"][0], bev_nhwc_bf16=bev)
        x, preds, losses = self._dense_call(F_D_a, example, return_loss)
        if return_loss:
            return losses if not return_feature else (losses, F_D_a, F_D_b)
        if return_feature and return_recon_feature:
            return preds, F_D_a, F_D_b
        if kwargs.get("raw_preds", False):   # forward-only use (SECOND config 1: anchor decode is out of scope)
            return preds
        boxes = self.bbox_head.predict(example, preds, self.test_cfg)
        return boxes if not return_feature else (boxes, F_D_a, F_D_b)


    def forward_two_stage(self, example, return_loss=True, **kwargs):
        """first-stage pass of TwoStageDetector (voxelnet.py:107-141): decoded + NMS'd boxes, the neck map, voxel features"""
        prefix = "dense_" if "dense_voxels" in example else ""
        data = dict(features=self._read(example, prefix), coors=example[prefix + "coordinates"], batch_size=len(example[prefix + "num_voxels"]),
                    input_shape=example["shape"][0], bev_private=False)
        x, voxel_feature, F_D_a = self.extract_feat(data)
        preds = self._dense(self.bbox_head, x)
        boxes = self.bbox_head.predict(example, [{k: v.detach() for k, v in p.items()} for p in preds], self.test_cfg)
        if return_loss:
            return boxes, x, voxel_feature, self.bbox_head.loss(example, preds)
        return boxes, x, voxel_feature, None, F_D_a, F_D_a


@DETECTORS.register_module
class KD_VoxelNet(VoxelNet):
    def _bev(self, data):
        # the BEV map goes straight to NHWC bf16 when only the bf16 neck reads it (as in VoxelNet._bev)
        bev = bool(self.dense_channels_last and self.dense_dtype == torch.bfloat16 and data["features"].is_cuda)
        return self.backbone(data["features"], data["coors"], data["batch_size"], data["input_shape"], bev_nhwc_bf16=bev)

    def extract_feat(self, data, train_pcm=True):
        x, voxel_feature = self._bev(data)
        # F_S_a / F_S_b stay in the neck's compute dtype (sparse2dense_loss upcasts per element)
        x, gen_offset_2, gen_mask_2, gen_offset_4, gen_mask_4, F_S_a, F_S_b = self._dense(self.neck, x, keep_first=True, keep=(5, 6))
        return x, gen_offset_2, gen_mask_2, gen_offset_4, gen_mask_4, F_S_a, F_S_b, voxel_feature

    mask_offset_loss = staticmethod(mask_offset_loss)

    def _recon_gt(self, example, scale, batch_size):
        pre = f"reconstruction_"
        feats = self._read_scaled(example, scale)
        shape = np.array(example["shape"][0][::-1] / scale).astype("int64")   # voxelnet.py:199,210
        coors = example[f"reconstruction_coordinates_{scale}"].int()
        return SparseConvTensor(feats, coors, shape, batch_size).dense()

    def _read_scaled(self, example, scale):
        key = f"reconstruction_voxel_mean_{scale}"
        if key in example:
            return example[key]
        return self.reader(example[f"reconstruction_voxels_{scale}"], example[f"reconstruction_num_points_{scale}"])

    def _dense_part(self, x, example, return_loss, want_pcr=False):
        """everything behind the BEV map (S2D module + PCR head + RPN trunk + CenterHead + losses, voxelnet.py:216-265): shapes depend on
        the batch size only - the segment `use_hip_graphs()` replays as HIP graphs"""
        batch_size = x.shape[0]
        if want_pcr and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and hasattr(self.neck, "pcr_targets"):   # (the kernels' dtypes)
            # hand the recon voxels to the neck: its PCR levels return their losses directly (heads.pcr_level)
            self.neck.pcr_targets = {s: (example[f"reconstruction_coordinates_{s}"], self._read_scaled(example, s)) for s in (4, 2)}
        x, gen_offset_2, gen_mask_2, gen_offset_4, gen_mask_4, F_S_a, F_S_b = self._dense(self.neck, x, keep_first=True, keep=(5, 6))
        mask_loss = comp_loss = 0
        if want_pcr:
            if gen_offset_2.dim() == 0:   # fused levels: the slots already hold the losses
                m4, o4, m2, o2 = gen_mask_4, gen_offset_4, gen_mask_2, gen_offset_2
            elif gen_offset_2.is_cuda and gen_offset_2.dtype == torch.float32:
                # the reconstruction targets stay sparse: both PCR losses are evaluated at the recon voxels plus one dense
                # reduction over the occupancy logits (csrc/losses.hip) - no [B,5,20,752,752] / [B,5,10,376,376] volumes
                m4, o4 = mask_offset_loss_sparse(gen_offset_4, gen_mask_4, example["reconstruction_coordinates_4"],
                                                 self._read_scaled(example, 4))
                m2, o2 = mask_offset_loss_sparse(gen_offset_2, gen_mask_2, example["reconstruction_coordinates_2"],
                                                 self._read_scaled(example, 2))
            else:   # the reference's formulation (voxelnet.py:194-215,230-249), used by the CPU oracle stack
                recon_gt_2 = self._recon_gt(example, 2, batch_size)
                recon_gt_4 = self._recon_gt(example, 4, batch_size)
                n, _, d, h, w = recon_gt_4.shape
                grid_4 = metric_grid(n, d, h, w, recon_gt_4)
                m4, o4 = mask_offset_loss(gen_offset_4, gen_mask_4, recon_gt_4, grid_4)
                n, _, d, h, w = gen_offset_2.shape
                grid_2 = metric_grid(n, d, h, w, gen_offset_2)
                m2, o2 = mask_offset_loss(gen_offset_2, gen_mask_2, recon_gt_2, grid_2)
            mask_loss, comp_loss = m2 + m4, o2 + o4
        preds = self._dense(self.bbox_head, x)
        losses = self.bbox_head.loss(example, preds) if return_loss else None
        return losses, F_S_a, F_S_b, preds, mask_loss, comp_loss

    def _padded_recon(self, example, scale):
        return self._padded_list(scale, example[f"reconstruction_coordinates_{scale}"], self._read_scaled(example, scale))

    def _dense_part_s2d(self, x, example, want_pcr):
        """first half of `_dense_part` as a segment of its own: S2D module + PCR head + PCR losses -> (F_S_a, F_S_b, mask_loss, comp_loss)"""
        if want_pcr and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and hasattr(self.neck, "pcr_targets"):   # (the kernels' dtypes)
            self.neck.pcr_targets = {s: (example[f"reconstruction_coordinates_{s}"], self._read_scaled(example, s)) for s in (4, 2)}
        go2, gm2, go4, gm4, F_S_a, F_S_b = self._dense(self.neck.forward_s2d, x, keep=(4, 5))
        mask_loss = comp_loss = 0
        if want_pcr:
            assert go2.dim() == 0, "the split segments need the fused PCR levels"
            mask_loss, comp_loss = gm2 + gm4, go2 + go4
        return F_S_a, F_S_b, mask_loss, comp_loss

    def _dense_part_head(self, F_S_a, example, return_loss):
        """second half: RPN trunk + CenterHead (+ losses) -> (preds, losses)"""
        x = self._dense(self.neck.forward_trunk, F_S_a, keep_first=True)
        preds = self._dense(self.bbox_head, x)
        return preds, (self.bbox_head.loss(example, preds) if return_loss else None)

    def _dense_call(self, x, example, return_loss, want_pcr=False):
        fused = want_pcr and return_loss and hasattr(self.neck, "pcr_targets")
        import os
        if (self._graph_ok(x) and fused and hasattr(self.neck, "forward_s2d") and self.training and os.environ.get("S2D_GRAPH_SPLIT", "0") == "1"):
            # opt-in (S2D_GRAPH_SPLIT=1): TWO segments in sequence - [S2D module + PCR head + PCR losses] and [RPN trunk + CenterHead + losses]: in
            # the backward pass the second segment's weight-gradient graph (side.GRAPH_DEFER) runs beside the first segment's chain graph, the
            # first segment's beside the eager sparse backward.  Measured r05 (B = 4, 150 k points): 19.89 ms against 19.72 ms with one segment -
            # the PCR head's chip-filling streams leave the second graph nothing to fill, and the 72 MB hand-over copies cost what is gained
            tasks = len(self.bbox_head.tasks)
            tag = f"{int(torch.is_grad_enabled())}"
            recon = []
            for s_ in (4, 2):
                recon += list(self._padded_recon(example, s_))
            self._flush_recon()

            def part_a(x_, c4, f4, c2, f2):
                ex = {"reconstruction_coordinates_4": c4, "reconstruction_voxel_mean_4": f4, "reconstruction_coordinates_2": c2,
                      "reconstruction_voxel_mean_2": f2}
                return self._dense_part_s2d(x_, ex, True)
            F_S_a, F_S_b, mask_loss, comp_loss = self._run_segment("train:s2d+pcr:" + tag, part_a, x, recon, modules=[self.neck])
            preds, losses = self._run_segment("train:trunk+head:" + tag,
                                              lambda fa, *flat: self._dense_part_head(fa, self._unflat_targets(flat, tasks), True), F_S_a,
                                              self._flat_targets(example, tasks), modules=[self.neck, self.bbox_head])
            return losses, F_S_a, F_S_b, preds, mask_loss, comp_loss
        if self._graph_ok(x) and (fused or not want_pcr):
            tasks = len(self.bbox_head.tasks)
            name = f"{'train' if self.training else 'eval'}:{'loss' if return_loss else 'fwd'}:{int(want_pcr)}:{int(torch.is_grad_enabled())}"
            side = self._flat_targets(example, tasks) if return_loss else []
            nt = len(side)
            if want_pcr:
                for s_ in (4, 2):
                    side += list(self._padded_recon(example, s_))
                self._flush_recon()

            def part(x_, *flat):
                ex = self._unflat_targets(flat[:nt], tasks) if return_loss else {}
                if want_pcr:
                    for k, s_ in enumerate((4, 2)):
                        ex[f"reconstruction_coordinates_{s_}"] = flat[nt + 2 * k]
                        ex[f"reconstruction_voxel_mean_{s_}"] = flat[nt + 2 * k + 1]
                return self._dense_part(x_, ex, return_loss, want_pcr)
            return self._run_segment(name, part, x, side)
        return self._dense_part(x, example, return_loss, want_pcr)

    def forward(self, example, return_loss=True, return_feature=False, **kwargs):
        batch_size = len(example["num_voxels"])
        data = dict(features=self._read(example), coors=example["coordinates"], batch_size=batch_size,
                    input_shape=example["shape"][0])
        want_pcr = self.training and return_loss
        bev, _ = self._bev(data)
        losses, F_S_a, F_S_b, preds, mask_loss, comp_loss = self._dense_call(bev, example, return_loss, want_pcr)
        if return_loss:
            if not return_feature:
                return losses, preds
            return losses, F_S_a, F_S_b, preds, mask_loss, comp_loss
        boxes = self.bbox_head.predict(example, preds, self.test_cfg)
        return boxes if not return_feature else (boxes, F_S_a, F_S_b)

    def forward_two_stage(self, example, return_loss=True, **kwargs):
        """voxelnet.py:266-301 (the PCR head is not evaluated: train_pcm=False)"""
        data = dict(features=self._read(example), coors=example["coordinates"], batch_size=len(example["num_voxels"]),
                    input_shape=example["shape"][0])
        was_training = self.neck.training
        self.neck.eval() if not return_loss else None   # S2D_RPN builds its PCR outputs only in training mode
        try:
            x, _, _, _, _, F_S_a, F_S_b, voxel_feature = self.extract_feat(data, train_pcm=False)
        finally:
            self.neck.train(was_training)
        preds = self._dense(self.bbox_head, x)
        boxes = self.bbox_head.predict(example, [{k: v.detach() for k, v in p.items()} for p in preds], self.test_cfg)
        if return_loss:
            return boxes, x, voxel_feature, self.bbox_head.loss(example, preds)
        return boxes, x, voxel_feature, None, F_S_a, F_S_b
