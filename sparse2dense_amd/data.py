"""Synthetic batches in the reference's `example` format, built on the device.

Field names and dtypes follow `Voxelization.__call__` + `AssignLabel` + `Reformat` + `collate_kitti`
(/root/reference/det3d/datasets/pipelines/preprocess.py:316-463,553-624,
 /root/reference/det3d/torchie/parallel/collate.py:91-161) and `example_to_device`
(/root/reference/det3d/torchie/trainer/trainer.py:78-124).  The voxel fields are produced by the
HIP voxelizer from device-resident points; targets are CPU numpy (AssignLabel is CPU in the
reference too) uploaded once.
"""
import numpy as np
import torch

from . import scene
from .voxel_ops import VoxelGenerator, voxelize_batch, voxelize_batches

WAYMO_TRAIN_MAX_VOXELS = 150000


def waymo_generators(distill=False):
    g = {"": VoxelGenerator(scene.WAYMO_VOXEL, scene.WAYMO_RANGE, 5, WAYMO_TRAIN_MAX_VOXELS)}
    if distill:  # preprocess.py:289-314: same grid for dense/reconstruction, 2x and 4x voxels for the PCR targets
        vs = np.asarray(scene.WAYMO_VOXEL, np.float32)
        g["_2"] = VoxelGenerator(vs * 2, scene.WAYMO_RANGE, 5, WAYMO_TRAIN_MAX_VOXELS)
        g["_4"] = VoxelGenerator(vs * 4, scene.WAYMO_RANGE, 5, WAYMO_TRAIN_MAX_VOXELS)
    return g


class SyntheticFrames:
    """Holds B seeded scenes on the device (points + targets); `example()` runs the device
    voxelizer and returns the collated dict — i.e. the per-iteration data work of the hot path."""

    def __init__(self, batch_size, n_points=150000, seed=20240928, distill=False, device="cuda", beam_jitter=2e-4):
        self.device = torch.device(device)
        self.distill = distill
        self.gens = waymo_generators(distill)
        self.points, self.dense_points, self.recon_points = [], [], []
        gt_b, gt_c = [], []
        tg = {k: [] for k in ["hm", "anno_box", "ind", "mask", "cat"]}
        for b in range(batch_size):
            s = scene.make_scene(n_points, seed=seed + b, beam_jitter=beam_jitter)
            self.points.append(torch.from_numpy(s["points"]).to(self.device))
            if distill:
                d, r = scene.make_distill_points(s, seed=seed + 100 + b)
                self.dense_points.append(torch.from_numpy(d).to(self.device))
                self.recon_points.append(torch.from_numpy(r).to(self.device))
            # AssignLabel regroups a task's objects class by class (preprocess.py:506-530)
            order = np.concatenate([np.where(s["gt_classes"] == c)[0] for c in (1, 2, 3)])
            gt_b.append(s["gt_boxes"][order]); gt_c.append(s["gt_classes"][order])
            if self.device.type != "cuda":   # host restatement (tests without a GPU)
                t = scene.assign_targets(gt_b[-1], gt_c[-1])
                for k in tg:
                    tg[k].append(torch.from_numpy(t[k]))
        if self.device.type == "cuda":   # ground-truth boxes stay on the device; targets are assigned there every iteration
            from . import targets as _targets
            self.gt_boxes, self.gt_classes = _targets.pad_boxes(gt_b, gt_c, self.device)
            self.targets = None
        else:
            self.targets = {k: [torch.stack(v).to(self.device)] for k, v in tg.items()}
        self.grid_size = self.gens[""].grid_size

    def _targets(self):
        if self.targets is not None:
            return self.targets
        from . import targets as _targets
        return _targets.assign_label(self.gt_boxes, self.gt_classes)

    def example(self):
        if self.device.type == "cuda" and len(self.points) <= 64:
            # every cloud of the example is voxelized before the host reads the (five) row-offset vectors in one go
            req = [(self.gens[""], self.points, "")]
            if self.distill:
                req += [(self.gens[""], self.dense_points, "dense_"), (self.gens[""], self.recon_points, "reconstruction_"),
                        (self.gens["_2"], self.recon_points, "reconstruction@_2"), (self.gens["_4"], self.recon_points, "reconstruction@_4")]
            ex = {}
            for k, v in voxelize_batches(req).items():
                if "@" in k:   # scaled reconstruction clouds: the scale suffix goes to the END of the key (trainer.py:100-124)
                    head, rest = k.split("@")
                    suf, field = rest[:2], rest[2:]
                    ex[head + "_" + field + suf] = v
                else:
                    ex[k] = v
        else:
            ex = voxelize_batch(self.gens[""], self.points)
            if self.distill:
                ex.update(voxelize_batch(self.gens[""], self.dense_points, prefix="dense_"))
                ex.update(voxelize_batch(self.gens[""], self.recon_points, prefix="reconstruction_"))
                for suf in ("_2", "_4"):
                    r = voxelize_batch(self.gens[suf], self.recon_points, prefix="reconstruction_")
                    for k, v in r.items():
                        ex[k + suf] = v
        ex["shape"] = np.stack([self.grid_size] * len(self.points))
        ex.update(self._targets())
        return ex


class SyntheticPillarFrames:
    """Scene C: the same sweeps voxelized as pillars (0.32 m, 20 points, 32 000 pillars,
    configs/waymo/pp/...:156-162) plus the object-only cloud for the PCR target; targets on the
    468 x 468 map (out_size_factor 1)."""

    def __init__(self, batch_size, n_points=150000, seed=20240928, device="cuda"):
        self.device = torch.device(device)
        self.gen = VoxelGenerator(scene.PILLAR_VOXEL, scene.PILLAR_RANGE, 20, 32000)
        self.points, self.recon_points = [], []
        tg = {k: [] for k in ["hm", "anno_box", "ind", "mask", "cat"]}
        for b in range(batch_size):
            s = scene.make_scene(n_points, seed=seed + b, pc_range=scene.PILLAR_RANGE)
            self.points.append(torch.from_numpy(s["points"]).to(self.device))
            self.recon_points.append(torch.from_numpy(s["object_points"]).to(self.device))
            t = scene.assign_targets(s["gt_boxes"], s["gt_classes"], pc_range=scene.PILLAR_RANGE,
                                     voxel_size=scene.PILLAR_VOXEL, out_size_factor=1, grid_xy=(468, 468))
            for k in tg:
                tg[k].append(torch.from_numpy(t[k]))
        self.targets = {k: [torch.stack(v).to(self.device)] for k, v in tg.items()}
        self.grid_size = self.gen.grid_size

    def example(self):
        ex = voxelize_batch(self.gen, self.points)
        ex.update(voxelize_batch(self.gen, self.recon_points, prefix="reconstruction_"))
        ex["shape"] = np.stack([self.grid_size] * len(self.points))
        ex.update(self.targets)
        return ex
