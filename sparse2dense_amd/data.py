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


def attach_geometry(example, backbone, keys=("coordinates", "dense_coordinates", "reconstruction_coordinates")):
    """Build every rulebook of a backbone pass from the coordinates alone (backbones.build_geometry) and hang the plan on the
    coordinate tensor: the backbone's forward then finds it instead of building it (and reading four row counts) itself."""
    from .backbones import build_geometry
    strided, subm = backbone._specs()
    shape = np.array(example["shape"][0][::-1]) + [1, 0, 0]   # (z + 1, y, x), scn.py:159
    for key in keys:
        coors = example.get(key)
        if coors is None or not coors.is_cuda or coors.dtype != torch.int32:
            continue
        batch = len(example[key.replace("coordinates", "num_voxels")])
        plan = build_geometry(coors, batch, shape, strided, subm)
        # The plan hangs on the coordinate tensor; a rulebook of the plan that holds that very tensor object (the SubM layers' out_coors)
        # would close a reference cycle - tensor -> plan -> rulebook -> tensor - and the example's ~0.75 GB of device memory (gather maps,
        # voxels) would then live until Python's CYCLIC collector happens to run instead of dying with the step (r05: 15 GB held after 20
        # steps with the collector off).  The plan gets an alias of the tensor instead (same storage, no attribute dictionary).
        for rb in plan.values():
            for name, val in list(vars(rb).items()):
                if val is coors:
                    setattr(rb, name, coors.detach())
        coors._s2d_geometry = (tuple(int(s) for s in shape), batch, plan)
    return example


def _tensors_of(obj, out=None, depth=0):
    """every tensor reachable from an example (dict / list / tuple values, and the geometry plan attach_geometry hangs on the
    coordinate tensors)"""
    out = [] if out is None else out
    if torch.is_tensor(obj):
        out.append(obj)
        geo = getattr(obj, "_s2d_geometry", None)
        if geo is not None and depth < 4:
            _tensors_of(geo, out, depth + 1)
    elif isinstance(obj, dict):
        for v in obj.values():
            _tensors_of(v, out, depth)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _tensors_of(v, out, depth)
    elif hasattr(obj, "__dict__") and depth < 4:      # Rulebook dataclasses of the plan
        _tensors_of(vars(obj), out, depth + 1)
    return out


class PrefetchLoader:
    """The reference overlaps its data pipeline (voxelization, AssignLabel: DataLoader workers) with the training step; this is the
    device-side equivalent.  A worker thread builds example k+1 on a second HIP stream - device voxelization, target assignment and
    (with `backbone`) the rulebooks, including the host reads of the row counts those need - while the main thread enqueues step k.
    The main stream then never waits on the host inside a step, so the launch queue stays ahead of the device.

    Memory hand-over without record_stream(): tensors of example k are allocated on the side stream's pool; a block freed by the
    main thread is only reused by side-stream work of a LATER prefetch, and every prefetch starts (side.wait_event) behind the
    main-stream position recorded when it was requested, i.e. behind all work that could still read such a block.  The one
    prefetch that is NOT behind step k is k+1 (requested before step k was enqueued, running beside it): the loader therefore keeps
    its own references to every tensor of example k (`_held`) until example k+1 has been handed out, so that a caller who drops
    a tensor mid-step (`step(loader.example())`, `del ex`, popped keys) cannot return its block to the side pool while
    prefetch k+1 still allocates."""

    def __init__(self, frames, backbone=None, geometry_keys=None):
        import queue
        import threading
        self.frames = frames
        self.backbone = backbone
        self.geometry_keys = geometry_keys
        self.device = frames.device
        self.side = torch.cuda.Stream(self.device)
        self._go, self._out = queue.Queue(), queue.Queue()
        self._closed = False
        self._held = None
        self._thread = threading.Thread(target=self._run, name="s2d-prefetch", daemon=True)
        self._thread.start()
        self._request()

    def __getattr__(self, name):   # grid_size, gens, points, ... of the wrapped frames
        return getattr(self.frames, name)

    def _build(self):
        ex = self.frames.example()
        if self.backbone is not None:
            kw = {} if self.geometry_keys is None else {"keys": self.geometry_keys}
            attach_geometry(ex, self.backbone, **kw)
        return ex

    def _request(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._go.put(ev)

    def _run(self):
        torch.cuda.set_device(self.device)
        while True:
            ev = self._go.get()
            if ev is None:
                return
            try:
                from .graphed import CAPTURE_LOCK
                with CAPTURE_LOCK, torch.cuda.stream(self.side):   # (never beside a HIP-graph capture: see graphed.CAPTURE_LOCK)
                    self.side.wait_event(ev)
                    ex = self._build()
                    done = torch.cuda.Event()
                    done.record(self.side)
                self._out.put((ex, done, None))
            except BaseException as err:   # surfaced by the next example()
                self._out.put((None, None, err))

    def example(self):
        if self._closed:
            raise RuntimeError("PrefetchLoader is closed")
        ex, done, err = self._out.get()
        if err is not None:
            self._closed = True
            raise err
        torch.cuda.current_stream(self.device).wait_event(done)
        # prefetch k+1 is fully enqueued (it is `ex`): example k's tensors may go; example k+1's are pinned until the next call
        self._held = _tensors_of(ex)
        self._request()
        return ex

    def close(self):
        if not self._closed:
            self._closed = True
            self._go.put(None)
            self._thread.join(timeout=30)
