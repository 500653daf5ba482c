"""Drop-in for `det3d.ops.point_cloud.point_cloud_ops.points_to_voxel` and
`det3d.core.input.voxel_generator.VoxelGenerator`
(/root/reference/det3d/ops/point_cloud/point_cloud_ops.py:112-184,
 /root/reference/det3d/core/input/voxel_generator.py:5-46), backed by the HIP voxelizer.

numpy in -> numpy out keeps the reference contract; torch cuda in -> torch cuda out is the
device overload the training step uses (no PCIe round trip).  `voxelize_batch` is the device-side
counterpart of `Voxelization.__call__` + `collate_kitti` (preprocess.py:316-345,
torchie/parallel/collate.py:105-144): per-sample voxelization, concatenation, batch index
prepended to the (z,y,x) coordinates.
"""
import numpy as np
import torch

from . import hip_ops as H


def points_to_voxel(points, voxel_size, coors_range, max_points=35, reverse_index=True, max_voxels=20000,
                    return_mean=False):
    """Same signature and return layout as the reference function.  `reverse_index=False`
    (xyz coordinate order) is served by flipping the columns."""
    is_np = isinstance(points, np.ndarray)
    if is_np:
        if not torch.cuda.is_available():
            raise RuntimeError("points_to_voxel: no ROCm device visible and there is no CPU fallback")
        pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).cuda()
    else:
        pts = points
    vs = np.asarray(voxel_size, dtype=np.float32)
    rng = np.asarray(coors_range, dtype=np.float32)
    voxels, coors, num, mean = H.voxelize(pts, vs, rng, int(max_points), int(max_voxels), with_mean=return_mean)
    if not reverse_index:
        coors = coors.flip(1)
    if is_np:
        out = (voxels.cpu().numpy(), coors.cpu().numpy(), num.cpu().numpy())
        return out + (mean.cpu().numpy(),) if return_mean else out
    return (voxels, coors, num, mean) if return_mean else (voxels, coors, num)


class VoxelGenerator:
    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000):
        point_cloud_range = np.array(point_cloud_range, dtype=np.float32)
        voxel_size = np.array(voxel_size, dtype=np.float32)
        grid_size = (point_cloud_range[3:] - point_cloud_range[:3]) / voxel_size
        self._grid_size = np.round(grid_size).astype(np.int64)  # voxel_generator.py:10-11
        self._voxel_size = voxel_size
        self._point_cloud_range = point_cloud_range
        self._max_num_points = max_num_points
        self._max_voxels = max_voxels

    def generate(self, points, max_voxels=-1, return_mean=False):
        if max_voxels == -1:
            max_voxels = self._max_voxels
        return points_to_voxel(points, self._voxel_size, self._point_cloud_range, self._max_num_points, True,
                               max_voxels, return_mean=return_mean)

    @property
    def voxel_size(self):
        return self._voxel_size

    @property
    def max_num_points_per_voxel(self):
        return self._max_num_points

    @property
    def point_cloud_range(self):
        return self._point_cloud_range

    @property
    def grid_size(self):
        return self._grid_size


def voxelize_batches(requests):
    """Several collated voxelizations (a training example voxelizes the same frames as input / dense / reconstruction clouds at up to
    three scales) with ONE host read between them: requests = [(generator, point_clouds, prefix), ...] -> merged dict of the
    `voxelize_batch` fields.  CUDA point clouds, <= 64 frames each."""
    pend = []
    for generator, point_clouds, prefix in requests:
        offs = [0]
        for pts in point_clouds:
            offs.append(offs[-1] + int(pts.shape[0]))
        cat = point_clouds[0] if len(point_clouds) == 1 else torch.cat([p.float() for p in point_clouds], 0)
        pend.append((prefix, H.voxelize_batch_launch(cat, offs, generator.voxel_size, generator.point_cloud_range,
                                                     generator.max_num_points_per_voxel, generator._max_voxels)))
    bases = torch.cat([p[1][4] for p in pend]).cpu().tolist()   # the one host read
    out, at = {}, 0
    for prefix, p in pend:
        k = p[4].numel()
        v, c, n, m, counts = H.voxelize_batch_collect(p, bases[at:at + k])
        at += k
        out.update({prefix + "voxels": v, prefix + "coordinates": c, prefix + "num_points": n, prefix + "num_voxels": counts,
                    prefix + "voxel_mean": m})
    return out


def voxelize_batch(generator: VoxelGenerator, point_clouds, max_voxels=-1, prefix="", batched=True):
    """List of per-sample cuda point tensors -> the collated example fields
    `{prefix}voxels f32[sum M,P,C]`, `{prefix}coordinates i32[sum M,4] (b,z,y,x)`,
    `{prefix}num_points i32[sum M]`, `{prefix}num_voxels i64[B]`, plus `{prefix}voxel_mean`
    (the fused reader output).  Key names: collate.py:105-144 / trainer.py:78-124."""
    mv = generator._max_voxels if max_voxels == -1 else max_voxels
    if batched and point_clouds[0].is_cuda and len(point_clouds) <= 64:
        # device-side batched voxelizer: one launch chain and one host read for the whole batch (csrc/voxelize.hip)
        offs = [0]
        for pts in point_clouds:
            offs.append(offs[-1] + int(pts.shape[0]))
        cat = point_clouds[0] if len(point_clouds) == 1 else torch.cat([p.float() for p in point_clouds], 0)
        v, c, n, m, counts = H.voxelize_batch(cat, offs, generator.voxel_size, generator.point_cloud_range,
                                              generator.max_num_points_per_voxel, mv)
        return {prefix + "voxels": v, prefix + "coordinates": c, prefix + "num_points": n, prefix + "num_voxels": counts,
                prefix + "voxel_mean": m}
    # launch every frame's voxelizer first, then read the B voxel counts with ONE host sync
    pend = [H.voxelize_async(pts, generator.voxel_size, generator.point_cloud_range, generator.max_num_points_per_voxel,
                             generator._max_voxels if max_voxels == -1 else max_voxels, with_mean=True)
            for pts in point_clouds]
    counts = torch.cat([p[4] for p in pend]).cpu().tolist()
    if counts and min(counts) < 0:
        raise H._lib.S2DError("s2d_voxelize_run: the look-back scan timed out on the device (csrc/scan.h)")
    vs, cs, ns, ms = [], [], [], []
    for b, ((v, c, n, m, _), k) in enumerate(zip(pend, counts)):
        cb = torch.empty((k, 4), dtype=torch.int32, device=c.device)
        cb[:, 0] = b
        cb[:, 1:] = c[:k]
        vs.append(v[:k]); cs.append(cb); ns.append(n[:k]); ms.append(m[:k])
    dev = vs[0].device
    return {
        prefix + "voxels": torch.cat(vs, 0),
        prefix + "coordinates": torch.cat(cs, 0),
        prefix + "num_points": torch.cat(ns, 0),
        prefix + "num_voxels": torch.tensor(counts, dtype=torch.int64, device=dev),
        prefix + "voxel_mean": torch.cat(ms, 0),
    }
