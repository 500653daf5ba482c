"""ORACLE (test infrastructure only) — ctypes front-end of oracle/voxelize.c plus a pure-numpy
loop twin for tiny inputs.

Follows /root/reference/det3d/ops/point_cloud/point_cloud_ops.py:7-55,112-184 (hard voxelizer)
and /root/reference/det3d/models/readers/voxel_encoder.py:17-24 (mean reader).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile oracle/*.c into oracle/_build/liboracle.so (gcc, no fast-math)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return os.path.join(_HERE, "_build", "liboracle.so")


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        lib.s2d_oracle_points_to_voxel.restype = ctypes.c_int
        lib.s2d_oracle_points_to_voxel.argtypes = [
            ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.s2d_oracle_voxel_mean.restype = None
        lib.s2d_oracle_voxel_mean.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
            ctypes.c_int, ctypes.c_void_p]
        _LIB = lib
    return _LIB


def points_to_voxel(points, voxel_size, coors_range, max_points=35, max_voxels=20000):
    """C oracle of points_to_voxel(..., reverse_index=True) (point_cloud_ops.py:112-184).

    Returns (voxels f32[M,max_points,ndim], coors i32[M,3] (z,y,x), num_points i32[M])."""
    points = np.ascontiguousarray(points, dtype=np.float32)
    vs = np.ascontiguousarray(voxel_size, dtype=np.float32)
    rng = np.ascontiguousarray(coors_range, dtype=np.float32)
    n, ndim = points.shape
    voxels = np.zeros((max_voxels, max_points, ndim), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    num = np.zeros((max_voxels,), np.int32)
    m = _lib().s2d_oracle_points_to_voxel(
        points.ctypes.data, n, ndim, vs.ctypes.data, rng.ctypes.data, max_points, max_voxels,
        voxels.ctypes.data, coors.ctypes.data, num.ctypes.data)
    if m < 0:
        raise MemoryError("oracle voxelizer: lookup grid allocation failed")
    return voxels[:m].copy(), coors[:m].copy(), num[:m].copy()


def voxel_mean(voxels, num_points, n_feat=None):
    """C oracle of VoxelFeatureExtractorV3.forward (voxel_encoder.py:17-24)."""
    voxels = np.ascontiguousarray(voxels, dtype=np.float32)
    num_points = np.ascontiguousarray(num_points, dtype=np.int32)
    m, t, ndim = voxels.shape
    n_feat = ndim if n_feat is None else n_feat
    out = np.empty((m, n_feat), np.float32)
    _lib().s2d_oracle_voxel_mean(voxels.ctypes.data, num_points.ctypes.data, m, t, ndim, n_feat,
                                 out.ctypes.data)
    return out


def points_to_voxel_py(points, voxel_size, coors_range, max_points=35, max_voxels=20000):
    """Pure-Python loop twin (tiny inputs only) — a second, independent restatement of
    point_cloud_ops.py:33-54 used to cross-check the C file."""
    points = np.asarray(points, np.float32)
    vs = np.asarray(voxel_size, np.float32)
    rng = np.asarray(coors_range, np.float32)
    grid = np.round((rng[3:] - rng[:3]) / vs).astype(np.int32)
    lut = {}
    voxels, coors, num = [], [], []
    for i in range(points.shape[0]):
        c = np.floor((points[i, :3] - rng[:3]) / vs)
        if np.any(c < 0) or np.any(c >= grid):
            continue
        key = (int(c[2]), int(c[1]), int(c[0]))
        v = lut.get(key, -1)
        if v == -1:
            if len(coors) >= max_voxels:
                continue
            v = len(coors)
            lut[key] = v
            coors.append(key)
            voxels.append(np.zeros((max_points, points.shape[1]), np.float32))
            num.append(0)
        if num[v] < max_points:
            voxels[v][num[v]] = points[i]
            num[v] += 1
    if not coors:
        return (np.zeros((0, max_points, points.shape[1]), np.float32), np.zeros((0, 3), np.int32),
                np.zeros((0,), np.int32))
    return np.stack(voxels), np.asarray(coors, np.int32), np.asarray(num, np.int32)
