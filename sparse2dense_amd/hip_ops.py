"""Host-side launchers of the HIP kernels: torch tensors in, raw device pointers + the current
HIP stream into the C-ABI (include/s2d.h).  No compute happens in Python and nothing here
falls back to the CPU: tensors must live on a ROCm device.

Reference interfaces these stand in for (file:line under /root/reference):
  voxelize            det3d/ops/point_cloud/point_cloud_ops.py:112-184 + readers/voxel_encoder.py:17-24
  subm/conv rulebook  spconv.ops.get_indice_pairs          (call sites det3d/models/backbones/scn.py:104-152)
  spconv fwd/dgrad/wgrad  spconv.ops.indice_conv[_backward] (same call sites)
  bn1d_*              nn.BatchNorm1d on .features           (scn.py:73-83)
  densify             spconv.SparseConvTensor.dense()       (scn.py:173)
"""
import ctypes
import os
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check, f3, f6, i3


def _stream():
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())   # raw hipStream_t (cheap accessor, see dense2d._stream)


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    return t.data_ptr()   # a plain int: ctypes converts it for the c_void_p parameters


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.S2DError("sparse2dense_amd HIP op called with a CPU tensor: there is no CPU fallback")


def _ws(nbytes, device):
    """private workspace (rulebook / voxelizer jobs keep state in it between their two calls)"""
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _ws_shared(nbytes, device):
    from .dense2d import _ws as shared_ws   # one grow-only scratch buffer per (device, stream), see there
    return shared_ws(nbytes, device)


# ------------------------------------------------------------------------------------------------
# voxelization
# ------------------------------------------------------------------------------------------------
def voxelize_async(points: torch.Tensor, voxel_size, coors_range, max_points: int, max_voxels: int, with_mean=True):
    """Launches the device hard-voxelizer and returns UNTRIMMED buffers plus the device scalar M:
    (voxels f32[rows,max_points,ndim], coors i32[rows,3], num_points i32[rows], mean|None, out_m i32[1])."""
    lib = _lib.load()
    _need_gpu(points)
    points = points.contiguous().float()
    n, ndim = points.shape
    dev = points.device
    rows = max(min(n, max_voxels), 1)   # ids are ranks of first points: never more rows than this
    voxels = torch.empty((rows, max_points, ndim), dtype=torch.float32, device=dev)
    coors = torch.empty((rows, 3), dtype=torch.int32, device=dev)
    num = torch.empty((rows,), dtype=torch.int32, device=dev)
    mean = torch.empty((rows, ndim), dtype=torch.float32, device=dev) if with_mean else None
    out_m = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws = _ws(lib.s2d_voxelize_workspace_bytes(n, max_points, max_voxels), dev)
    check(lib.s2d_voxelize_run(_ptr(points), n, ndim, f6(coors_range), f3(voxel_size), max_points, max_voxels,
                               _ptr(voxels), _ptr(coors), _ptr(num), _ptr(mean), _ptr(out_m), _ptr(ws), ws.numel(),
                               _stream()), "s2d_voxelize_run")
    return voxels, coors, num, mean, out_m


def voxelize(points: torch.Tensor, voxel_size, coors_range, max_points: int, max_voxels: int, with_mean=True):
    """Device hard-voxelizer.  points f32[N,ndim] (cuda).  Returns
    (voxels f32[M,max_points,ndim], coors i32[M,3] (z,y,x), num_points i32[M], mean f32[M,ndim]|None).
    One host read (M) — the same size the reference API returns."""
    voxels, coors, num, mean, out_m = voxelize_async(points, voxel_size, coors_range, max_points, max_voxels, with_mean)
    m = int(out_m.item())
    if m < 0:
        raise _lib.S2DError("s2d_voxelize_run: the look-back scan timed out on the device (csrc/scan.h); no voxels were produced")
    return voxels[:m], coors[:m], num[:m], (mean[:m] if with_mean else None)


def voxelize_batch_launch(points_cat: torch.Tensor, offsets, voxel_size, coors_range, max_points: int, max_voxels: int):
    """Launch the batched voxelizer (B frames, points concatenated frame after frame, `offsets` = B+1 host ints) without reading
    anything back.  Returns the capacity-sized outputs + the device row offsets `out_base` i32[B+1]; voxelize_batch_collect trims
    them once the host knows out_base (several launches can share ONE host read)."""
    lib = _lib.load()
    _need_gpu(points_cat)
    points_cat = points_cat.contiguous().float()
    n, ndim = points_cat.shape
    frames = len(offsets) - 1
    assert offsets[0] == 0 and offsets[-1] == n
    offs = (ctypes.c_int64 * (frames + 1))(*[int(o) for o in offsets])
    dev = points_cat.device
    rows = max(sum(min(int(offsets[b + 1] - offsets[b]), max_voxels) for b in range(frames)), 1)
    voxels = torch.empty((rows, max_points, ndim), dtype=torch.float32, device=dev)
    coors = torch.empty((rows, 4), dtype=torch.int32, device=dev)
    num = torch.empty((rows,), dtype=torch.int32, device=dev)
    mean = torch.empty((rows, ndim), dtype=torch.float32, device=dev)
    out_m = torch.empty((frames,), dtype=torch.int32, device=dev)
    out_base = torch.empty((frames + 1,), dtype=torch.int32, device=dev)
    ws = _ws(lib.s2d_voxelize_batch_workspace_bytes(frames, offs, max_points, max_voxels), dev)
    rec = None
    if PROFILE is not None:
        rec = dict(kernel="voxelize", tag="voxelize", cin=0, cout=0, n_points=int(n), frames=frames, out_base=out_base, max_points=max_points,
                   ndim=int(ndim), start=torch.cuda.Event(enable_timing=True), end=torch.cuda.Event(enable_timing=True))
        rec["start"].record()
    check(lib.s2d_voxelize_batch_run(_ptr(points_cat), frames, offs, ndim, f6(coors_range), f3(voxel_size), max_points, max_voxels,
                                     _ptr(voxels), _ptr(coors), _ptr(num), _ptr(mean), _ptr(out_m), _ptr(out_base), _ptr(ws), ws.numel(),
                                     _stream()), "s2d_voxelize_batch_run")
    if rec is not None:
        rec["end"].record()
        PROFILE.append(rec)
    return (voxels, coors, num, mean, out_base)


def voxelize_batch_collect(pending, base):
    """trim a voxelize_batch_launch result with its row offsets `base` (B+1 host ints)"""
    voxels, coors, num, mean, out_base = pending
    m = int(base[-1])
    if m < 0:
        raise _lib.S2DError("s2d_voxelize_batch_run: the look-back scan timed out on the device (csrc/scan.h); no voxels were produced")
    counts = torch.tensor([int(base[b + 1]) - int(base[b]) for b in range(len(base) - 1)], dtype=torch.int64, device=voxels.device)
    return voxels[:m], coors[:m], num[:m], mean[:m], counts


def voxelize_batch(points_cat: torch.Tensor, offsets, voxel_size, coors_range, max_points: int, max_voxels: int):
    """B frames voxelized in ONE launch chain straight into the collated layout.  Returns (voxels f32[M,P,C], coors i32[M,4]
    (b,z,y,x), num_points i32[M], mean f32[M,C], num_voxels i64[B]) with ONE host read (the B+1 row offsets)."""
    pending = voxelize_batch_launch(points_cat, offsets, voxel_size, coors_range, max_points, max_voxels)
    return voxelize_batch_collect(pending, pending[4].cpu().tolist())   # the one host read of the batch


# ------------------------------------------------------------------------------------------------
# rulebooks
# ------------------------------------------------------------------------------------------------
@dataclass
class Rulebook:
    """Dense gather maps of one sparse conv layer (see include/s2d.h)."""
    subm: bool
    kvol: int
    n_in: int
    n_out: int
    nbr_out: torch.Tensor              # i32[K, n_out]: input row per (offset, output row) or -1
    nbr_in: Optional[torch.Tensor]     # i32[K, n_in]: output row per (offset, input row) or -1 (None for subm)
    pair_count: torch.Tensor           # i32[K]
    out_coors: Optional[torch.Tensor]  # i32[n_out,4] (None for subm: same as input)
    out_shape: Tuple[int, int, int]
    coors: Optional[torch.Tensor] = None   # subm: the i32[n,4] site coordinates

    def pairs(self):
        """spconv-style per-offset (in_idx, out_idx) lists, on the host (tests / export only)."""
        nb = self.nbr_out.cpu().numpy()
        res = []
        for k in range(self.kvol):
            o = np.nonzero(nb[k] >= 0)[0]
            res.append((nb[k][o].astype(np.int64), o.astype(np.int64)))
        return res


def conv_out_shape(shape, ksize, stride, padding, dilation=(1, 1, 1)):
    return tuple((int(shape[i]) + 2 * padding[i] - dilation[i] * (ksize[i] - 1) - 1) // stride[i] + 1 for i in range(3))


def build_subm_rulebook(coors: torch.Tensor, batch: int, shape, ksize, dilation=(1, 1, 1)) -> Rulebook:
    lib = _lib.load()
    _need_gpu(coors)
    coors = coors.contiguous()
    assert coors.dtype == torch.int32 and coors.dim() == 2 and coors.shape[1] == 4
    n = coors.shape[0]
    dev = coors.device
    kvol = int(ksize[0] * ksize[1] * ksize[2])
    nbr = torch.empty((kvol, n), dtype=torch.int32, device=dev)
    cnt = torch.empty((kvol,), dtype=torch.int32, device=dev)
    ws = _ws(lib.s2d_rulebook_workspace_bytes(batch, i3(shape), n), dev)
    rec = None
    if PROFILE is not None:
        rec = dict(kernel="rulebook_subm", tag="rulebook", cin=0, cout=0, n_out=int(n), kvol=kvol, pairs=cnt,
                   start=torch.cuda.Event(enable_timing=True), end=torch.cuda.Event(enable_timing=True))
        rec["start"].record()
    check(lib.s2d_rulebook_subm_build(_ptr(coors), n, batch, i3(shape), i3(ksize), i3(dilation), _ptr(nbr), _ptr(cnt),
                                      _ptr(ws), ws.numel(), _stream()), "s2d_rulebook_subm_build")
    if rec is not None:
        rec["end"].record()
        PROFILE.append(rec)
    return Rulebook(True, kvol, n, n, nbr, None, cnt, None, tuple(int(s) for s in shape), coors=coors)


class ConvRulebookJob:
    """Two-phase strided-conv rulebook: `count()` launches the marking/numbering kernels,
    `finish()` reads n_out back (one 4-byte D2H copy) and fills the maps."""

    def __init__(self, coors, batch, shape, ksize, stride, padding, dilation=(1, 1, 1)):
        self.lib = _lib.load()
        _need_gpu(coors)
        self.coors = coors.contiguous()
        assert self.coors.dtype == torch.int32 and self.coors.shape[1] == 4
        self.batch, self.shape = int(batch), tuple(int(s) for s in shape)
        self.ksize, self.stride = tuple(map(int, ksize)), tuple(map(int, stride))
        self.padding, self.dilation = tuple(map(int, padding)), tuple(map(int, dilation))
        self.out_shape = conv_out_shape(self.shape, self.ksize, self.stride, self.padding, self.dilation)
        self.kvol = self.ksize[0] * self.ksize[1] * self.ksize[2]
        dev = self.coors.device
        self.ws = _ws(self.lib.s2d_rulebook_workspace_bytes(self.batch, i3(self.out_shape), 0), dev)
        self.out_n = torch.zeros((1,), dtype=torch.int32, device=dev)

    def count(self):
        n = self.coors.shape[0]
        self._rec = None
        if PROFILE is not None:
            self._rec = dict(kernel="rulebook_conv", tag="rulebook", cin=0, cout=0, n_in=int(n), kvol=self.kvol,
                             start=torch.cuda.Event(enable_timing=True), end=torch.cuda.Event(enable_timing=True),
                             start2=torch.cuda.Event(enable_timing=True), end2=torch.cuda.Event(enable_timing=True))
            self._rec["start"].record()
        check(self.lib.s2d_rulebook_conv_count(_ptr(self.coors), n, self.batch, i3(self.shape), i3(self.ksize),
                                               i3(self.stride), i3(self.padding), i3(self.dilation), _ptr(self.out_n),
                                               _ptr(self.ws), self.ws.numel(), _stream()), "s2d_rulebook_conv_count")
        if self._rec is not None:
            self._rec["end"].record()
        return self

    def finish(self) -> Rulebook:
        n_out = int(self.out_n.item())   # the one host read of this layer (spconv reads indice_pair_num here)
        n = self.coors.shape[0]
        dev = self.coors.device
        out_coors = torch.empty((n_out, 4), dtype=torch.int32, device=dev)
        nbr_out = torch.empty((self.kvol, n_out), dtype=torch.int32, device=dev)
        nbr_in = torch.empty((self.kvol, n), dtype=torch.int32, device=dev)
        cnt = torch.empty((self.kvol,), dtype=torch.int32, device=dev)
        rec = getattr(self, "_rec", None)
        if rec is not None:
            rec["start2"].record()
        check(self.lib.s2d_rulebook_conv_fill(_ptr(self.coors), n, self.batch, i3(self.shape), i3(self.ksize),
                                              i3(self.stride), i3(self.padding), i3(self.dilation), n_out,
                                              _ptr(out_coors), _ptr(nbr_out), _ptr(nbr_in), _ptr(cnt), _ptr(self.ws),
                                              self.ws.numel(), _stream()), "s2d_rulebook_conv_fill")
        if rec is not None and PROFILE is not None:
            rec["end2"].record()
            rec.update(n_out=int(n_out), pairs=cnt)
            PROFILE.append(rec)
        return Rulebook(False, self.kvol, n, n_out, nbr_out, nbr_in, cnt, out_coors, self.out_shape)


def build_conv_rulebook(coors, batch, shape, ksize, stride, padding, dilation=(1, 1, 1)) -> Rulebook:
    return ConvRulebookJob(coors, batch, shape, ksize, stride, padding, dilation).count().finish()


def _i32_flat(rows):
    flat = [int(v) for r in rows for v in r]
    return (ctypes.c_int32 * len(flat))(*flat)


def rulebook_chain_supported(batch, shape, strided_specs):
    """`strided_specs[l]` = (ksize, stride, padding) triples of the strided convs in order (s2d_rulebook_chain_supported)"""
    if os.environ.get("S2D_RULEBOOK", "chain") != "chain" or not strided_specs:
        return False
    ks, st, pd = (_i32_flat([s[j] for s in strided_specs]) for j in range(3))
    return bool(_lib.load().s2d_rulebook_chain_supported(int(batch), i3(shape), len(strided_specs), ks, st, pd))


def build_rulebook_chain(coors: torch.Tensor, batch: int, shape, strided_specs, want_subm):
    """Every rulebook of one backbone pass (s2d_rulebook_chain_plan -> ONE host read -> s2d_rulebook_chain_fill).
    `strided_specs[l]` = (ksize, stride, padding) of strided conv l; `want_subm[l]` = whether the SubM 3x3x3 map at resolution l
    (0 = input) is needed.  Returns (subm[l] or None for l in 0..n, conv[l] for l in 0..n-1) as Rulebook objects."""
    lib = _lib.load()
    _need_gpu(coors)
    coors = coors.contiguous()
    assert coors.dtype == torch.int32 and coors.dim() == 2 and coors.shape[1] == 4
    dev = coors.device
    n0, nst = int(coors.shape[0]), len(strided_specs) + 1
    ks, st, pd = (_i32_flat([s[j] for s in strided_specs]) for j in range(3))
    shapes = [tuple(int(v) for v in shape)]
    for k, s, p in strided_specs:
        shapes.append(conv_out_shape(shapes[-1], k, s, p))
    kvols = [int(k[0] * k[1] * k[2]) for k, _, _ in strided_specs]
    ws = _ws(lib.s2d_rulebook_chain_workspace_bytes(int(batch), i3(shape), nst - 1, ks, st, pd), dev)
    counts = torch.empty((nst,), dtype=torch.int32, device=dev)
    rec = None
    if PROFILE is not None:
        rec = dict(kernel="rulebook_chain", tag="rulebook", cin=0, cout=0, n_in=n0, kvol=27,
                   start=torch.cuda.Event(enable_timing=True), end=torch.cuda.Event(enable_timing=True),
                   start2=torch.cuda.Event(enable_timing=True), end2=torch.cuda.Event(enable_timing=True))
        rec["start"].record()
    check(lib.s2d_rulebook_chain_plan(_ptr(coors), n0, int(batch), i3(shape), nst - 1, ks, st, pd, _ptr(counts), _ptr(ws), ws.numel(),
                                      _stream()), "s2d_rulebook_chain_plan")
    if rec is not None:
        rec["end"].record()
    got = counts.tolist()   # the one host read of the pass (spconv reads indice_pair_num once per layer)
    if got[-1] != 0:
        raise _lib.S2DError("s2d_rulebook_chain_plan: look-back scan timed out")
    rows = [n0] + got[:-1]

    # one allocation for every map of the pass; each piece 16-byte aligned
    def up4(v):
        return (v + 3) // 4 * 4
    sizes = []
    for l in range(nst):
        sizes.append(("coors", l, 4 * rows[l] if l else 0))
        sizes.append(("subm", l, 27 * rows[l] if want_subm[l] else 0))
        if l + 1 < nst:
            sizes.append(("out", l, kvols[l] * rows[l + 1]))
            sizes.append(("in", l, kvols[l] * rows[l]))
    sizes.append(("cnt", 0, 27 * (2 * nst - 1)))
    sizes.append(("child", 0, 8 * rows[1]))
    flat = torch.empty((sum(up4(sz) for _, _, sz in sizes),), dtype=torch.int32, device=dev)
    view, off = {}, 0
    for kind, l, sz in sizes:
        view[(kind, l)] = flat[off:off + sz]
        off += up4(sz)
    cnt = view[("cnt", 0)].view(2 * nst - 1, 27)
    vp = ctypes.c_void_p

    def ptrs(kind, count):
        return (vp * count)(*[(_ptr(view[(kind, l)]) if view[(kind, l)].numel() else None) for l in range(count)])
    n_rows = (ctypes.c_int64 * (nst - 1))(*rows[1:])
    coors_ptrs = (vp * (nst - 1))(*[(_ptr(view[("coors", l)]) if rows[l] else None) for l in range(1, nst)])
    if rec is not None:
        rec["start2"].record()
    check(lib.s2d_rulebook_chain_fill(_ptr(coors), n0, int(batch), i3(shape), nst - 1, ks, st, pd, n_rows, coors_ptrs, ptrs("subm", nst),
                                      ptrs("out", nst - 1), ptrs("in", nst - 1), _ptr(cnt), _ptr(view[("child", 0)]) if rows[1] else None,
                                      _ptr(ws), ws.numel(), _stream()), "s2d_rulebook_chain_fill")
    if rec is not None and PROFILE is not None:
        rec["end2"].record()
        rec.update(rows=rows, subm_pairs=[cnt[l] if want_subm[l] else None for l in range(nst)],
                   conv_pairs=[cnt[nst + l][:kvols[l]] for l in range(nst - 1)], kvols=kvols)
        PROFILE.append(rec)
    stage_coors = [coors] + [view[("coors", l)].view(rows[l], 4) for l in range(1, nst)]
    subm = [Rulebook(True, 27, rows[l], rows[l], view[("subm", l)].view(27, rows[l]), None, cnt[l], None, shapes[l], coors=stage_coors[l])
            if want_subm[l] else None for l in range(nst)]
    conv = [Rulebook(False, kvols[l], rows[l], rows[l + 1], view[("out", l)].view(kvols[l], rows[l + 1]),
                     view[("in", l)].view(kvols[l], rows[l]), cnt[nst + l][:kvols[l]], stage_coors[l + 1], shapes[l + 1])
            for l in range(nst - 1)]
    return subm, conv


# ------------------------------------------------------------------------------------------------
# sparse conv arithmetic
# ------------------------------------------------------------------------------------------------
PROFILE = None  # bench.py sets this to a list to collect per-launch HIP events (roofline pass)
# "f32": fp32 storage, fp32 MFMA (the parity mode).  "bf16": fp32 storage, MFMA inputs rounded to bf16 in the kernels.
# "s16": bf16 feature storage end to end (csrc/spconv_s16.hip + the row-major bf16 batch norm), fp32 accumulate/statistics.
SPARSE_COMPUTE_DTYPE = "f32"


def set_sparse_compute_dtype(name):
    global SPARSE_COMPUTE_DTYPE
    assert name in ("f32", "bf16", "s16")
    SPARSE_COMPUTE_DTYPE = name


def spconv_gather_gemm(feat: torch.Tensor, weight_kio: torch.Tensor, bias: Optional[torch.Tensor], nbr: torch.Tensor,
                       n_out: int, pair_count: Optional[torch.Tensor] = None, tag: str = "fwd", transpose: bool = False,
                       flip: bool = False) -> torch.Tensor:
    """out[o] = sum_k feat[nbr[k][o]] @ W[k] (+bias), W = weight_kio f32[K,Cin,Cout].
    transpose/flip: multiply by W[k]^T resp. W[K-1-k] instead (data gradient) — on the bf16 path this
    is folded into the weight pre-pack, otherwise done with torch ops here."""
    lib = _lib.load()
    _need_gpu(feat, weight_kio, nbr)
    kvol = weight_kio.shape[0]
    cin, cout = (weight_kio.shape[2], weight_kio.shape[1]) if transpose else (weight_kio.shape[1], weight_kio.shape[2])
    fold = (SPARSE_COMPUTE_DTYPE == "bf16" and lib.s2d_spconv_bf16_supported(cin, cout) and feat.shape[0] > 0)
    if (transpose or flip) and not fold:
        if flip:
            weight_kio = weight_kio.flip(0)
        if transpose:
            weight_kio = weight_kio.transpose(1, 2)
        weight_kio = weight_kio.contiguous()
        transpose = flip = False
    if cin % 16 and cin < 16 and cout % 16 == 0:
        # narrow input layer (5 point features): zero-pad K to one MFMA step so it rides the matrix path
        feat = torch.nn.functional.pad(feat, (0, 16 - cin))
        weight_kio = torch.nn.functional.pad(weight_kio, (0, 0, 0, 16 - cin))
        cin = 16
    feat = feat.contiguous()
    weight_kio = weight_kio.contiguous()
    assert feat.dtype == torch.float32 and weight_kio.dtype == torch.float32
    assert feat.shape[1] == cin and nbr.shape == (kvol, n_out) and nbr.is_contiguous(), (feat.shape, cin, nbr.shape)
    out = torch.empty((n_out, cout), dtype=torch.float32, device=feat.device)
    b = bias.contiguous() if bias is not None else None
    use_bf16 = SPARSE_COMPUTE_DTYPE == "bf16" and lib.s2d_spconv_bf16_supported(cin, cout) and feat.shape[0] > 0
    rec = None
    if PROFILE is not None:
        rec = dict(kernel="spconv_fwd_bf16" if use_bf16 else
                   ("spconv_fwd_mfma" if (cin % 16 == 0 and cout % 16 == 0) else "spconv_fwd_valu"), tag=tag,
                   cin=cin, cout=cout, n_out=int(n_out), kvol=kvol, pairs=pair_count,
                   start=torch.cuda.Event(enable_timing=True), end=torch.cuda.Event(enable_timing=True))
        rec["start"].record()
    if use_bf16:
        packed = torch.empty((kvol * cin * cout,), dtype=torch.bfloat16, device=feat.device)
        check(lib.s2d_spconv_pack_weights_bf16(_ptr(weight_kio), kvol, cin, cout, int(transpose), int(flip), _ptr(packed),
                                               _stream()), "s2d_spconv_pack_weights_bf16")
        check(lib.s2d_spconv_fwd_bf16(_ptr(feat), feat.shape[0], _ptr(packed), _ptr(b), _ptr(nbr), n_out, kvol, cin, cout,
                                      _ptr(out), _stream()), "s2d_spconv_fwd_bf16")
    else:
        check(lib.s2d_spconv_fwd_f32(_ptr(feat), feat.shape[0], _ptr(weight_kio), _ptr(b), _ptr(nbr), n_out, kvol, cin,
                                     cout, _ptr(out), _stream()), "s2d_spconv_fwd_f32")
    if rec is not None:
        rec["end"].record()
        PROFILE.append(rec)
    return out


def spconv_wgrad(feat: torch.Tensor, dout: torch.Tensor, nbr: torch.Tensor, kvol: int,
                 pair_count: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW[k] = sum_o feat[nbr[k][o]]^T dout[o]  ->  f32[K,Cin,Cout]."""
    lib = _lib.load()
    _need_gpu(feat, dout, nbr)
    dout = dout.contiguous()
    n_out, cout = dout.shape
    cin_true = feat.shape[1]
    if cin_true % 16 and cin_true < 16 and cout % 16 == 0:
        feat = torch.nn.functional.pad(feat, (0, 16 - cin_true))
    feat = feat.contiguous()
    cin = feat.shape[1]
    dw = torch.empty((kvol, cin, cout), dtype=torch.float32, device=feat.device)
    ws = _ws_shared(lib.s2d_spconv_wgrad_workspace_bytes(n_out, kvol, cin, cout), feat.device)
    bf = SPARSE_COMPUTE_DTYPE == "bf16"
    fn = lib.s2d_spconv_wgrad_bf16 if bf else lib.s2d_spconv_wgrad_f32
    rec = None
    if PROFILE is not None:
        rec = dict(kernel="spconv_wgrad_bf16" if (bf and cin % 16 == 0 and cout % 16 == 0) else "spconv_wgrad_mfma",
                   tag="wgrad", cin=cin, cout=cout, n_out=int(n_out), kvol=kvol, pairs=pair_count,
                   start=torch.cuda.Event(enable_timing=True), end=torch.cuda.Event(enable_timing=True))
        rec["start"].record()
    check(fn(_ptr(feat), feat.shape[0], _ptr(dout), _ptr(nbr), n_out, kvol, cin, cout, _ptr(dw), _ptr(ws), ws.numel(),
             _stream()), "s2d_spconv_wgrad")
    if rec is not None:
        rec["end"].record()
        PROFILE.append(rec)
    return dw if cin == cin_true else dw[:, :cin_true].contiguous()


# ------------------------------------------------------------------------------------------------
# BatchNorm1d on features
# ------------------------------------------------------------------------------------------------
def bn1d_stats(x: torch.Tensor) -> torch.Tensor:
    """[2C]: per-channel sum and sum of squares (deterministic)."""
    lib = _lib.load()
    _need_gpu(x)
    x = x.contiguous()
    n, c = x.shape
    stats = torch.empty((2 * c,), dtype=torch.float32, device=x.device)
    ws = _ws_shared(lib.s2d_bn1d_workspace_bytes(n, c), x.device)
    check(lib.s2d_bn1d_stats_f32(_ptr(x), n, c, _ptr(stats), _ptr(ws), ws.numel(), _stream()), "s2d_bn1d_stats_f32")
    return stats


def bn1d_finalize_fwd(stats, count, gamma, beta, eps, momentum, running_mean=None, running_var=None, batches_tracked=None):
    """-> packed f32[4,C]: rows mean, invstd, scale, shift (one launch; running stats and the int64
    num_batches_tracked counter updated in place)."""
    lib = _lib.load()
    c = gamma.shape[0]
    out = torch.empty((4, c), dtype=torch.float32, device=stats.device)
    check(lib.s2d_bn1d_finalize_fwd_f32(_ptr(stats), _ptr(count), _ptr(gamma), _ptr(beta), float(eps), float(momentum), c,
                                        _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(out[3]), _ptr(running_mean),
                                        _ptr(running_var), _ptr(batches_tracked), _stream()), "s2d_bn1d_finalize_fwd_f32")
    return out


def bn1d_finalize_bwd(sums_local, sums_global, count, gamma, mean, invstd):
    """-> packed f32[5,C]: rows dgamma, dbeta, a, b, d."""
    lib = _lib.load()
    c = gamma.shape[0]
    out = torch.empty((5, c), dtype=torch.float32, device=gamma.device)
    check(lib.s2d_bn1d_finalize_bwd_f32(_ptr(sums_local), _ptr(sums_global), _ptr(count), _ptr(gamma), _ptr(mean),
                                        _ptr(invstd), c, _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(out[3]),
                                        _ptr(out[4]), _stream()), "s2d_bn1d_finalize_bwd_f32")
    return out


def bn1d_stats_finalize(x, gamma, beta, eps, momentum, running_mean=None, running_var=None, batches_tracked=None):
    """single-GPU: batch statistics + finalisation in two launches -> packed f32[4,C] (mean, invstd, scale, shift)."""
    lib = _lib.load()
    _need_gpu(x)
    x = x.contiguous()
    n, c = x.shape
    out = torch.empty((4, c), dtype=torch.float32, device=x.device)
    ws = _ws_shared(lib.s2d_bn1d_workspace_bytes(n, c), x.device)
    check(lib.s2d_bn1d_stats_finalize_f32(_ptr(x), n, c, _ptr(gamma), _ptr(beta), float(eps), float(momentum), _ptr(out[0]),
                                          _ptr(out[1]), _ptr(out[2]), _ptr(out[3]), _ptr(running_mean), _ptr(running_var),
                                          _ptr(batches_tracked), _ptr(ws), ws.numel(), _stream()), "s2d_bn1d_stats_finalize_f32")
    return out


def bn1d_bwd_reduce_finalize(dy, y, x, relu, gamma, mean, invstd):
    """single-GPU: g = dy*(y>0 if relu), its sums and the finalisation -> (g, packed f32[5,C]: dgamma, dbeta, a, b, d)."""
    lib = _lib.load()
    dy = dy.contiguous()
    x = x.contiguous()
    n, c = x.shape
    g = torch.empty_like(x)
    out = torch.empty((5, c), dtype=torch.float32, device=x.device)
    ws = _ws_shared(lib.s2d_bn1d_workspace_bytes(n, c), x.device)
    check(lib.s2d_bn1d_bwd_reduce_finalize_f32(_ptr(dy), _ptr(y), _ptr(x), int(relu), n, c, _ptr(gamma), _ptr(mean),
                                               _ptr(invstd), _ptr(g), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(out[3]),
                                               _ptr(out[4]), _ptr(ws), ws.numel(), _stream()),
          "s2d_bn1d_bwd_reduce_finalize_f32")
    return g, out


def bn1d_apply(x, scale, shift, residual=None, relu=False):
    lib = _lib.load()
    _need_gpu(x, scale, shift, residual)
    x = x.contiguous()
    n, c = x.shape
    y = torch.empty_like(x)
    r = residual.contiguous() if residual is not None else None
    check(lib.s2d_bn1d_apply_f32(_ptr(x), _ptr(scale.contiguous()), _ptr(shift.contiguous()), _ptr(r), int(relu), n, c,
                                 _ptr(y), _stream()), "s2d_bn1d_apply_f32")
    return y


def bn1d_bwd_reduce(dy, y, x, relu, want_g=True):
    """g = dy * (y>0 if relu); returns (g or None, sums[2C] = [sum g, sum g*x])."""
    lib = _lib.load()
    _need_gpu(dy, x)
    dy = dy.contiguous()
    x = x.contiguous()
    n, c = x.shape
    g = torch.empty_like(x) if want_g else None
    sums = torch.empty((2 * c,), dtype=torch.float32, device=x.device)
    ws = _ws_shared(lib.s2d_bn1d_workspace_bytes(n, c), x.device)
    check(lib.s2d_bn1d_bwd_reduce_f32(_ptr(dy), _ptr(y), _ptr(x), int(relu), n, c, _ptr(g), _ptr(sums), _ptr(ws),
                                      ws.numel(), _stream()), "s2d_bn1d_bwd_reduce_f32")
    return g, sums


def bn1d_bwd_apply(g, x, a, b, d):
    lib = _lib.load()
    n, c = x.shape
    dx = torch.empty_like(x)
    check(lib.s2d_bn1d_bwd_apply_f32(_ptr(g), _ptr(x), _ptr(a.contiguous()), _ptr(b.contiguous()), _ptr(d.contiguous()),
                                     n, c, _ptr(dx), _stream()), "s2d_bn1d_bwd_apply_f32")
    return dx


# ------------------------------------------------------------------------------------------------
# densify
# ------------------------------------------------------------------------------------------------
def densify(feat, coors, batch, shape):
    lib = _lib.load()
    _need_gpu(feat, coors)
    feat = feat.contiguous()
    n, c = feat.shape
    out = torch.empty((batch, c, shape[0], shape[1], shape[2]), dtype=torch.float32, device=feat.device)
    check(lib.s2d_densify_fwd_f32(_ptr(feat), _ptr(coors.contiguous()), n, batch, i3(shape), c, _ptr(out), _stream()),
          "s2d_densify_fwd_f32")
    return out


def densify_bwd(dout, coors, batch, shape, c):
    lib = _lib.load()
    dout = dout.contiguous()
    n = coors.shape[0]
    dfeat = torch.empty((n, c), dtype=torch.float32, device=dout.device)
    check(lib.s2d_densify_bwd_f32(_ptr(dout), _ptr(coors.contiguous()), n, batch, i3(shape), c, _ptr(dfeat), _stream()),
          "s2d_densify_bwd_f32")
    return dfeat


def densify_bev_bf16(feat, coors, batch, shape):
    """-> bf16 [batch, C*D, H, W] in channels_last memory (== dense().view(N, C*D, H, W) cast to bf16, NHWC)"""
    lib = _lib.load()
    _need_gpu(feat, coors)
    feat = feat.contiguous()
    n, c = feat.shape
    out = torch.empty((batch, c * shape[0], shape[1], shape[2]), dtype=torch.bfloat16, device=feat.device,
                      memory_format=torch.channels_last)
    assert feat.dtype in (torch.float32, torch.bfloat16)
    check(lib.s2d_densify_bev_fwd_bf16(_ptr(feat), int(feat.dtype == torch.bfloat16), _ptr(coors.contiguous()), n, batch,
                                       i3(shape), c, _ptr(out), _stream()), "s2d_densify_bev_fwd_bf16")
    return out


def densify_bev_bf16_bwd(dout, coors, batch, shape, c, feat_dtype=torch.float32):
    lib = _lib.load()
    dout = dout.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    n = coors.shape[0]
    dfeat = torch.empty((n, c), dtype=feat_dtype, device=dout.device)
    check(lib.s2d_densify_bev_bwd_bf16(_ptr(dout), _ptr(coors.contiguous()), n, batch, i3(shape), c, _ptr(dfeat),
                                       int(feat_dtype == torch.bfloat16), _stream()), "s2d_densify_bev_bwd_bf16")
    return dfeat


# ------------------------------------------------------------------------------------------------
# bf16-storage sparse conv ("s16" path, csrc/spconv_s16.hip)
# ------------------------------------------------------------------------------------------------
_zero_pages = {}


def zero_page(device):
    z = _zero_pages.get(device)
    if z is None:
        z = torch.zeros(64, dtype=torch.uint8, device=device)
        _zero_pages[device] = z
    return z


def spconv_s16_pack(weight_kio, n_out, transpose=False, flip=False):
    """fp32 [K, cin, cout] (or the forward layer's [K, cout, cin] with transpose) -> the kernel's bf16 weight image
    for a launch over n_out rows; returns (packed, kvol, cin, cout) of the packed operand."""
    lib = _lib.load()
    kvol = weight_kio.shape[0]
    cin, cout = (weight_kio.shape[2], weight_kio.shape[1]) if transpose else (weight_kio.shape[1], weight_kio.shape[2])
    w = weight_kio.detach().float().contiguous()
    packed = torch.empty(lib.s2d_spconv_s16_packed_elems(kvol, cin, cout), dtype=torch.bfloat16, device=w.device)
    check(lib.s2d_spconv_s16_pack_weights(_ptr(w), kvol, cin, cout, int(transpose), int(flip), int(n_out), _ptr(packed),
                                          _stream()), "s2d_spconv_s16_pack_weights")
    return packed, kvol, cin, cout


def spconv_s16_pack_pair(weight_kio, n_out_fwd, n_out_dgrad, flip_dgrad, with_launch=False):
    """fp32 [K, cin, cout] -> the forward image AND the data-gradient image ([cout -> cin], offsets mirrored when flip_dgrad) in one
    launch; returns ((packed, kvol, cin, cout), (packed_d, kvol, cout, cin)) as spconv_s16_pack does for each"""
    lib = _lib.load()
    kvol, cin, cout = weight_kio.shape
    w = weight_kio.detach().float().contiguous()
    pf = torch.empty(lib.s2d_spconv_s16_packed_elems(kvol, cin, cout), dtype=torch.bfloat16, device=w.device)
    pd = torch.empty(lib.s2d_spconv_s16_packed_elems(kvol, cout, cin), dtype=torch.bfloat16, device=w.device)
    launch = lambda: check(lib.s2d_spconv_s16_pack_weights_pair(_ptr(w), kvol, cin, cout, int(flip_dgrad), int(n_out_fwd), int(n_out_dgrad), _ptr(pf),
                                                                _ptr(pd), _stream()), "s2d_spconv_s16_pack_weights_pair")
    launch()
    if with_launch:   # (the launch, the tensor it reads): dense2d.register_repack re-runs it in place after the optimizer step
        return (pf, kvol, cin, cout), (pd, kvol, cout, cin), launch, w
    return (pf, kvol, cin, cout), (pd, kvol, cout, cin)


def spconv_s16_run(feat, packed, kvol, cin, cout, bias, nbr, n_out, pair_count=None, tag="fwd", bn_stats=False):
    """bn_stats: also return the per-workgroup (sum, sum of squares) rows [tiles, 2, cout] of the stored output - the statistics pass of
    the BatchNorm1d that follows, done in the epilogue"""
    lib = _lib.load()
    assert feat.dtype == torch.bfloat16 and feat.shape[1] == cin and feat.is_contiguous()
    out = torch.empty((n_out, cout), dtype=torch.bfloat16, device=feat.device)
    partial = None
    if bn_stats and n_out > 0:
        partial = torch.empty((lib.s2d_spconv_s16_stats_tiles(int(n_out), kvol, cin, cout), 2, cout), dtype=torch.float32, device=feat.device)
    rec = None
    if PROFILE is not None:
        rec = dict(kernel="spconv_fwd_s16", tag=tag, cin=cin, cout=cout, n_out=int(n_out), kvol=kvol, pairs=pair_count,
                   elem_bytes=2, start=torch.cuda.Event(enable_timing=True), end=torch.cuda.Event(enable_timing=True))
        rec["start"].record()
    check(lib.s2d_spconv_s16_fwd_stats(_ptr(feat), feat.shape[0], _ptr(packed), _ptr(bias), _ptr(nbr), int(n_out), kvol, cin, cout,
                                       _ptr(zero_page(feat.device)), _ptr(out), _ptr(partial), _stream()), "s2d_spconv_s16_fwd_stats")
    if rec is not None:
        rec["end"].record()
        PROFILE.append(rec)
    return (out, partial) if bn_stats else out


# submanifold 64 -> 64 / 128 -> 128 layers over mask-sorted rows (csrc/rulebook_sort.hip).  OPT-IN (S2D_RG_SORTED=1 / set_sorted_rows): on the
# benchmark scene it is 5-10 % slower than the plain kernel (profiles/r06_sparse_sorted_rows_ab.txt) - the gathers, not the MFMAs, bound it.
SORTED_ROWS = os.environ.get("S2D_RG_SORTED", "0") not in ("0", "")


def set_sorted_rows(on):
    """switch the sorted-row form on / off in the library and here; returns the previous setting.  The 64 -> 64 layers change their weight
    image with it, so every cached image is dropped."""
    global SORTED_ROWS
    from .dense2d import clear_pack_cache
    was = bool(_lib.load().s2d_spconv_s16_set_sorted_rows(int(bool(on))))
    SORTED_ROWS = bool(on)
    clear_pack_cache()
    return was



def rulebook_sorted_rows(rb):
    """(perm i32[n], pmask i32[n] (bit pattern of the u32 masks), nbr_perm i32[K, n]) of a submanifold rulebook: its rows grouped by
    neighbour mask, built once per rulebook (three launches, no host read) and kept on it - every layer and both passes of the stage reuse it"""
    cached = getattr(rb, "_sorted_rows", None)
    if cached is not None:
        return cached
    lib = _lib.load()
    nbr = rb.nbr_out
    n = int(rb.n_out)
    dev = nbr.device
    assert rb.subm and nbr.dtype == torch.int32 and nbr.is_contiguous() and nbr.shape == (rb.kvol, n)
    perm = torch.empty((n,), dtype=torch.int32, device=dev)
    pmask = torch.empty((n,), dtype=torch.int32, device=dev)
    nbr_perm = torch.empty_like(nbr)
    ws = _ws(lib.s2d_rulebook_sort_workspace_bytes(n), dev)
    check(lib.s2d_rulebook_sort_by_mask(_ptr(nbr), rb.kvol, n, _ptr(perm), _ptr(pmask), _ptr(nbr_perm), _ptr(ws), ws.numel(), _stream()),
          "s2d_rulebook_sort_by_mask")
    rb._sorted_rows = (perm, pmask, nbr_perm)
    return rb._sorted_rows


def spconv_s16_sorted_ok(rb, kvol, cin, cout, n_out):
    return bool(SORTED_ROWS and rb is not None and rb.subm and n_out == rb.n_out and n_out > 0
                and _lib.load().s2d_spconv_s16_sorted_supported(int(kvol), int(cin), int(cout)))


def spconv_s16_run_sorted(feat, packed, kvol, cin, cout, bias, rb, tag="fwd", bn_stats=False):
    """spconv_s16_run over the mask-sorted rows of the submanifold rulebook `rb` (same packed image, same outputs in the canonical row order)"""
    lib = _lib.load()
    n_out = int(rb.n_out)
    assert feat.dtype == torch.bfloat16 and feat.shape[1] == cin and feat.is_contiguous()
    perm, pmask, nbr_perm = rulebook_sorted_rows(rb)
    out = torch.empty((n_out, cout), dtype=torch.bfloat16, device=feat.device)
    partial = None
    if bn_stats:
        partial = torch.empty((lib.s2d_spconv_s16_stats_tiles(n_out, kvol, cin, cout), 2, cout), dtype=torch.float32, device=feat.device)
    rec = None
    if PROFILE is not None:
        rec = dict(kernel="spconv_fwd_s16", tag=tag, cin=cin, cout=cout, n_out=n_out, kvol=kvol, pairs=rb.pair_count, sorted_rows=True,
                   elem_bytes=2, start=torch.cuda.Event(enable_timing=True), end=torch.cuda.Event(enable_timing=True))
        rec["start"].record()
    check(lib.s2d_spconv_s16_fwd_sorted(_ptr(feat), feat.shape[0], _ptr(packed), _ptr(bias), _ptr(nbr_perm), _ptr(perm), _ptr(pmask), n_out, kvol,
                                        cin, cout, _ptr(out), _ptr(partial), _stream()), "s2d_spconv_s16_fwd_sorted")
    if rec is not None:
        rec["end"].record()
        PROFILE.append(rec)
    return (out, partial) if bn_stats else out


def col_sums_bf16(x):
    """per-column fp32 sums of a row-major bf16 matrix [n, c] (c % 8 == 0): bias gradients"""
    lib = _lib.load()
    n, c = x.shape
    stats = torch.empty((2 * c,), dtype=torch.float32, device=x.device)
    ws = _ws_shared(lib.s2d_bnrow_workspace_bytes(n, c), x.device)
    check(lib.s2d_bnrow_stats_bf16(_ptr(x), n, c, _ptr(stats), 0, _ptr(ws), ws.numel(), _stream()), "s2d_bnrow_stats_bf16")
    return stats[:c]


def spconv_s16(feat, weight_kio, bias, nbr, n_out, transpose=False, flip=False, pair_count=None, tag="fwd"):
    """feat bf16 [n_in, cin]; weight_kio fp32 [K, cin, cout] (or [K, cout, cin] of the forward layer when transpose:
    then the result is the data gradient w.r.t. that layer's input); -> bf16 [n_out, cout of the packed operand]."""
    _need_gpu(feat, weight_kio, nbr)
    packed, kvol, cin, cout = spconv_s16_pack(weight_kio, n_out, transpose, flip)
    b = None if bias is None else bias.detach().float().contiguous()
    return spconv_s16_run(feat.contiguous(), packed, kvol, cin, cout, b, nbr, n_out, pair_count, tag)


def spconv_s16_wgrad(feat, dout, nbr, kvol, pair_count=None):
    """dW[k] = gather(feat, nbr[k])^T @ dout for bf16 feat [n_in, cin], dout [n_out, cout] -> fp32 [K, cin, cout]"""
    lib = _lib.load()
    assert feat.dtype == torch.bfloat16 and dout.dtype == torch.bfloat16
    feat, dout = feat.contiguous(), dout.contiguous()
    cin, cout, n_out = feat.shape[1], dout.shape[1], dout.shape[0]
    dw = torch.empty((kvol, cin, cout), dtype=torch.float32, device=feat.device)
    ws = _ws_shared(lib.s2d_spconv_wgrad_workspace_bytes(n_out, kvol, cin, cout), feat.device)
    rec = None
    if PROFILE is not None:
        rec = dict(kernel="spconv_wgrad_s16", tag="wgrad", cin=cin, cout=cout, n_out=int(n_out), kvol=kvol, pairs=pair_count,
                   elem_bytes=2, start=torch.cuda.Event(enable_timing=True), end=torch.cuda.Event(enable_timing=True))
        rec["start"].record()
    check(lib.s2d_spconv_s16_wgrad(_ptr(feat), feat.shape[0], _ptr(dout), _ptr(nbr), n_out, kvol, cin, cout, _ptr(dw), _ptr(ws),
                                   ws.numel(), _stream()), "s2d_spconv_s16_wgrad")
    if rec is not None:
        rec["end"].record()
        PROFILE.append(rec)
    return dw
