"""TEST INFRASTRUCTURE ONLY — an oracle-backed stand-in for `sparse2dense_amd.hip_ops`.

`install(monkeypatch)` swaps every HIP launcher for a plain-torch implementation built from
`oracle/` so that the HOST logic of the product (module wiring, autograd functions, rulebook
caching/planning, SyncBN + DDP data-parallel path over gloo) can be exercised by `-m "not gpu"`
tests in a container without a GPU.  Nothing under sparse2dense_amd/ imports this file; the
product path itself has no CPU fallback.

Device: every stand-in here is device-agnostic torch (rulebooks / voxelization: numpy on the host, results
moved to the tensors' device).  The GPU parity tests run this "oracle stack" in float64 ON the MI355X
(`oracle_stack(device="cuda:0")`): torch's own float64 kernels do the arithmetic - a 169 s host run becomes
seconds - and for the whole time the stack runs `sparse2dense_amd._lib.load` RAISES, so no s2d kernel can take
part in its own reference (a product dispatch that would reach one fails the test loudly).
tests/test_oracle_device.py holds device run == host run.
"""
import contextlib
import sys

import numpy as np
import torch

from oracle import spconv_ref as R
from oracle import voxelize as OV
from sparse2dense_amd import hip_ops as H


# bf16-STORAGE emulation of the sparse stack (the product's benchmarked "s16" mode): with STORAGE_BF16[0] set, every tensor the product
# writes as bf16 rows - conv outputs (forward and data gradient), fused BN(+residual)(+ReLU) outputs and their gradients - and every
# operand its kernels read as bf16 (features, packed weights) is rounded to bf16 here, while all arithmetic stays in the tensors'
# dtype (float64 in the calibration runs): the price of the storage type alone, with exact accumulation.
STORAGE_BF16 = [False]


def _st(x):
    return x.to(torch.bfloat16).to(x.dtype) if STORAGE_BF16[0] and x is not None else x


def _pairs_to_maps(pairs, n_in, n_out, want_in):
    kvol = len(pairs)
    nbr_out = np.full((kvol, n_out), -1, np.int32)
    nbr_in = np.full((kvol, n_in), -1, np.int32) if want_in else None
    cnt = np.zeros((kvol,), np.int32)
    for k, (i_in, i_out) in enumerate(pairs):
        nbr_out[k, i_out] = i_in
        if want_in:
            nbr_in[k, i_in] = i_out
        cnt[k] = len(i_in)
    return nbr_out, nbr_in, cnt


def _np(t):
    return t.detach().cpu().numpy()


def voxelize(points, voxel_size, coors_range, max_points, max_voxels, with_mean=True):
    dev = points.device
    v, c, n = OV.points_to_voxel(_np(points), voxel_size, coors_range, max_points, max_voxels)
    mean = torch.from_numpy(OV.voxel_mean(v, n)).to(dev) if with_mean else None
    return torch.from_numpy(v).to(dev), torch.from_numpy(c).to(dev), torch.from_numpy(n).to(dev), mean


def voxelize_async(points, voxel_size, coors_range, max_points, max_voxels, with_mean=True):
    v, c, n, m = voxelize(points, voxel_size, coors_range, max_points, max_voxels, with_mean)
    return v, c, n, m, torch.tensor([c.shape[0]], dtype=torch.int32, device=points.device)


def build_subm_rulebook(coors, batch, shape, ksize, dilation=(1, 1, 1)):
    c, dev = _np(coors), coors.device
    pairs = R.rulebook_subm(c, tuple(shape), ksize, dilation)
    nbr_out, _, cnt = _pairs_to_maps(pairs, c.shape[0], c.shape[0], False)
    return H.Rulebook(True, len(pairs), c.shape[0], c.shape[0], torch.from_numpy(nbr_out).to(dev), None, torch.from_numpy(cnt).to(dev),
                      None, tuple(int(s) for s in shape))


def build_conv_rulebook(coors, batch, shape, ksize, stride, padding, dilation=(1, 1, 1)):
    c, dev = _np(coors), coors.device
    oc, oshape, pairs = R.rulebook_conv(c, tuple(shape), ksize, stride, padding, dilation)
    nbr_out, nbr_in, cnt = _pairs_to_maps(pairs, c.shape[0], oc.shape[0], True)
    return H.Rulebook(False, len(pairs), c.shape[0], oc.shape[0], torch.from_numpy(nbr_out).to(dev), torch.from_numpy(nbr_in).to(dev),
                      torch.from_numpy(cnt).to(dev), torch.from_numpy(oc).to(dev), oshape)


def spconv_gather_gemm(feat, weight_kio, bias, nbr, n_out, pair_count=None, tag="fwd", transpose=False, flip=False):
    if flip:
        weight_kio = weight_kio.flip(0)
    if transpose:
        weight_kio = weight_kio.transpose(1, 2)
    feat, weight_kio = _st(feat), _st(weight_kio)
    out = feat.new_zeros((n_out, weight_kio.shape[2]))
    for k in range(weight_kio.shape[0]):
        o = (nbr[k] >= 0).nonzero().squeeze(1)
        if o.numel():
            out.index_add_(0, o, feat[nbr[k][o].long()] @ weight_kio[k])
    return _st(out + bias if bias is not None else out)


def spconv_wgrad(feat, dout, nbr, kvol, pair_count=None):
    feat, dout = _st(feat), _st(dout)
    dw = feat.new_zeros((kvol, feat.shape[1], dout.shape[1]))
    for k in range(kvol):
        o = (nbr[k] >= 0).nonzero().squeeze(1)
        if o.numel():
            dw[k] = feat[nbr[k][o].long()].t() @ dout[o]
    return dw


def bn1d_stats(x):
    return torch.cat([x.sum(0), (x * x).sum(0)])


def bn1d_finalize_fwd(stats, count, gamma, beta, eps, momentum, running_mean=None, running_var=None, batches_tracked=None):
    if batches_tracked is not None:
        batches_tracked += 1
    c = gamma.shape[0]
    mean = stats[:c] / count
    var = (stats[c:] / count - mean * mean).clamp(min=0)
    invstd = torch.rsqrt(var + eps)
    scale = gamma.detach() * invstd
    shift = beta.detach() - mean * scale
    if running_mean is not None:
        with torch.no_grad():
            unbiased = var * (count / (count - 1).clamp(min=1))
            running_mean.mul_(1 - momentum).add_(mean * momentum)
            running_var.mul_(1 - momentum).add_(unbiased * momentum)
    return torch.stack([mean, invstd, scale, shift])


def bn1d_finalize_bwd(sums_local, sums_global, count, gamma, mean, invstd):
    c = gamma.shape[0]
    dbeta = sums_local[:c]
    dgamma = invstd * (sums_local[c:] - mean * sums_local[:c])
    sg = sums_global[:c]
    dg_all = invstd * (sums_global[c:] - mean * sg)
    a = gamma.detach() * invstd
    b = -(a * invstd) * dg_all / count
    d = -(a * sg) / count - b * mean
    return torch.stack([dgamma, dbeta, a, b, d])


def bn1d_stats_finalize(x, gamma, beta, eps, momentum, running_mean=None, running_var=None, batches_tracked=None):
    count = torch.full((1,), float(x.shape[0]), dtype=x.dtype, device=x.device)
    return bn1d_finalize_fwd(bn1d_stats(x), count, gamma, beta, eps, momentum, running_mean, running_var, batches_tracked)


def bn1d_bwd_reduce_finalize(dy, y, x, relu, gamma, mean, invstd):
    g, sums = bn1d_bwd_reduce(dy, y, x, relu)
    count = torch.full((1,), float(x.shape[0]), dtype=x.dtype, device=x.device)
    return g, bn1d_finalize_bwd(sums, sums, count, gamma, mean, invstd)


def bn1d_apply(x, scale, shift, residual=None, relu=False):
    y = x * scale + shift
    if residual is not None:
        y = y + residual
    return _st(y.relu() if relu else y)


def bn1d_bwd_reduce(dy, y, x, relu, want_g=True):
    g = dy * (y > 0) if relu else dy.clone()
    return g, torch.cat([g.sum(0), (g * x).sum(0)])


def bn1d_bwd_apply(g, x, a, b, d):
    return _st(a * g + b * x + d)


def densify(feat, coors, batch, shape):
    return R.densify(feat, _np(coors), tuple(shape), batch)


def densify_bwd(dout, coors, batch, shape, c):
    i = coors.long().to(dout.device)
    return dout[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]].contiguous()


_NAMES = ["voxelize", "voxelize_async", "build_subm_rulebook", "build_conv_rulebook", "spconv_gather_gemm", "spconv_wgrad", "bn1d_stats",
          "bn1d_finalize_fwd", "bn1d_finalize_bwd", "bn1d_stats_finalize", "bn1d_bwd_reduce_finalize", "bn1d_apply", "bn1d_bwd_reduce", "bn1d_bwd_apply", "densify", "densify_bwd"]


class OracleReachedHip(AssertionError):
    pass


def _no_hip_library():
    raise OracleReachedHip("the oracle stack reached a HIP launcher (sparse2dense_amd._lib.load): a product dispatch took the device path "
                           "for a tensor of the float64 reference run - its predicate must name the dtypes the kernel supports")


def install(monkeypatch=None, guard=False):
    """Patch sparse2dense_amd.hip_ops (and the CUDA-only guard of FeatureBatchNorm1d) in place.  guard=True (the device runs of the
    oracle stack): additionally make the library loader raise and report the rulebook chain as unsupported, so that every
    route into an s2d kernel is closed while the reference is computed."""
    import sparse2dense_amd.spconv as sp
    from sparse2dense_amd import _lib
    g = globals()
    for n in _NAMES:
        if monkeypatch is not None:
            monkeypatch.setattr(H, n, g[n])
        else:
            setattr(H, n, g[n])
    if monkeypatch is not None:
        monkeypatch.setattr(sp.FeatureBatchNorm1d, "_REQUIRE_CUDA", False)
    else:
        sp.FeatureBatchNorm1d._REQUIRE_CUDA = False
    if guard:
        assert monkeypatch is not None, "the guarded installation must be undone: pass a MonkeyPatch"
        monkeypatch.setattr(_lib, "load", _no_hip_library)
        monkeypatch.setattr(H, "rulebook_chain_supported", lambda *a, **k: False)


@contextlib.contextmanager
def oracle_stack(device="cpu", storage_bf16=False):
    """`with oracle_stack("cuda:0"):` - the product's host code computes with the oracle's launchers inside the block; on a GPU device
    no s2d kernel can be reached (see the module docstring).  Everything is restored on exit."""
    import pytest
    mp = pytest.MonkeyPatch()
    try:
        install(mp, guard=torch.device(device).type == "cuda")
        if storage_bf16:
            mp.setattr(sys.modules[__name__], "STORAGE_BF16", [True])
        yield mp
    finally:
        mp.undo()


def to_device(obj, device, dtype=None):
    """an example dict / list / tensor moved to `device`; floating tensors cast to `dtype` when given"""
    if torch.is_tensor(obj):
        out = obj.to(device=device, dtype=dtype) if (dtype is not None and obj.is_floating_point()) else obj.to(device)
        # always a NEW tensor object: python attributes hung on an example's tensors by the product's data pipeline (a pre-built HIP
        # geometry plan, data.attach_geometry) must not travel into the oracle's run
        return out.clone() if out is obj else out
    if isinstance(obj, dict):
        return {k: to_device(v, device, dtype) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_device(v, device, dtype) for v in obj)
    return obj
