"""ORACLE (test infrastructure only) — CPU restatement of the sparse-convolution arithmetic.

PARITY UNPINNED: the arithmetic lives in third-party `spconv` (traveller59/spconv v1.x, commit
73427720a539caf9a44ec58abe3af7aa9ddb8e39, /root/reference/docs/INSTALL.md:12,65-72), which is
neither under /root/reference nor installed here, and the reference has no tests.  This file
restates spconv v1.x's published algorithm and is anchored on the reference's call sites:
  /root/reference/det3d/models/backbones/scn.py:16-26   (conv3x3 -> SubMConv3d k3 p1)
  /root/reference/det3d/models/backbones/scn.py:42-85   (SparseBasicBlock)
  /root/reference/det3d/models/backbones/scn.py:88-185  (SpMiddleResNetFHD)
  /root/reference/det3d/models/backbones/scn.py:187-289 (SpMiddleFHD)
  /root/reference/det3d/models/detectors/voxelnet.py:203-215 (SparseConvTensor(...).dense())
It is cross-checked (tests/test_oracle_spconv.py) against an INDEPENDENT dense formulation
(`F.conv3d` on the densified tensor, masked to the active set) and against brute-force loops.

spconv v1.x semantics restated here:
  * out extent = floor((in + 2p - d(k-1) - 1)/s) + 1 per axis; SubM keeps the input extent;
  * SubM: output sites = input sites in input order; centred (pad = k//2, stride 1) whatever
    `padding=` says (scn.py:105 passes none, scn.py:18-26 passes 1 — same result);
  * regular conv: an output site exists wherever >=1 active input reaches it; spconv's GPU path
    numbers outputs by sorted linear (b,z,y,x) index — that is the canonical order used here;
  * kernel offset index = row-major over (kz,ky,kx); weight layout [kD,kH,kW,Cin,Cout];
  * out[o] = sum_k W[k]^T in[j] over pairs (j -> o, k) where pos(j) = pos(o)*s - p + k*d (+bias);
  * .dense(): zeros[B,D,H,W,C] scatter rows, permute to [B,C,D,H,W].

Device: rulebooks are numpy on the host, always.  The feature arithmetic is plain torch (index_select / mm /
index_add) and runs wherever the feature tensor lives: the CPU by default; the GPU parity tests also run it in
float64 on the device through torch's own kernels (no s2d kernel can be reached: tests/cpu_backend.py guards the
library loader while the oracle runs), with tests/test_oracle_device.py holding device == CPU.
"""
import itertools

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


def _triple(v):
    if isinstance(v, (list, tuple, np.ndarray)):
        assert len(v) == 3
        return tuple(int(x) for x in v)
    return (int(v),) * 3


def conv_out_shape(shape, ksize, stride, padding, dilation=1):
    ksize, stride, padding, dilation = map(_triple, (ksize, stride, padding, dilation))
    return tuple((int(shape[i]) + 2 * padding[i] - dilation[i] * (ksize[i] - 1) - 1) // stride[i] + 1
                 for i in range(3))


def _lin(coors, shape):
    c = coors.astype(np.int64)
    return ((c[:, 0] * shape[0] + c[:, 1]) * shape[1] + c[:, 2]) * shape[2] + c[:, 3]


def rulebook_subm(coors, shape, ksize, dilation=1):
    """Submanifold rulebook.  coors i32[N,4] (b,z,y,x).  Returns list over K offsets of
    (in_idx i64[n_k], out_idx i64[n_k]); pair (j -> i, k) means pos(j) = pos(i) + (k - k//2)*d."""
    ksize, dilation = _triple(ksize), _triple(dilation)
    coors = np.asarray(coors)
    n = coors.shape[0]
    keys = _lin(coors, shape)
    order = np.argsort(keys, kind="stable")
    skeys = keys[order]
    pairs = []
    for kz, ky, kx in itertools.product(range(ksize[0]), range(ksize[1]), range(ksize[2])):
        off = np.array([(kz - ksize[0] // 2) * dilation[0], (ky - ksize[1] // 2) * dilation[1],
                        (kx - ksize[2] // 2) * dilation[2]])
        nb = coors.astype(np.int64).copy()
        nb[:, 1:] += off
        ok = np.all((nb[:, 1:] >= 0) & (nb[:, 1:] < np.asarray(shape)), axis=1)
        nkeys = _lin(nb, shape)
        pos = np.searchsorted(skeys, nkeys)
        pos_c = np.minimum(pos, max(n - 1, 0))
        found = ok & (pos < n)
        if n:
            found &= skeys[pos_c] == nkeys
        out_idx = np.nonzero(found)[0]
        in_idx = order[pos_c[found]] if n else np.zeros((0,), np.int64)
        pairs.append((in_idx.astype(np.int64), out_idx.astype(np.int64)))
    return pairs


def rulebook_conv(coors, shape, ksize, stride, padding, dilation=1):
    """Regular (strided) sparse-conv rulebook.  Returns (out_coors i32[M,4] sorted by linear
    index, out_shape, pairs) with pairs as in rulebook_subm but indexing the new output rows."""
    ksize, stride, padding, dilation = map(_triple, (ksize, stride, padding, dilation))
    coors = np.asarray(coors)
    out_shape = conv_out_shape(shape, ksize, stride, padding, dilation)
    cand = []
    for k, (kz, ky, kx) in enumerate(itertools.product(range(ksize[0]), range(ksize[1]), range(ksize[2]))):
        kk = np.array([kz * dilation[0], ky * dilation[1], kx * dilation[2]])
        num = coors[:, 1:].astype(np.int64) + np.asarray(padding) - kk
        ok = np.all(num % np.asarray(stride) == 0, axis=1)
        o = num // np.asarray(stride)
        ok &= np.all((o >= 0) & (o < np.asarray(out_shape)), axis=1)
        j = np.nonzero(ok)[0]
        oc = np.concatenate([coors[j, :1].astype(np.int64), o[j]], axis=1)
        cand.append((j, oc))
    allk = np.concatenate([_lin(oc, out_shape) for _, oc in cand]) if cand else np.zeros((0,), np.int64)
    ukeys = np.unique(allk)
    m = ukeys.shape[0]
    out_coors = np.zeros((m, 4), np.int32)
    rem = ukeys.copy()
    out_coors[:, 3] = rem % out_shape[2]; rem //= out_shape[2]
    out_coors[:, 2] = rem % out_shape[1]; rem //= out_shape[1]
    out_coors[:, 1] = rem % out_shape[0]; rem //= out_shape[0]
    out_coors[:, 0] = rem
    pairs = []
    for j, oc in cand:
        o_idx = np.searchsorted(ukeys, _lin(oc, out_shape))
        pairs.append((j.astype(np.int64), o_idx.astype(np.int64)))
    return out_coors, out_shape, pairs


def rulebook_bruteforce(coors, shape, ksize, stride, padding, dilation, subm):
    """Pure-Python loops (tiny inputs): second restatement, checks the vectorised builders.
    Returns (out_coors, pairs-as-set {(k, in, out)})."""
    ksize, stride, padding, dilation = map(_triple, (ksize, stride, padding, dilation))
    coors = [tuple(int(v) for v in c) for c in np.asarray(coors)]
    if subm:
        padding = tuple(k // 2 for k in ksize)
        stride = (1, 1, 1)
        out_shape = tuple(shape)
        out_coors = coors
    else:
        out_shape = conv_out_shape(shape, ksize, stride, padding, dilation)
        outs = set()
        for (b, z, y, x) in coors:
            for kz, ky, kx in itertools.product(range(ksize[0]), range(ksize[1]), range(ksize[2])):
                num = (z + padding[0] - kz * dilation[0], y + padding[1] - ky * dilation[1],
                       x + padding[2] - kx * dilation[2])
                if any(num[i] % stride[i] for i in range(3)):
                    continue
                o = tuple(num[i] // stride[i] for i in range(3))
                if all(0 <= o[i] < out_shape[i] for i in range(3)):
                    outs.add((b,) + o)
        out_coors = sorted(outs)
    lut = {c: i for i, c in enumerate(coors)}
    pairs = set()
    for oi, (b, z, y, x) in enumerate(out_coors):
        for k, (kz, ky, kx) in enumerate(itertools.product(range(ksize[0]), range(ksize[1]), range(ksize[2]))):
            ip = (b, z * stride[0] - padding[0] + kz * dilation[0], y * stride[1] - padding[1] + ky * dilation[1],
                  x * stride[2] - padding[2] + kx * dilation[2])
            j = lut.get(ip)
            if j is not None:
                pairs.add((k, j, oi))
    return np.asarray(out_coors, np.int32).reshape(-1, 4), pairs


def pairs_to_set(pairs):
    s = set()
    for k, (i_in, i_out) in enumerate(pairs):
        s.update((k, int(a), int(b)) for a, b in zip(i_in, i_out))
    return s


# ----------------------------------------------------------------------------------------------
# bf16-STORAGE emulation (the product's "s16" mode keeps sparse features in bf16 between kernels,
# fp32 accumulation / statistics).  `with bf16_storage():` makes the module-level restatement below
# round features to bf16 at exactly the points where the product writes a bf16 tensor - the backbone
# input, every conv output, every fused BN(+residual)(+ReLU) output - and round the gradients
# flowing back through those points the same way, in whatever dtype (fp32 / fp64) the oracle runs.
# ----------------------------------------------------------------------------------------------
_STORAGE_BF16 = False


class _RoundBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def _store(x):
    return _RoundBf16.apply(x) if _STORAGE_BF16 else x


class bf16_storage:
    def __enter__(self):
        global _STORAGE_BF16
        self.prev, _STORAGE_BF16 = _STORAGE_BF16, True

    def __exit__(self, *a):
        global _STORAGE_BF16
        _STORAGE_BF16 = self.prev


def sparse_conv(features, weight, bias, pairs, n_out):
    """spconv 'Native' algorithm: per offset gather -> mm -> scatter-add.  Differentiable (torch)."""
    kvol = weight.shape[0] * weight.shape[1] * weight.shape[2]
    w = weight.reshape(kvol, weight.shape[3], weight.shape[4])
    out = features.new_zeros((n_out, w.shape[2]))
    for k, (i_in, i_out) in enumerate(pairs):
        if len(i_in) == 0:
            continue
        i_in = torch.as_tensor(i_in, dtype=torch.long, device=features.device)
        i_out = torch.as_tensor(i_out, dtype=torch.long, device=features.device)
        out = out.index_add(0, i_out, features.index_select(0, i_in) @ w[k])
    if bias is not None:
        out = out + bias
    return out


def densify(features, coors, shape, batch_size):
    """SparseConvTensor.dense(): [B,C,D,H,W]."""
    c = torch.as_tensor(np.asarray(coors), dtype=torch.long, device=features.device)
    out = features.new_zeros((batch_size, shape[0], shape[1], shape[2], features.shape[1]))
    out[c[:, 0], c[:, 1], c[:, 2], c[:, 3]] = features
    return out.permute(0, 4, 1, 2, 3).contiguous()


# ----------------------------------------------------------------------------------------------
# Module-level restatement (same parameter names / weight layout as the spconv-based reference)
# ----------------------------------------------------------------------------------------------
class RefSparseTensor:
    def __init__(self, features, indices, spatial_shape, batch_size):
        self.features = features
        self.indices = np.asarray(indices).astype(np.int32)
        self.spatial_shape = tuple(int(v) for v in spatial_shape)
        self.batch_size = int(batch_size)
        self.indice_dict = {}

    def dense(self):
        return densify(self.features, self.indices, self.spatial_shape, self.batch_size)


class RefSparseConv(nn.Module):
    def __init__(self, cin, cout, ksize, stride=1, padding=0, dilation=1, bias=True, indice_key=None, subm=False):
        super().__init__()
        self.ksize, self.stride, self.padding, self.dilation = map(_triple, (ksize, stride, padding, dilation))
        self.subm = subm
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(*self.ksize, cin, cout))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        if bias:
            fan_in = cin * int(np.prod(self.ksize))
            self.bias = nn.Parameter(torch.empty(cout).uniform_(-1 / fan_in ** 0.5, 1 / fan_in ** 0.5))
        else:
            self.register_parameter("bias", None)

    def forward(self, x):
        if self.subm:
            key = self.indice_key
            if key is not None and key in x.indice_dict:
                pairs = x.indice_dict[key]
            else:
                pairs = rulebook_subm(x.indices, x.spatial_shape, self.ksize, self.dilation)
                if key is not None:
                    x.indice_dict[key] = pairs
            out = RefSparseTensor(_store(sparse_conv(x.features, self.weight, self.bias, pairs, x.indices.shape[0])),
                                  x.indices, x.spatial_shape, x.batch_size)
        else:
            oc, oshape, pairs = rulebook_conv(x.indices, x.spatial_shape, self.ksize, self.stride, self.padding,
                                              self.dilation)
            out = RefSparseTensor(_store(sparse_conv(x.features, self.weight, self.bias, pairs, oc.shape[0])),
                                  oc, oshape, x.batch_size)
        out.indice_dict = x.indice_dict
        return out


def RefSubMConv3d(cin, cout, ksize, stride=1, padding=0, bias=True, indice_key=None):
    return RefSparseConv(cin, cout, ksize, 1, padding, 1, bias, indice_key, subm=True)


def RefSparseConv3d(cin, cout, ksize, stride=1, padding=0, bias=True):
    return RefSparseConv(cin, cout, ksize, stride, padding, 1, bias, None, subm=False)


class RefSparseSequential(nn.Sequential):
    """spconv.SparseSequential: sparse modules take the tensor, plain modules its features."""

    def forward(self, x):
        mods = list(self)
        for i, m in enumerate(mods):
            if isinstance(m, (RefSparseConv, RefBasicBlock, RefSparseSequential)):
                x = m(x)
            else:
                x.features = m(x.features)
                # the product fuses BN + ReLU into one kernel with one bf16 store: round after the ReLU only
                if not (isinstance(m, nn.BatchNorm1d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)):
                    x.features = _store(x.features)
        return x


def _bn(c):
    return nn.BatchNorm1d(c, eps=1e-3, momentum=0.01)  # scn.py:100-101


class RefBasicBlock(nn.Module):
    """scn.py:42-85"""

    def __init__(self, c, key):
        super().__init__()
        self.conv1 = RefSubMConv3d(c, c, 3, padding=1, bias=True, indice_key=key)
        self.bn1 = _bn(c)
        self.relu = nn.ReLU()
        self.conv2 = RefSubMConv3d(c, c, 3, padding=1, bias=True, indice_key=key)
        self.bn2 = _bn(c)

    def forward(self, x):
        out = self.conv1(x)
        out.features = _store(self.relu(self.bn1(out.features)))
        out = self.conv2(out)
        out.features = self.bn2(out.features)
        out.features = _store(self.relu(out.features + x.features))
        return out


class RefSpMiddleResNetFHD(nn.Module):
    """scn.py:88-185"""

    def __init__(self, num_input_features=5):
        super().__init__()
        S = RefSparseSequential
        self.conv_input = S(RefSubMConv3d(num_input_features, 16, 3, bias=False, indice_key="res0"), _bn(16), nn.ReLU())
        self.conv1 = S(RefBasicBlock(16, "res0"), RefBasicBlock(16, "res0"))
        self.conv2 = S(RefSparseConv3d(16, 32, 3, 2, padding=1, bias=False), _bn(32), nn.ReLU(),
                       RefBasicBlock(32, "res1"), RefBasicBlock(32, "res1"))
        self.conv3 = S(RefSparseConv3d(32, 64, 3, 2, padding=1, bias=False), _bn(64), nn.ReLU(),
                       RefBasicBlock(64, "res2"), RefBasicBlock(64, "res2"))
        self.conv4 = S(RefSparseConv3d(64, 128, 3, 2, padding=[0, 1, 1], bias=False), _bn(128), nn.ReLU(),
                       RefBasicBlock(128, "res3"), RefBasicBlock(128, "res3"))
        self.extra_conv = S(RefSparseConv3d(128, 128, (3, 1, 1), (2, 1, 1), bias=False), _bn(128), nn.ReLU())

    def forward(self, voxel_features, coors, batch_size, input_shape):
        sparse_shape = np.array(input_shape[::-1]) + [1, 0, 0]  # scn.py:159
        x = RefSparseTensor(_store(voxel_features), coors, sparse_shape, batch_size)
        x = self.conv_input(x)
        c1 = self.conv1(x)
        c2 = self.conv2(c1)
        c3 = self.conv3(c2)
        c4 = self.conv4(c3)
        ret = self.extra_conv(c4).dense()
        n, c, d, h, w = ret.shape
        return ret.view(n, c * d, h, w), {"conv1": c1, "conv2": c2, "conv3": c3, "conv4": c4}


class RefSpMiddleFHD(nn.Module):
    """scn.py:187-289 (SECOND middle extractor; plain, non-residual)."""

    def __init__(self, num_input_features=5):
        super().__init__()
        L = []

        def subm(ci, co, key):
            L.extend([RefSubMConv3d(ci, co, 3, bias=False, indice_key=key), _bn(co), nn.ReLU()])

        def down(ci, co, pad):
            L.extend([RefSparseConv3d(ci, co, 3, 2, padding=pad, bias=False), _bn(co), nn.ReLU()])

        subm(num_input_features, 16, "subm0"); subm(16, 16, "subm0")
        down(16, 32, 1); subm(32, 32, "subm1"); subm(32, 32, "subm1")
        down(32, 64, 1); subm(64, 64, "subm2"); subm(64, 64, "subm2"); subm(64, 64, "subm2")
        down(64, 64, [0, 1, 1]); subm(64, 64, "subm3"); subm(64, 64, "subm3"); subm(64, 64, "subm3")
        self.middle_conv = RefSparseSequential(*L)
        self.extra_conv = RefSparseSequential(RefSparseConv3d(64, 64, (3, 1, 1), (2, 1, 1), bias=False), _bn(64),
                                              nn.ReLU())

    def forward(self, voxel_features, coors, batch_size, input_shape):
        sparse_shape = np.array(input_shape[::-1]) + [1, 0, 0]
        x = RefSparseTensor(_store(voxel_features), coors, sparse_shape, batch_size)
        c4 = self.middle_conv(x)
        ret = self.extra_conv(c4).dense()
        n, c, d, h, w = ret.shape
        return ret.view(n, c * d, h, w), c4


# ----------------------------------------------------------------------------------------------
# Independent dense formulation used to cross-check the above (known-answer generator)
# ----------------------------------------------------------------------------------------------
def dense_conv_reference(features, coors, shape, batch_size, weight, bias, ksize, stride, padding, subm):
    """Computes the same layer with F.conv3d on the densified input.  Returns
    (out_coors sorted canonical or input coors for subm, out_features)."""
    ksize, stride, padding = map(_triple, (ksize, stride, padding))
    x = densify(features, coors, shape, batch_size)
    occ = densify(torch.ones(features.shape[0], 1, dtype=features.dtype), coors, shape, batch_size)
    w = weight.permute(4, 3, 0, 1, 2)  # [Cout,Cin,kD,kH,kW]
    if subm:
        pad = tuple(k // 2 for k in ksize)
        y = F.conv3d(x, w, bias, stride=1, padding=pad)
        c = torch.as_tensor(np.asarray(coors), dtype=torch.long)
        return np.asarray(coors), y[c[:, 0], :, c[:, 1], c[:, 2], c[:, 3]]
    y = F.conv3d(x, w, bias, stride=stride, padding=padding)
    reach = F.conv3d(occ, torch.ones(1, 1, *ksize, dtype=features.dtype), None, stride=stride, padding=padding) > 0.5
    idx = reach[:, 0].nonzero()  # sorted lexicographically (b,z,y,x) == canonical order
    feats = y[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]]
    return idx.numpy().astype(np.int32), feats
