"""spconv-shaped module API on top of the HIP kernels.

The reference backbones are written against spconv v1.x (`/root/reference/det3d/models/backbones/
scn.py:2,8,18,104-152,162,173`); this module offers the same names, constructor arguments, weight
layout `[kD,kH,kW,Cin,Cout]` and `.features`-assignable tensor so that those call sites read the
same, while the arithmetic is ours:

  SparseConvTensor / .dense()        -> hip_ops.densify           (scn.py:162,173)
  SubMConv3d / SparseConv3d          -> rank/select rulebooks + MFMA implicit GEMM
  SparseSequential / SparseModule    -> container semantics of spconv.SparseSequential
  FeatureBatchNorm1d                 -> nn.BatchNorm1d drop-in (same state_dict) whose CUDA path
                                        is the fused stats/apply(+ReLU,+residual) kernels and which
                                        all-reduces its statistics when torch.distributed is up
                                        (apex SyncBN in the reference, apis/train.py:360-362)
There is no CPU path: every op raises on CPU tensors.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import collective as _collective
import torch.distributed as dist
from torch import nn

from . import hip_ops as H


def _triple(v):
    if isinstance(v, (list, tuple, np.ndarray)):
        assert len(v) == 3
        return tuple(int(x) for x in v)
    return (int(v),) * 3


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features = features
        self.indices = indices if indices.dtype == torch.int32 else indices.int()
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {}
        self.grid = grid

    @property
    def spatial_size(self):
        return int(np.prod(self.spatial_shape))

    def find_indice_pair(self, key):
        return None if key is None else self.indice_dict.get(key)

    def dense(self, channels_first=True):
        feats = self.features.float() if self.features.dtype == torch.bfloat16 else self.features   # bf16-storage stack
        out = _Densify.apply(feats, self.indices, self.batch_size, tuple(self.spatial_shape))
        return out if channels_first else out.permute(0, 2, 3, 4, 1).contiguous()

    def dense_bev(self, nhwc_bf16=False):
        """dense() folded to the BEV map [N, C*D, H, W] (scn.py:173-176); nhwc_bf16: channels_last bf16 output."""
        if nhwc_bf16 and self.features.is_cuda:
            return _DensifyBev.apply(self.features, self.indices, self.batch_size, tuple(self.spatial_shape))
        ret = self.dense()
        n, c, d, h, w = ret.shape
        return ret.view(n, c * d, h, w)


class _DensifyBev(torch.autograd.Function):
    """dense().view(N, C*D, H, W) emitted as NHWC bf16 for the bf16 BEV neck (no fp32 volume, no layout copy)."""

    @staticmethod
    def forward(ctx, feat, coors, batch, shape):
        ctx.save_for_backward(coors)
        ctx.meta = (batch, shape, feat.shape[1], feat.dtype)
        return H.densify_bev_bf16(feat, coors, batch, shape)

    @staticmethod
    def backward(ctx, dout):
        (coors,) = ctx.saved_tensors
        batch, shape, c, dt = ctx.meta
        return H.densify_bev_bf16_bwd(dout, coors, batch, shape, c, dt), None, None, None


class _Densify(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, coors, batch, shape):
        ctx.save_for_backward(coors)
        ctx.meta = (batch, shape, feat.shape[1])
        return H.densify(feat, coors, batch, shape)

    @staticmethod
    def backward(ctx, dout):
        (coors,) = ctx.saved_tensors
        batch, shape, c = ctx.meta
        return H.densify_bwd(dout, coors, batch, shape, c), None, None, None


class _SparseConvFn(torch.autograd.Function):
    """out = sum_k gather(feat, nbr[k]) @ W[k] (+bias); spconv.ops.indice_conv + its backward."""

    @staticmethod
    def _w_s16(weight, rb, cin_feat):
        """[kz,ky,kx,cin,cout] -> [K, cin_feat, cout]; the 5-channel input layer is zero-padded to the 16 stored channels"""
        w = weight.reshape(rb.kvol, weight.shape[-2], weight.shape[-1])
        if w.shape[1] < cin_feat:
            w = torch.nn.functional.pad(w, (0, 0, 0, cin_feat - w.shape[1]))
        return w

    @staticmethod
    def _s16(x, weight, bias, rb, nbr, n_out, transpose, flip, tag, cin_feat=None, bn_stats=False, pair_dgrad=None):
        """bf16-storage conv / data gradient with the weight image cached per parameter version (dense2d.cached_pack)"""
        from .dense2d import cached_pack
        cin_feat = x.shape[1] if cin_feat is None else cin_feat
        key = ("s16", bool(transpose), bool(flip), int(cin_feat))
        if pair_dgrad is not None and not transpose:
            # training forward: the data-gradient operand of this step is packed in the same launch (pair_dgrad = its (flip, n_out))
            from .dense2d import cached_pack_has, cached_pack_put, register_repack
            dkey = ("s16", True, bool(pair_dgrad[0]), int(cin_feat))
            if not cached_pack_has(weight, key) and not cached_pack_has(weight, dkey):
                pf, pd, launch, wsrc = H.spconv_s16_pack_pair(_SparseConvFn._w_s16(weight, rb, cin_feat), n_out, pair_dgrad[1], pair_dgrad[0],
                                                              with_launch=True)
                cached_pack_put(weight, key, pf)
                cached_pack_put(weight, dkey, pd)
                # the tile plan (and with it the image layout) does not depend on the row counts (csrc/spconv_s16.hip s16_plan): the same
                # launch refreshes both images in place after the optimizer step (dense2d.refresh_pack_cache) instead of being re-issued,
                # with its host work, in the middle of the next forward.  (The 5-channel input layer packs a padded copy: not registered.)
                register_repack(weight, [key, dkey], wsrc, launch)
        packed, kvol, cin, cout = cached_pack(
            weight, key, lambda: H.spconv_s16_pack(_SparseConvFn._w_s16(weight, rb, cin_feat), n_out, transpose, flip))
        b = None if bias is None else bias.detach().float().contiguous()
        if nbr is rb.nbr_out and H.spconv_s16_sorted_ok(rb, kvol, cin, cout, n_out):
            # r06: submanifold 64 -> 64 / 128 -> 128 layers (forward, and the data gradient = the same map with mirrored weights) run over the
            # rulebook's rows grouped by neighbour mask: offsets absent from a workgroup / tile are not multiplied (csrc/rulebook_sort.hip)
            return H.spconv_s16_run_sorted(x.contiguous(), packed, kvol, cin, cout, b, rb, tag, bn_stats=bn_stats)
        return H.spconv_s16_run(x.contiguous(), packed, kvol, cin, cout, b, nbr, n_out, rb.pair_count, tag, bn_stats=bn_stats)

    @staticmethod
    def forward(ctx, feat, weight, bias, rb, bn_stats=False):
        w = weight.reshape(rb.kvol, weight.shape[-2], weight.shape[-1])
        ctx.rb = rb
        ctx.has_bias = bias is not None
        ctx.bias_p = bias
        ctx.save_for_backward(feat, weight)
        ctx.s16 = feat.dtype == torch.bfloat16
        if ctx.s16:   # bf16 feature storage: gather -> LDS -> MFMA, bf16 out
            # the backward of this step will want the data-gradient operand (flip, rows) when the input needs a gradient
            pair = (rb.subm, rb.n_in) if (ctx.needs_input_grad[0] and weight.shape[-2] == feat.shape[1]) else None
            if bn_stats:   # second output: the statistics rows of the batch norm that follows (None on the neighbourhood-resident route)
                out, partial = _SparseConvFn._s16(feat, weight, bias, rb, rb.nbr_out, rb.n_out, False, False, "fwd", bn_stats=True, pair_dgrad=pair)
                if partial is not None:
                    ctx.mark_non_differentiable(partial)
                    ctx.set_materialize_grads(False)
                    return out, partial
                return out, torch.empty(0, device=out.device)
            return _SparseConvFn._s16(feat, weight, bias, rb, rb.nbr_out, rb.n_out, False, False, "fwd", pair_dgrad=pair)
        return H.spconv_gather_gemm(feat, w, bias, rb.nbr_out, rb.n_out, rb.pair_count, "fwd")

    @staticmethod
    def backward(ctx, dout, *_unused):
        feat, weight = ctx.saved_tensors
        rb = ctx.rb
        if ctx.s16:
            dout = dout.to(torch.bfloat16).contiguous()
            dfeat = dw = db = None
            if ctx.needs_input_grad[0]:
                nbr = rb.nbr_out if rb.subm else rb.nbr_in
                dfeat = _SparseConvFn._s16(dout, weight, None, rb, nbr, rb.n_in, True, rb.subm, "dgrad", cin_feat=feat.shape[1])
            if ctx.needs_input_grad[1]:   # off the chain to the previous layer: second stream when enabled (side.py)
                from . import side

                want_db = bool(ctx.has_bias and ctx.needs_input_grad[2])
                db_aside = want_db and "aux" in side.MODE

                def wgrad():   # (+ the bias gradient: a pass over dout nobody on the chain waits for)
                    dwf = H.spconv_s16_wgrad(feat, dout, rb.nbr_out, rb.kvol, rb.pair_count)
                    return dwf[:, : weight.shape[-2]].reshape(weight.shape).to(weight.dtype), (H.col_sums_bf16(dout) if db_aside else None)
                dw, db = side.run(weight, wgrad, feat, dout, rb.nbr_out, kind="sparse", bias=ctx.bias_p if db_aside else None, pair=True)
                if want_db and not db_aside:
                    db = H.col_sums_bf16(dout)
            elif ctx.has_bias and ctx.needs_input_grad[2]:
                db = H.col_sums_bf16(dout)
            return dfeat, dw, db, None, None
        dout = dout.contiguous()
        w = weight.reshape(rb.kvol, weight.shape[-2], weight.shape[-1])
        dfeat = dw = db = None
        if ctx.needs_input_grad[0]:
            if rb.subm:  # transposed map of a centred stencil = the same map with offsets mirrored
                dfeat = H.spconv_gather_gemm(dout, w, None, rb.nbr_out, rb.n_in, rb.pair_count, "dgrad", transpose=True,
                                             flip=True)
            else:
                dfeat = H.spconv_gather_gemm(dout, w, None, rb.nbr_in, rb.n_in, rb.pair_count, "dgrad", transpose=True)
        if ctx.needs_input_grad[1]:
            dw = H.spconv_wgrad(feat, dout, rb.nbr_out, rb.kvol, rb.pair_count).view_as(weight)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dout.sum(0)
        return dfeat, dw, db, None, None


class SparseModule(nn.Module):
    """marker base class: members of a SparseSequential that take the SparseConvTensor itself"""


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, output_padding=0, transposed=False, inverse=False, indice_key=None,
                 fused_bn=False, use_hash=False):
        super().__init__()
        assert ndim == 3 and groups == 1 and not transposed and not inverse, "only 3-D forward sparse convs on this path"
        self.ndim = ndim
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _triple(kernel_size), _triple(stride)
        self.padding, self.dilation = _triple(padding), _triple(dilation)
        self.subm = subm
        self.emit_bn_stats = False   # set by the owner when a FeatureBatchNorm1d consumes the output (SparseSequential / residual blocks)
        self.indice_key = indice_key
        self.plan_key = None   # set by a backbone: name under which a pre-planned strided rulebook is stored
        self.weight = nn.Parameter(torch.empty(*self.kernel_size, in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        # spconv v1.x: kaiming_uniform(a=sqrt(5)) on the [k,k,k,Cin,Cout] tensor, bias U(+-1/sqrt(fan_in))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        if self.bias is not None:
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1.0 / fan_in ** 0.5
            nn.init.uniform_(self.bias, -bound, bound)

    def rulebook(self, x: SparseConvTensor):
        if self.subm:
            rb = x.find_indice_pair(self.indice_key)
            if rb is None:
                rb = H.build_subm_rulebook(x.indices, x.batch_size, x.spatial_shape, self.kernel_size, self.dilation)
                if self.indice_key is not None:
                    x.indice_dict[self.indice_key] = rb
            return rb
        key = ("conv", self.plan_key if self.plan_key is not None else id(self))
        rb = x.indice_dict.get(key)
        if rb is None:
            rb = H.build_conv_rulebook(x.indices, x.batch_size, x.spatial_shape, self.kernel_size, self.stride,
                                       self.padding, self.dilation)
        return rb

    def forward(self, x: SparseConvTensor):
        assert isinstance(x, SparseConvTensor)
        rb = self.rulebook(x)
        feats = x.features
        if H.SPARSE_COMPUTE_DTYPE == "s16" and feats.is_cuda and feats.dtype != torch.bfloat16 \
                and self.out_channels in (16, 32, 64, 128) and (feats.shape[1] <= 16 or feats.shape[1] in (32, 64, 128)):
            # entry of the bf16-storage stack: the 5 point features are zero-padded to 16 stored channels
            if feats.shape[1] < 16:
                feats = torch.nn.functional.pad(feats, (0, 16 - feats.shape[1]))
            feats = feats.to(torch.bfloat16)
        if self.emit_bn_stats and self.training and torch.is_grad_enabled() and feats.dtype == torch.bfloat16 and rb.n_out > 0:
            feats, partial = _SparseConvFn.apply(feats, self.weight, self.bias, rb, True)
            if partial.numel():
                feats._s2d_bn_partial = partial   # read (forward pass only) by the FeatureBatchNorm1d that follows
        else:
            feats = _SparseConvFn.apply(feats, self.weight, self.bias, rb)
        if self.subm:
            out = SparseConvTensor(feats, x.indices, x.spatial_shape, x.batch_size)
        else:
            out = SparseConvTensor(feats, rb.out_coors, rb.out_shape, x.batch_size)
        out.indice_dict = x.indice_dict
        out.grid = x.grid
        return out


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, use_hash=False):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, subm=True,
                         indice_key=indice_key)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, use_hash=False):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, subm=False,
                         indice_key=indice_key)


# --------------------------------------------------------------------------------------------------
# BatchNorm1d on features
# --------------------------------------------------------------------------------------------------
_dist_on = _collective.sync_on


class _BNTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, relu, eps, sync, module):
        n, c = x.shape
        track = module is not None and module.track_running_stats
        mom = module.momentum if track else 0.0
        rm, rv = (module.running_mean, module.running_var) if track else (None, None)
        nbt = module.num_batches_tracked if track else None   # bumped inside the finalize kernel
        count = None
        if sync:   # statistics over the rows of ALL ranks: exchange [sum, sumsq, count] between the two kernels
            stats = H.bn1d_stats(x)
            count = torch.full((1,), float(n), device=x.device, dtype=x.dtype)
            packed = torch.cat([stats, count])
            _collective.allreduce_sum_(packed)
            stats, count = packed[:-1].contiguous(), packed[-1:].contiguous()
            fin = H.bn1d_finalize_fwd(stats, count, gamma, beta, eps, mom, rm, rv, nbt)
        else:
            fin = H.bn1d_stats_finalize(x, gamma, beta, eps, mom, rm, rv, nbt)
        mean, invstd, scale, shift = fin[0], fin[1], fin[2], fin[3]
        y = H.bn1d_apply(x, scale, shift, residual, relu)
        ctx.save_for_backward(x, y if relu else None, gamma, mean, invstd, count)
        ctx.relu, ctx.sync, ctx.has_res = relu, sync, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mean, invstd, count = ctx.saved_tensors
        if ctx.sync:
            g, sums = H.bn1d_bwd_reduce(dy.contiguous(), y, x, ctx.relu)
            # parameter grads use the LOCAL sums (DDP averages them over ranks afterwards, exactly
            # like torch.nn.SyncBatchNorm); the input grad needs the GLOBAL sums.
            sums_all = sums.clone()
            _collective.allreduce_sum_(sums_all)
            fin = H.bn1d_finalize_bwd(sums, sums_all, count, gamma, mean, invstd)
        else:
            g, fin = H.bn1d_bwd_reduce_finalize(dy, y, x, ctx.relu, gamma, mean, invstd)
        dx = H.bn1d_bwd_apply(g, x, fin[2], fin[3], fin[4]) if ctx.needs_input_grad[0] else None
        return dx, fin[0], fin[1], (g if ctx.has_res else None), None, None, None, None


class _BNEvalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, residual, relu, eps):
        invstd = torch.rsqrt(rvar + eps)
        scale = gamma * invstd
        shift = beta - rmean * scale
        y = H.bn1d_apply(x, scale, shift, residual, relu)
        ctx.save_for_backward(x, y if relu else None, scale, rmean, invstd)
        ctx.relu, ctx.has_res = relu, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, scale, rmean, invstd = ctx.saved_tensors
        c = x.shape[1]
        g, sums = H.bn1d_bwd_reduce(dy.contiguous(), y, x, ctx.relu)
        zero = torch.zeros_like(scale)
        dx = H.bn1d_bwd_apply(g, x, scale, zero, zero) if ctx.needs_input_grad[0] else None
        dbeta = sums[:c]
        dgamma = invstd * (sums[c:] - rmean * sums[:c])
        return dx, dgamma, dbeta, None, None, (g if ctx.has_res else None), None, None


class _EmptySyncFn(torch.autograd.Function):
    """Statistics exchange of a batch norm whose local feature matrix is empty: contributes zeros to the forward
    [sum, sumsq, count] and to the backward [sum g, sum g*x] all-reduces so that the collectives of all ranks match."""

    @staticmethod
    def forward(ctx, x, n_fwd, n_bwd):
        ctx.n_bwd = n_bwd
        ctx.dt = torch.float32 if x.dtype == torch.bfloat16 else x.dtype   # the statistics dtype of the other ranks
        _collective.allreduce_sum_(torch.zeros(n_fwd, dtype=ctx.dt, device=x.device))
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        _collective.allreduce_sum_(torch.zeros(ctx.n_bwd, dtype=ctx.dt, device=dy.device))
        return dy, None, None


class FeatureBatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d with identical parameters/buffers; CUDA [N,C] inputs run the fused HIP path."""

    _REQUIRE_CUDA = True   # tests/cpu_backend.py clears it together with swapping the HIP launchers for the oracle

    def forward(self, x, residual=None, relu=False):
        if (self._REQUIRE_CUDA and not x.is_cuda) or x.dim() != 2:
            raise RuntimeError("FeatureBatchNorm1d: expected a CUDA [N,C] feature matrix (no CPU fallback)")
        if self.momentum is None:
            raise RuntimeError("FeatureBatchNorm1d: momentum=None (cumulative average) is not supported by the fused kernels")
        use_batch_stats = self.training or not self.track_running_stats
        if x.shape[0] == 0:
            if use_batch_stats and _dist_on():
                # a rank without active sites still takes part in the statistics exchange of the other ranks (forward:
                # zero sums and a zero count; the backward collective is matched by _EmptySyncFn)
                return _EmptySyncFn.apply(x, 2 * self.num_features + 1, 2 * self.num_features)
            return x
        if x.dtype == torch.bfloat16:   # bf16-storage stack: the row-major bf16 kernels (also used by the BEV neck)
            from .dense2d import _BNRowFn
            if self.num_features % 8:
                raise RuntimeError("FeatureBatchNorm1d: bf16 features need a channel count that is a multiple of 8")
            partial = getattr(x, "_s2d_bn_partial", None) if use_batch_stats else None
            if partial is not None and (partial.shape[2] != self.num_features or not x.is_contiguous()):
                partial = None
            return _BNRowFn.apply(x.contiguous(), self.weight, self.bias, None if residual is None else residual.contiguous(),
                                  relu, self.eps, _dist_on() and use_batch_stats, self, use_batch_stats, partial)
        if use_batch_stats:
            return _BNTrainFn.apply(x, self.weight, self.bias, residual, relu, self.eps, _dist_on() and self.training, self)
        return _BNEvalFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var, residual, relu, self.eps)


# --------------------------------------------------------------------------------------------------
class SparseSequential(SparseModule):
    """spconv.SparseSequential: sparse modules receive the tensor, plain modules its `.features`.
    BN -> ReLU runs are fused into one kernel launch."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)
        self._mark_bn_producers()

    def _mark_bn_producers(self):
        mods = list(self._modules.values())
        for a, b in zip(mods, mods[1:]):
            if isinstance(a, SparseConvolution) and isinstance(b, FeatureBatchNorm1d):
                a.emit_bn_stats = True

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError(f"index {idx} is out of range")
        if idx < 0:
            idx += len(self)
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        name = str(len(self._modules)) if name is None else name
        if name in self._modules:
            raise KeyError("name exists")
        self.add_module(name, module)
        self._mark_bn_producers()

    def forward(self, x):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, SparseModule):
                x = m(x)
            elif isinstance(x, SparseConvTensor):
                if x.indices.shape[0] != 0:
                    if isinstance(m, FeatureBatchNorm1d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
                        x.features = m(x.features, relu=True)
                        i += 1
                    else:
                        x.features = m(x.features)
            else:
                x = m(x)
            i += 1
        return x
