"""Data parallelism: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl" on
ROCm), frames sharded by rank, no data-path collective except
  * the bucketed gradient all-reduce (`GradBuckets`, default route "overlap": parameters are grouped
    in reverse registration order - the order the backward produces their gradients, head and neck
    first - and a bucket's all-reduce is launched asynchronously from the gradient hook of its last
    parameter, so it flies while the sparse-conv backward still runs); and
  * SyncBN statistics (FeatureBatchNorm1d / FastBatchNorm2d / FastBatchNorm3d all-reduce their own
    [2C+1] vectors; other BatchNorm2d/3d layers are converted to torch.nn.SyncBatchNorm).
Reference: tools/train.py:86-96 (init_process_group), det3d/torchie/apis/train.py:360-391
(apex convert_syncbn_model + DistributedDataParallel).  The redundant second gradient all-reduce
of the reference's DistOptimizerHook (apis/train.py:313-316) is intentionally dropped: the
gradients have already been averaged, so the result is identical.
"""
import os

import torch
import torch.distributed as dist
from torch import nn

from .dense2d import FastBatchNorm2d
from .dense3d import FastBatchNorm3d
from .spconv import FeatureBatchNorm1d


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("S2D_FORCE_DDP", "0") == "1"   # exercise the N>1 code path with one rank (tests)
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def convert_syncbn(module: nn.Module):
    """BatchNorm2d/3d -> SyncBatchNorm; FeatureBatchNorm1d, FastBatchNorm2d and FastBatchNorm3d are left alone (they
    synchronise themselves inside their fused kernels' autograd functions)."""
    out = module
    if isinstance(module, nn.modules.batchnorm._BatchNorm) and not isinstance(module, (FeatureBatchNorm1d, FastBatchNorm3d, FastBatchNorm2d, nn.SyncBatchNorm)):
        out = nn.SyncBatchNorm(module.num_features, module.eps, module.momentum, module.affine, module.track_running_stats)
        if module.affine:
            with torch.no_grad():
                out.weight = module.weight
                out.bias = module.bias
        out.running_mean = module.running_mean
        out.running_var = module.running_var
        out.num_batches_tracked = module.num_batches_tracked
        out.training = module.training
    for name, child in module.named_children():
        out.add_module(name, convert_syncbn(child))
    return out


def dp_mode():
    """S2D_DP_MODE selects the gradient route:
      "overlap" (default)  GradBuckets, all-reduce of a bucket launched from the backward hooks (overlaps the backward);
      "flat"               GradBuckets, every bucket launched after the backward (no overlap; A/B reference);
      "ddp"                torch DistributedDataParallel."""
    return os.environ.get("S2D_DP_MODE", "overlap")


_BUCKETERS = {}   # id(parameter) -> GradBuckets that owns it


class GradBuckets:
    """Bucketed gradient averaging without DDP's bookkeeping.  The bucket layout is static (every parameter that
    requires a gradient, whether or not this step produces one on this rank: a missing gradient is sent as zeros, so
    ranks can never disagree on message sizes).  Per step: `prepare()` -> backward -> `finish()`; afterwards every
    `p.grad` is a view into its bucket's flat buffer holding the average over the ranks."""

    def __init__(self, params, bucket_bytes=None, overlap=True):
        self.params = [p for p in params if p.requires_grad]
        self.overlap = overlap
        cap = int(float(os.environ.get("S2D_BUCKET_MB", 16)) * (1 << 20)) if bucket_bytes is None else int(bucket_bytes)
        self.buckets = []     # dicts: params, flat, views, pending, work
        cur, cur_bytes, cur_key = [], 0, None
        for p in reversed(self.params):   # reverse registration order = (approximately) gradient production order
            key = (p.dtype, p.device)
            nb = p.numel() * p.element_size()
            if cur and (key != cur_key or cur_bytes + nb > cap):
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p); cur_bytes += nb; cur_key = key
        if cur:
            self._close(cur)
        # r04: the buckets travel on THEIR OWN process group (communicator).  On the default group a bucket that `_launch_ready` holds
        # back on one rank (a parameter without a gradient there) would be issued AFTER SyncBN-backward all-reduces which another rank
        # issues BEFORE it - two ranks with different collective sequences on one communicator.  With a group of their own the bucket
        # sequence is 0, 1, 2, ... on every rank and the SyncBN sequence is the autograd order on every rank, each on its own
        # communicator (and, with NCCL/RCCL, its own stream).  Created collectively: every rank constructs its GradBuckets at the same point.
        self.group = dist.new_group() if (dist.is_available() and dist.is_initialized()) else None
        self._handles = []
        self._next, self.launch_log = 0, []
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                _BUCKETERS[id(p)] = self
                self._handles.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
        self._armed = False

    def _close(self, ps):
        flat = torch.zeros(sum(p.numel() for p in ps), dtype=ps[0].dtype, device=ps[0].device)
        views, off = [], 0
        for p in ps:
            views.append(flat[off:off + p.numel()].view_as(p)); off += p.numel()
        self.buckets.append(dict(index=len(self.buckets), params=ps, flat=flat, views=views, pending=len(ps), work=None, launched=False))

    def _make_hook(self, bi):
        def hook(_p):
            if not self._armed:
                return
            b = self.buckets[bi]
            b["pending"] -= 1
            if b["pending"] == 0 and self.overlap:
                self._launch_ready()
        return hook

    def _launch_ready(self):
        """Launch, IN BUCKET-INDEX ORDER, every bucket whose predecessors have all been launched and whose own hooks have all
        fired.  A bucket that is complete while an earlier one still waits for a gradient (a parameter unused on this rank this
        step, a gated branch, an empty frame) is held back; `finish()` then launches the remainder in index order.  Every rank
        therefore issues bucket 0, 1, 2, ... on the communicator whatever its local gradient arrival order was (ranks that
        disagree on which parameters got a gradient would otherwise pair all-reduces of different buckets - torch DDP reduces
        in index order for the same reason)."""
        while self._next < len(self.buckets) and self.buckets[self._next]["pending"] == 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    def _launch(self, b):
        # Gradients produced on the weight-gradient stream (side.py) are complete only in THAT stream's order.  The bucket's copies and
        # its all-reduce launch are issued from the side stream (which first waits for an event of the current stream: the gradients the
        # chain produced itself), so the chain never waits for the side stream's backlog; `finish()` waits for the collective as before.
        from . import side
        dev = b["flat"].device.index if b["flat"].is_cuda else None
        st = side.stream_after(dev)
        cur = torch.cuda.current_stream(dev) if st is not None else None
        if st is not None:
            torch.cuda.set_stream(st)
        try:
            src, dst = [], []
            for p, v in zip(b["params"], b["views"]):
                if p.grad is None:
                    v.zero_()
                elif p.grad.data_ptr() != v.data_ptr():
                    src.append(p.grad); dst.append(v)
            if src:
                torch._foreach_copy_(dst, src)
            b["work"] = dist.all_reduce(b["flat"], group=self.group, async_op=True)   # c10d: ordered after the copies, runs on its own stream
        finally:
            if st is not None:
                torch.cuda.set_stream(cur)
        b["launched"] = True
        self.launch_log.append(b["index"])

    def prepare(self):
        for b in self.buckets:
            b["pending"], b["work"], b["launched"] = len(b["params"]), None, False
            for p in b["params"]:
                p.grad = None
        self._next = 0
        self.launch_log = []      # bucket indices in launch order (tests assert it is 0, 1, 2, ... on every rank)
        self._armed = True

    def finish(self):
        self._armed = False
        world = dist.get_world_size()
        for b in self.buckets[self._next:]:   # in index order: buckets held back behind one whose hooks did not all fire
            self._launch(b)                  # (unused parameters), or every bucket on the "flat" route
        self._next = len(self.buckets)
        for b in self.buckets:
            b["work"].wait()     # NCCL: the current stream waits for the collective, the host does not block
            if world > 1:
                b["flat"].div_(world)
            for p, v in zip(b["params"], b["views"]):
                p.grad = v

    def remove(self):
        for h in self._handles:
            h.remove()
        for b in self.buckets:
            for p in b["params"]:
                _BUCKETERS.pop(id(p), None)


def bucketers_of(params):
    seen, out = set(), []
    for p in params:
        g = _BUCKETERS.get(id(p))
        if g is not None and id(g) not in seen:
            seen.add(id(g)); out.append(g)
    return out


def wrap_ddp(model: nn.Module, local_rank=None, bucket_cap_mb=40, find_unused_parameters=False):
    force = os.environ.get("S2D_FORCE_DDP", "0") == "1"
    if not (dist.is_initialized() and (dist.get_world_size() > 1 or force)):
        return model
    model = convert_syncbn(model)
    mode = dp_mode()
    if mode in ("overlap", "flat"):
        if next(model.parameters()).is_cuda:
            # SyncBN vectors: RCCL on the compute stream through the C library's own communicator (collective.py); the
            # gradient buckets go through torch.distributed (c10d's communicator and stream).  Both kinds of collective
            # are issued from the host in program / autograd order, which is the same on every rank.
            from . import collective
            collective.init_direct(torch.cuda.current_device() if local_rank is None else local_rank)
        with torch.no_grad():   # what the DDP constructor does: every rank starts from rank 0's parameters and buffers
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t, 0)
        model._s2d_grad_buckets = GradBuckets(list(model.parameters()), overlap=(mode == "overlap"))
        return model
    bucket_cap_mb = float(os.environ.get("S2D_DDP_BUCKET_MB", bucket_cap_mb))
    kwargs = dict(bucket_cap_mb=bucket_cap_mb, find_unused_parameters=find_unused_parameters,
                  gradient_as_bucket_view=True, static_graph=os.environ.get("S2D_DDP_STATIC_GRAPH", "0") == "1")
    if next(model.parameters()).is_cuda:
        dev = torch.cuda.current_device() if local_rank is None else local_rank
        return nn.parallel.DistributedDataParallel(model, device_ids=[dev], output_device=dev, **kwargs)
    return nn.parallel.DistributedDataParallel(model, **kwargs)
