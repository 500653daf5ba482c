"""The shape-static dense segment of a training step as two HIP graphs (forward, backward).

Why.  One S2D student step enqueues ~880 launches from Python; ~600 of them belong to the BEV neck + PCR head + CenterHead +
losses, whose tensor shapes depend on nothing but the batch size.  The reference pays the same host cost per layer
(/root/reference/det3d/torchie/trainer/trainer.py:775-811 calls the modules one by one and lets cuDNN / spconv queue their kernels); on
MI355X the kernels of that segment take ~12 ms, the Python that launches them 10-20 ms depending on the host, so the step's wall time was
the host's.  Here the segment is recorded once and replayed: two `hipGraphLaunch` calls per step (forward, backward), no Python
in between, the GIL free for the data-pipeline thread.

What is captured.  `fn(*tensors) -> tuple(tensors)`: neck, head and every loss term whose inputs are static-shaped.  The sparse
backbone (row counts change with every point cloud) stays eager on both sides: its BEV map is the segment's differentiable input, the
gradient of that map the backward graph's output.  Variable-length side inputs (the PCR head's reconstruction voxels) are handed over
in capacity-sized buffers padded with out-of-range rows, which the kernels skip (csrc/losses.hip: `(unsigned)c.x >= batch -> continue`).

Protocol.
  * the first `WARMUP` calls with a given signature run `fn` eagerly (they also fill the packed-weight caches, so that no pack launch
    ends up in the graph) and record which outputs receive a gradient;
  * the next call captures: static copies of the inputs, the forward under `torch.cuda.graph`, then
    `torch.autograd.grad(outputs that get gradients, differentiable inputs + parameters)` into a second graph, both in one memory pool;
  * afterwards a call = copy the inputs into the static buffers + replay; an autograd node (`_Replay`) hands the static outputs to
    the caller, and in the backward pass copies the incoming gradients, replays the backward graph, binds each parameter's static
    gradient buffer as its `.grad` (accumulating out of place when one is already there - a `.grad` that is the static buffer itself is
    copied out before the replay overwrites it -, running the parameter's post-accumulate hooks as AccumulateGrad would) and returns
    the static input gradients.
  * a replay is refused - the capture dropped, the eager path taken and a new warm-up started - when a parameter's version counter,
    storage or `requires_grad` changed since the capture (load_state_dict, manual surgery: the packed images the graph reads would be
    stale), when the gradient pattern differs, or under the per-launch profiling pass of bench.py.
Outputs are views of static buffers: they are valid until the segment's next call (a training step consumes them at once).
"""
import os
import threading
import time

import torch

# Held while a segment is being captured.  Other threads that talk to the HIP runtime (data.PrefetchLoader builds the next example on its
# own stream: allocations, host reads) take it around their work: stream capture in the default (global) mode turns a synchronising or
# allocating runtime call made by ANY thread during the capture into an error.
CAPTURE_LOCK = threading.RLock()

WARMUP = int(os.environ.get("S2D_GRAPH_WARMUP", "2"))
MAX_CAPTURES = int(os.environ.get("S2D_GRAPH_MAX_CAPTURES", "6"))   # per segment: train / eval x grad modes x a couple of input signatures
stats = {"eager": 0, "capture": 0, "replay": 0, "dropped": 0, "parked": 0, "launch_host_ms": 0.0}   # launch_host_ms: host time spent inside hipGraphLaunch calls


def enabled():
    return os.environ.get("S2D_DENSE_GRAPH", "1") != "0"


def _capture_mode():
    """hipStreamBeginCapture mode.  "global" (torch's default): an unsafe runtime call on ANY thread invalidates the capture - right while this
    process is the only one talking to the device.  With a process group up, c10d's watchdog thread polls its work events (hipEventQuery)
    whenever it likes: under "global" that aborted the process in the middle of a capture (r05); "thread_local" confines the check to the
    capturing thread.  S2D_GRAPH_CAPTURE_MODE overrides."""
    mode = os.environ.get("S2D_GRAPH_CAPTURE_MODE")
    if mode:
        return mode
    import torch.distributed as dist
    return "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"


def _sig(t):
    return (tuple(t.shape), t.dtype, tuple(t.stride()), bool(t.requires_grad))


def _static_like(t):
    s = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device)
    return s


def _copy_all(dst, src):
    """dst[i].copy_(src[i]) for all i in as few launches as possible: pairs of equal dtype and strides go through one multi-tensor kernel"""
    groups = {}
    for d, t in zip(dst, src):
        if d.dtype == t.dtype and d.shape == t.shape and d.stride() == t.stride() and d.device == t.device:
            g = groups.setdefault(d.dtype, ([], []))
            g[0].append(d)
            g[1].append(t.detach())
        else:
            d.copy_(t)
    for fast_d, fast_s in groups.values():
        if len(fast_d) == 1:
            fast_d[0].copy_(fast_s[0])
        else:
            torch._foreach_copy_(fast_d, fast_s)


class _Capture:
    __slots__ = ("fwd", "bwd", "bwd2", "keep", "s_in", "s_out", "g_idx", "s_gout", "s_gin", "s_gparams", "diff_idx", "params", "pstate", "pool", "stream",
                 "bound")


class _Replay(torch.autograd.Function):
    """autograd node of one replayed segment call: forward = replay of the forward graph (done by the caller, which hands the static
    outputs in), backward = replay of the backward graph"""

    @staticmethod
    def forward(ctx, seg, cap, anchor, *diff_inputs):
        ctx.seg, ctx.cap = seg, cap
        t0 = time.perf_counter()
        cap.fwd.replay()
        stats["launch_host_ms"] += (time.perf_counter() - t0) * 1e3
        outs = tuple(o.detach() for o in cap.s_out)
        ctx.mark_non_differentiable(*[o for i, o in enumerate(outs) if i not in cap.g_idx])
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        cap, seg = ctx.cap, ctx.seg
        dst, src = [], []
        for k, i in enumerate(cap.g_idx):
            g = gouts[i]
            if g is None:
                cap.s_gout[k].zero_()
            else:
                dst.append(cap.s_gout[k])
                src.append(g)
        if dst:
            _copy_all(dst, src)
        for i, g in enumerate(gouts):
            if g is not None and i not in cap.g_idx:
                # an output that received no gradient during the warm-up now has one: this capture cannot deliver it
                raise RuntimeError(f"GraphedSegment '{seg.name}': output {i} received a gradient that the captured backward does not cover; "
                                   "call .reset() after changing which losses are used")
        # Gradient accumulation (a second backward before the gradients were cleared, zero_grad(set_to_none=False)): a `.grad` that IS one of
        # this capture's static buffers - bound by the previous backward - is about to be overwritten by the replay: its values move to a
        # private copy first (enqueued ahead of both replays), and the sums below wait for the weight-gradient graph's stream.  ADVICE r05.
        accumulate = False
        for p, g in zip(cap.params, cap.s_gparams):
            if g is not None and p.grad is not None:
                accumulate = True
                if p.grad.data_ptr() == g.data_ptr():
                    p.grad = p.grad.clone()
        t0 = time.perf_counter()
        cap.bwd.replay()
        stats["launch_host_ms"] += (time.perf_counter() - t0) * 1e3
        dev = cap.s_in[0].device.index
        joined = cap.bwd2 is None
        if cap.bwd2 is not None:   # the weight gradients' graph: on the side stream, beside whatever the caller's backward does next
            from . import side
            side.replay_on_side(cap.bwd2, dev)
            if accumulate:
                side.join(dev)
                joined = True
        cap.bound = True   # static buffers are (about to be) somebody's .grad: see GraphedSegment.__call__
        for p, g in zip(cap.params, cap.s_gparams):
            if g is None:
                continue
            if p.grad is None:
                p.grad = g
            else:
                p.grad = p.grad + g      # (never in place: p.grad may be another segment's static buffer)
            hooks = getattr(p, "_post_accumulate_grad_hooks", None)
            if hooks:
                if not joined:   # only the data-parallel buckets' hooks order themselves behind the side stream (dp.GradBuckets._launch);
                    from . import dp, side   # any other hook reads p.grad on this stream: it must see the finished gradient
                    if len(hooks) != 1 or id(p) not in dp._BUCKETERS:
                        side.join(dev)
                        joined = True
                for h in list(hooks.values()):
                    h(p)
        return (None, None, None) + tuple(cap.s_gin)


_SEGMENTS = None   # weak set of live segments (grads_consumed)


def grads_consumed():
    """Called by a training step AFTER its optimizer has read the gradients (train_step.backward_and_step): the `.grad` tensors that are static
    buffers of a capture need not survive the next forward replay - the step clears every `.grad` before its next backward anyway.  Without this
    declaration the next forward replay first moves each of them to a private copy (gradient accumulation must not read a buffer the
    forward graph reuses) - 164 clones per step for the S2D student (0.64 ms of copy kernels and ~1.6 ms of launch-thread time, r06 profile of the
    graphed mode: `__amd_rocclr_copyBuffer` 180 launches per step against 16 in r05).  Reading such a `.grad` after the NEXT forward is then
    undefined; code that accumulates over micro-batches simply does not call this."""
    if _SEGMENTS is None:
        return
    for seg in list(_SEGMENTS):
        for cap in seg._caps.values():
            cap.bound = False


class GraphedSegment:
    def __init__(self, fn, modules, name="dense"):
        """fn(*tensors) -> tuple of tensors; modules: the nn.Modules whose parameters fn reads (those with requires_grad get their
        gradient from the backward graph)"""
        self.fn = fn
        self.name = name
        self.modules = list(modules)
        self.params = [p for m in self.modules for p in m.parameters()]
        self._caps = {}       # signature -> _Capture
        self._seen = {}       # signature -> [calls so far, set of output indices that received a gradient]
        self.pool = None
        global _SEGMENTS
        if _SEGMENTS is None:
            import weakref
            _SEGMENTS = weakref.WeakSet()
        _SEGMENTS.add(self)

    def reset(self):
        self._caps.clear()
        self._seen.clear()

    # ------------------------------------------------------------------------------------------------------------------
    def _pstate(self):
        return [(p._version, p.data_ptr(), p.requires_grad) for p in self.params]

    def _usable(self, inputs):
        from . import hip_ops as H
        if not enabled() or H.PROFILE is not None or torch.cuda.is_current_stream_capturing():
            return False
        return all(torch.is_tensor(t) and t.is_cuda for t in inputs)

    def __call__(self, *inputs):
        if not self._usable(inputs):
            stats["eager"] += 1
            return self.fn(*inputs)
        grad_mode = torch.is_grad_enabled()
        from . import side
        key = (tuple(_sig(t) for t in inputs), grad_mode, torch.is_autocast_enabled(), torch.get_autocast_gpu_dtype(), frozenset(side.GRAPH_KINDS), frozenset(side.GRAPH_DEFER))
        cap = self._caps.get(key)
        if cap is not None:
            ps = cap.pstate
            for i, p in enumerate(self.params):
                v = ps[i]
                if p._version != v[0] or p.requires_grad != v[2] or p.data_ptr() != v[1]:
                    cap = None
                    break
            if cap is None:   # parameters were edited behind the graph's back: forget everything recorded for this signature
                del self._caps[key]
                self._seen.pop(key, None)
                stats["dropped"] += 1
        if cap is None:
            seen = self._seen.setdefault(key, [0, set()])
            if seen[0] < WARMUP:
                seen[0] += 1
                stats["eager"] += 1
                outs = self.fn(*inputs)
                if grad_mode:
                    for i, o in enumerate(outs):
                        if torch.is_tensor(o) and o.requires_grad:
                            o.register_hook(lambda g, i=i, s=seen[1]: s.add(i))
                return outs
            if self.__dict__.get("_broken"):
                stats["eager"] += 1
                return self.fn(*inputs)
            try:
                cap = self._capture(inputs, sorted(seen[1]), grad_mode)
            except Exception as e:   # something inside the segment cannot be recorded (a library call that allocates / synchronises): keep the
                # eager path for this segment from now on, and say so once
                import sys
                self._broken = repr(e)[:300]
                stats["capture_failed"] = stats.get("capture_failed", 0) + 1
                print(f"[s2d] GraphedSegment '{self.name}': capture failed, segment stays eager: {self._broken}", file=sys.stderr, flush=True)
                torch.cuda.synchronize()
                return self.fn(*inputs)
            self._caps[key] = cap
            while len(self._caps) > MAX_CAPTURES:   # signatures that can no longer occur (a padded side input grew its capacity) would only
                old_key = next(iter(self._caps))    # hold graph-pool memory: the oldest capture goes
                del self._caps[old_key]
                self._seen.pop(old_key, None)
                stats["dropped"] += 1
            stats["capture"] += 1
        stats["replay"] += 1
        if getattr(cap, "bound", False):
            # The previous backward bound static gradient buffers as `.grad`.  A training step clears them before the next forward; when it
            # did not (gradient accumulation over micro-batches, zero_grad(set_to_none=False)) the values must leave the graph's memory pool
            # BEFORE this forward replays: forward and backward graph share the pool, a block that holds a gradient in the backward graph
            # may be a temporary of the forward graph (r06: a second backward accumulated onto garbage - tests/test_graph_gpu.py).
            for p, g in zip(cap.params, cap.s_gparams):
                if g is not None and p.grad is not None and p.grad.data_ptr() == g.data_ptr():
                    p.grad = p.grad.clone()
                    stats["parked"] += 1   # (a training step that called grads_consumed() never gets here)
            cap.bound = False
        with torch.no_grad():   # ONE multi-tensor copy kernel for all inputs (a hipMemcpyAsync per tensor costs the host ~60 us each under load)
            pairs = [(s, t) for s, t in zip(cap.s_in, inputs) if s.data_ptr() != t.data_ptr()]
            if pairs:
                _copy_all([s for s, _ in pairs], [t for _, t in pairs])
        if not grad_mode or not cap.g_idx:
            cap.fwd.replay()
            return tuple(o.detach() for o in cap.s_out)
        # (the anchor makes the node part of the autograd graph even when no input is differentiable: the parameters' gradients come out
        # of its backward either way)
        return _Replay.apply(self, cap, self._anchor(inputs[0].device), *[inputs[i] for i in cap.diff_idx])

    def _anchor(self, dev):
        a = self.__dict__.get("_anchor_t")
        if a is None or a.device != dev:
            a = self._anchor_t = torch.zeros((), device=dev, requires_grad=True)
        return a

    def static_input(self, inputs, index):
        """the static buffer input `index` will be copied into at the next replay of the signature `inputs` have (a producer that
        writes there directly saves the copy), or None while that signature has no capture"""
        from . import side
        key = (tuple(_sig(t) for t in inputs), torch.is_grad_enabled(), torch.is_autocast_enabled(), torch.get_autocast_gpu_dtype(), frozenset(side.GRAPH_KINDS), frozenset(side.GRAPH_DEFER))
        cap = self._caps.get(key)
        return None if cap is None else cap.s_in[index]

    # ------------------------------------------------------------------------------------------------------------------
    def _capture(self, inputs, g_idx, grad_mode):
        from . import dense2d
        cap = _Capture()
        dev = inputs[0].device
        cap.pstate = self._pstate()
        # The capture differentiates with respect to fresh leaf ALIASES of the parameters (same storage, same version counter), swapped into
        # the modules for the duration of the forward capture.  Reason: the autograd engine orders a gradient that flows to a parameter
        # against the stream its AccumulateGrad node was created on - normally the default stream, during an earlier eager step whose
        # autograd graph something still references - and an event wait that drags the default stream into a HIP stream capture crashes
        # hipStreamEndCapture.  The aliases' accumulator nodes are born inside the capture, on the capture stream.
        import contextlib
        from torch.nn.utils.stateless import _reparametrize_module
        cap.params, proxies, swaps = [], [], []
        for m in self.modules:
            repl = {}
            for name, p in m.named_parameters():
                if p.requires_grad:
                    q = p.detach().requires_grad_(True)
                    q._s2d_origin = p          # identity-keyed caches (dense2d.cached_pack) keep ONE entry per parameter
                    repl[name] = q
                    cap.params.append(p)
                    proxies.append(q)
            swaps.append((m, repl))
        cap.s_in = []
        cap.diff_idx = []
        for i, t in enumerate(inputs):
            if getattr(t, "_s2d_static", False) and not t.requires_grad:   # a persistent buffer of the caller's: replayed in place
                cap.s_in.append(t)
                continue
            s = _static_like(t)
            s.copy_(t.detach())
            if t.requires_grad and grad_mode:
                s.requires_grad_(True)
                cap.diff_idx.append(i)
            cap.s_in.append(s)
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        cap.pool = self.pool
        cap.stream = torch.cuda.Stream(dev)
        torch.cuda.synchronize(dev)
        cap.fwd = torch.cuda.CUDAGraph()
        CAPTURE_LOCK.acquire()
        dense2d.WS_PRIVATE = True      # reduction workspaces: plain allocations from the graph's pool while capturing
        dense2d.CAPTURE_PACKS = {}     # packed-weight images without an in-place refresh: built inside this capture (dense2d.cached_pack)
        # No cyclic garbage collection while a stream is capturing: a collection that runs in the middle of the capture (it can start at any
        # allocation, on the autograd thread as well) may finalise an old CUDAGraph / event / tensor of an earlier segment or test, and
        # destroying those is a runtime call the capture turns into an abort (seen in the full test suite: "Garbage-collecting" inside a
        # layer's backward, then SIGABRT).  Collect once up front, switch the collector off, restore afterwards.
        import gc
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            with contextlib.ExitStack() as stack:
                for m, repl in swaps:
                    stack.enter_context(_reparametrize_module(m, repl))
                with torch.cuda.graph(cap.fwd, pool=cap.pool, stream=cap.stream, capture_error_mode=_capture_mode()):
                    outs = self.fn(*cap.s_in)
            outs = tuple(outs)
            cap.s_out = outs
            # outputs whose gradient the backward graph covers: those that received one during the warm-up; when the warm-up calls never ran
            # a backward (forward-only sanity / timing calls in train mode) every output that requires one - a superset whose unused
            # entries replay with zero-filled gradients - instead of a capture that silently detaches its outputs
            if grad_mode and not g_idx:
                g_idx = [i for i, o in enumerate(outs) if torch.is_tensor(o) and o.requires_grad]
            cap.g_idx = [i for i in g_idx if torch.is_tensor(outs[i]) and outs[i].requires_grad] if grad_mode else []
            cap.s_gout, cap.s_gin, cap.s_gparams, cap.bwd, cap.bwd2, cap.keep = [], [], [], None, None, None
            if cap.g_idx:
                cap.s_gout = [torch.zeros_like(outs[i]) for i in cap.g_idx]
                wrt = [cap.s_in[i] for i in cap.diff_idx] + proxies
                cap.bwd = torch.cuda.CUDAGraph()
                from . import side
                side.take_deferred()
                with torch.cuda.graph(cap.bwd, pool=cap.pool, stream=cap.stream, capture_error_mode=_capture_mode()):
                    grads = torch.autograd.grad([outs[i] for i in cap.g_idx], wrt, cap.s_gout, allow_unused=True)
                    side.join_capture()      # weight-gradient groups forked onto a second capturing stream (side.GRAPH_KINDS) rejoin here
                side.capture_done()
                nd = len(cap.diff_idx)
                cap.s_gin = list(grads[:nd])
                cap.s_gparams = list(grads[nd:])
                # weight-gradient groups the layers queued instead of launching (side.GRAPH_DEFER): a second, linear graph.  Their operands
                # (layer inputs and output gradients: tensors of the chain's graph) stay referenced by `cap.keep` for the capture's lifetime -
                # the pool must never hand their blocks to a later capture while this graph can still be replayed.
                items = side.take_deferred()
                if items:
                    slot = {id(q): k for k, q in enumerate(proxies)}
                    cap.bwd2 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(cap.bwd2, pool=cap.pool, stream=cap.stream, capture_error_mode=_capture_mode()):
                        for weight, bias, fn, _inputs in items:
                            out = fn()
                            dw, db = out if isinstance(out, (tuple, list)) else (out, None)
                            for prm, g in ((weight, dw), (bias, db)):
                                if prm is None or g is None or id(prm) not in slot:
                                    continue
                                # a parameter's gradient must obey the parameter's layout and dtype (the fused Adam pairs elements by
                                # storage offset; AccumulateGrad, which would re-lay it, is not in the picture)
                                if g.shape != prm.shape or g.stride() != prm.stride() or g.dtype != prm.dtype:
                                    g = torch.empty_like(prm).copy_(g.reshape(prm.shape) if g.shape != prm.shape else g)
                                assert cap.s_gparams[slot[id(prm)]] is None, "a deferred parameter also received a gradient on the chain"
                                cap.s_gparams[slot[id(prm)]] = g
                    cap.keep = items
            cap.s_out = tuple(o.detach() if torch.is_tensor(o) else o for o in outs)
            cap.bound = False
        finally:
            if gc_was_on:
                gc.enable()
            dense2d.WS_PRIVATE = False
            dense2d.CAPTURE_PACKS = None
            CAPTURE_LOCK.release()
        return cap
