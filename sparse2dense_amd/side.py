"""Weight-gradient launches on a second HIP stream (opt-in since r05: S2D_WGRAD_STREAM=1 | dense | sparse | aux, or side.enable(...);
bench.py measures it against the single-stream run on the box it runs on and keeps it only where it is faster - r04's default-on
cost 11 ms per step on a slow host for 1.4 ms gained on a fast one).

In the backward of a conv layer only the data gradient is on the chain to the next layer; the weight gradient (and its
split-K fold, and the bias gradient that rides on it) is needed by nobody before the optimizer.  On one stream both are
serialised with the chain's many small kernels (batch-norm finalize launches of ONE workgroup, folds, row passes that
cannot fill 256 CUs); on a second stream the weight-gradient kernels fill those holes.  The reference gets the same
overlap from cuDNN / spconv kernels queued behind each other by the autograd engine on a device with concurrent kernel
execution (/root/reference/det3d/torchie/trainer/trainer.py:775-811 `loss.backward()`); here it is explicit.

Protocol (one side stream per device):
  * `run(weight, fn, *inputs)` inside an autograd backward: the side stream waits for an event recorded on the current
    stream (everything `fn` reads has been enqueued there), `fn()` runs with the side stream current (its launches, its
    workspace - `_ws` is keyed by stream - and its output allocations belong to that stream).  A reference to every input
    is kept until the join: the caching allocator
    cannot hand their memory out meanwhile, and the autograd engine cannot accumulate a later gradient INTO one of them in
    place on the main stream (it does that with a buffered gradient nobody else references - e.g. the gradient of a
    residual add, which is both the `dy` of the branch's last conv and the buffered gradient of the block input).
  * the returned gradient is handed to autograd unchanged.  With `weight.grad is None` AccumulateGrad adopts the tensor
    without launching anything, so no main-stream kernel touches it before the join.  The data-parallel buckets
    (dp.GradBuckets) copy gradients into their flat buffers when a bucket is complete: `_launch` does that ON the side stream
    (`stream_after`: behind the weight gradients enqueued so far and behind an event of the chain, for the gradients the
    chain produced) and issues the all-reduce from there - the chain itself never waits for the side stream's backlog (a join
    per bucket cost 1.7 ms per step at one rank).  Every other case (accumulation into an existing
    `.grad`, foreign hooks, graph capture, the profiling pass of bench.py, double backward) takes the plain path: `fn()` on
    the current stream.
  * the join: the first `run` of a backward pass queues an engine callback that makes the stream `backward()` was called on
    wait for the side stream; `join()` does the same explicitly (train_step / solver call it before reading gradients).
"""
import os

import torch

KINDS = ("dense", "sparse", "aux", "pcr")   # pcr: the PCR head's up-sampler / 1x1x1 weight gradients (r06: on the eager stream too, see _parse)


def _parse(v):
    """"1" / "all": every wired layer kind; a comma list of kinds for A/B runs; anything else: off.  Kinds: dense (3x3 / 1x1 NHWC convs),
    sparse (sparse convs), aux (the small ones: depth-wise 7x7, CenterHead output convs, 2x2 / 4x4 stride-2 forms, sparse-conv bias sums).
    pcr (the PCR head's up-sampler / 1x1x1 weight gradients): r04 time-neutral, r05 kept off the eager stream - the 16 -> 3 up-sampler's MFMA weight
    gradient running beside `pcr_level_bwd_dense` made that kernel's packed-FP32 high lanes timing-dependent (rule 36; 68 of 70 stress runs differed).
    r06 builds losses.hip without packed FP32 by default: 0 of 60 runs at 12 k points, 0 of 8 with the double-launch instrument, 0 of 4 at 4 x 150 k
    (tools/side_stress.py <n> aux+dense+pcr+sparse:0), and 1.2 ms of weight-gradient kernels leave the chain: 18.0 -> 17.7 ms per eager step."""
    v = (v or "0").strip().lower()
    if v in ("1", "all"):
        if os.environ.get("S2D_BUILD_LOSSES_SLP") == "1":   # (the packed-FP32 build of losses.hip: rule 36's victim is back - keep the PCR kinds on the chain)
            return set(KINDS) - {"pcr"}
        return set(KINDS)
    return {k for k in v.split(",") if k in KINDS}


MODE = _parse(os.environ.get("S2D_WGRAD_STREAM", "0"))

_streams = {}    # device index -> torch.cuda.Stream
_pending = {}    # device index -> True while side work has been launched since the last join
_queued_for = {}  # device index -> graph-task id of the backward pass that has its join callback queued
_events = {}     # device index -> ring of reusable events
_keep = {}       # device index -> list of the tensors the side stream's work since the last join reads
stats = {"side": 0, "plain": 0}
CHECK = os.environ.get("S2D_WGRAD_STREAM_CHECK", "0") == "1"   # tests: remember what was handed to autograd
_handed = []


def adopted():
    """(test hook, CHECK mode) True when every gradient produced on the side stream since the last call IS the parameter's .grad"""
    ok = all(p is None or (w.grad is not None and w.grad.data_ptr() == p) for w, p in _handed)
    n = len(_handed)
    _handed.clear()
    return ok, n


def enable(on=True):
    """True / False or a mode string as in S2D_WGRAD_STREAM"""
    global MODE
    MODE = _parse(on) if isinstance(on, str) else _parse("1" if on else "0")


def _side(dev):
    s = _streams.get(dev)
    if s is None:
        s = torch.cuda.Stream(device=dev)
        _streams[dev] = s
        _events[dev] = [[torch.cuda.Event() for _ in range(512)], 0]
        _keep[dev] = []
    return s


def _event(dev):
    ring = _events[dev]
    ev = ring[0][ring[1]]
    ring[1] = (ring[1] + 1) % len(ring[0])
    return ev


_pass_ids = {}   # device index -> (graph-task id, ids of the parameters that went to the side stream in that pass)


def usable(weight, kind="dense", bias=None):
    if kind not in MODE or not weight.is_cuda or weight.grad is not None:
        return False
    if getattr(weight, "_backward_hooks", None):
        return False
    if bias is not None and (bias.grad is not None or getattr(bias, "_backward_hooks", None)):
        return False   # AccumulateGrad would add into the existing bias.grad in place on the chain, before the join
    # a parameter that receives a SECOND gradient in one backward pass (a module applied twice, tied weights): the engine sums the two
    # on the chain's stream without an event from the side stream - the second one takes the plain path behind a join
    gid = torch._C._current_graph_task_id()
    ent = _pass_ids.get(weight.device.index)
    if ent is None or ent[0] != gid:
        ent = _pass_ids[weight.device.index] = (gid, set())
    if id(weight) in ent[1]:
        join(weight.device.index)
        return False
    hooks = getattr(weight, "_post_accumulate_grad_hooks", None)
    if hooks:   # the data-parallel buckets' hooks only count arrivals; dp.GradBuckets._launch joins before it copies gradients
        from . import dp
        if len(hooks) != 1 or id(weight) not in dp._BUCKETERS:
            return False
    if torch.is_grad_enabled() or torch.cuda.is_current_stream_capturing():   # double backward / capture
        return False
    from . import hip_ops as H
    return H.PROFILE is None


def join(device=None):
    """make the current stream wait for the weight-gradient stream(s)"""
    for dev, on in list(_pending.items()):
        if on and (device is None or dev == device):
            torch.cuda.current_stream(dev).wait_stream(_streams[dev])
            _pending[dev] = False
            _keep[dev].clear()   # everything enqueued on this stream from here on is ordered behind the side work
    for dev, on in list(_gpending.items()):
        if on and (device is None or dev == device):
            torch.cuda.current_stream(dev).wait_stream(_gstreams[dev])
            _gpending[dev] = False


def stream_after(dev):
    """A stream on which every weight gradient launched off the chain so far is complete in stream order - the eager side stream, or the stream
    the graphed segment's weight-gradient graph was replayed on (`replay_on_side`) - made to wait for everything enqueued so far on the current
    stream; None when no such work is pending.  For consumers of the gradients that need not be on the chain either (dp.GradBuckets: bucket
    copies + all-reduce launch)."""
    if dev is None:
        return None
    sp, gp = bool(_pending.get(dev)), bool(_gpending.get(dev))
    if not (sp or gp):
        return None
    target = _gstreams[dev] if gp else _streams[dev]
    ev = _event(dev)
    ev.record(torch.cuda.current_stream(dev))
    target.wait_event(ev)
    if gp and sp:   # both kinds of side work pending: the graph stream also waits for the eager side stream
        ev = _event(dev)
        ev.record(_streams[dev])
        target.wait_event(ev)
    return target


# ---- inside a HIP-graph capture (graphed.GraphedSegment): the same overlap as parallel branches of the backward graph ----------------
# A weight-gradient launch group forks off the capturing stream (event record / wait = a graph dependency) onto a second capturing
# stream and is joined once, at the end of the backward capture: in the replayed graph the groups are branches that depend on nothing
# but their inputs, and the runtime runs them beside the chain - with no host work at all per step.
GRAPH_KINDS = _parse(os.environ.get("S2D_GRAPH_FORK", "0"))   # layer kinds forked inside a capture (bench.py measures "dense,aux" vs none)
_cap = {}   # device index -> dict(stream, keep, pending)


def graph_fork(kinds):
    """kinds as in S2D_WGRAD_STREAM; takes effect for captures made afterwards"""
    global GRAPH_KINDS
    GRAPH_KINDS = _parse(kinds) if isinstance(kinds, str) else _parse("1" if kinds else "0")


def _run_forked(weight, fn, inputs):
    dev = weight.device.index
    st = _cap.get(dev)
    if st is None:
        st = _cap[dev] = dict(stream=torch.cuda.Stream(device=dev), keep=[], pending=False)
    cur = torch.cuda.current_stream(dev)
    ev = torch.cuda.Event()
    ev.record(cur)
    st["stream"].wait_event(ev)
    torch.cuda.set_stream(st["stream"])
    try:
        out = fn()
    finally:
        torch.cuda.set_stream(cur)
    # the branch reads its inputs while the chain runs on: their blocks of the graph's memory pool must not be handed to a later chain
    # allocation of the same capture - keep them (and the outputs) referenced until the capture is over
    st["keep"].append((inputs, out))
    st["pending"] = True
    stats["forked"] = stats.get("forked", 0) + 1
    return out


def join_capture():
    """inside the capture, after the last forked group: the capturing stream waits for the fork stream (an unjoined branch is an error
    at hipStreamEndCapture)"""
    for dev, st in _cap.items():
        if st["pending"]:
            ev = torch.cuda.Event()
            ev.record(st["stream"])
            torch.cuda.current_stream(dev).wait_event(ev)
            st["pending"] = False


def capture_done():
    """after the capture: the kept tensors may go (their storage stays reserved for the graph by the pool)"""
    for st in _cap.values():
        st["keep"].clear()


# ---- deferral into a graph of their own (r05, the default of the graphed dense segment) -------------------------------------------
# While graphed.GraphedSegment captures the backward, a weight-gradient group of a kind in GRAPH_DEFER is not launched at all: its
# closure is queued, the layer's backward returns no gradient for the weight (and bias), and after the chain's graph is closed the
# queued closures are captured, one after the other, into a SECOND graph.  A step then replays the chain graph on the launch stream -
# the gradient of the BEV map is out as early as possible - and the weight-gradient graph on the side stream, where it runs beside the
# eager sparse backward (latency-bound gather kernels next to matrix-core contractions); train_step's side.join() orders the optimizer
# behind it.  Both graphs are linear chains: no intra-graph branches (see _run_forked: those need host-side signal handling).
GRAPH_DEFER = _parse(os.environ.get("S2D_GRAPH_DEFER", "dense,aux,pcr"))
_deferred = []          # (weight, bias, fn, inputs) queued by the capture in progress
_deferred_ids = set()   # id(parameter) queued: a parameter that gets a second gradient in one pass takes the plain path


class _Deferred:
    def __repr__(self):
        return "<weight gradient deferred to the segment's second graph>"


DEFERRED = _Deferred()


def undefer(t):
    """what a layer's backward returns to autograd for a deferred gradient"""
    return None if t is DEFERRED else t


def graph_defer(kinds):
    """kinds as in S2D_WGRAD_STREAM; takes effect for captures made afterwards"""
    global GRAPH_DEFER
    GRAPH_DEFER = _parse(kinds) if isinstance(kinds, str) else _parse("1" if kinds else "0")


def take_deferred():
    items = list(_deferred)
    _deferred.clear()
    _deferred_ids.clear()
    return items


_gstreams = {}    # device index -> the stream the segment's weight-gradient graph is replayed on (its own: the eager weight-gradient groups of
_gpending = {}    # the sparse layers keep the side stream to themselves and run beside it)


def replay_on_side(graph, dev):
    """launch a captured graph on the graph stream behind everything enqueued so far on the current stream; joined like any side work"""
    _side(dev)   # (event ring)
    gs = _gstreams.get(dev)
    if gs is None:
        gs = _gstreams[dev] = torch.cuda.Stream(device=dev)
    cur = torch.cuda.current_stream(dev)
    ev = _event(dev)
    ev.record(cur)
    gs.wait_event(ev)
    torch.cuda.set_stream(gs)
    try:
        graph.replay()
    finally:
        torch.cuda.set_stream(cur)
    _gpending[dev] = True
    gid = torch._C._current_graph_task_id()
    if gid >= 0 and _queued_for.get(dev) != gid:
        _queued_for[dev] = gid
        torch.autograd.Variable._execution_engine.queue_callback(lambda: join(dev))


_PCR_ONLY = None if not os.environ.get("S2D_SIDE_PCR_ONLY") else {int(v) for v in os.environ["S2D_SIDE_PCR_ONLY"].split(",")}
_pcr_count = [None, 0]


def run(weight, fn, *inputs, kind="dense", bias=None, pair=False):
    """fn() -> gradient tensor(s) of `weight` (and its bias); on the side stream when the protocol above allows it.
    bias: the bias parameter when fn returns (dw, db); pair: fn returns a 2-tuple"""
    if weight.is_cuda and torch.cuda.is_current_stream_capturing():
        if not torch.is_grad_enabled():
            if kind in GRAPH_DEFER and id(weight) not in _deferred_ids:
                _deferred.append((weight, bias, fn, inputs))
                _deferred_ids.add(id(weight))
                stats["deferred"] = stats.get("deferred", 0) + 1
                return (DEFERRED, DEFERRED) if pair else DEFERRED
            if kind in GRAPH_KINDS:
                return _run_forked(weight, fn, inputs)
        stats["plain"] += 1
        return fn()
    if not usable(weight, kind, bias):
        stats["plain"] += 1
        return fn()
    if kind == "pcr" and _PCR_ONLY is not None:   # debugging aid: only the listed pcr-kind calls of a backward pass (in call order) leave the chain
        gid = torch._C._current_graph_task_id()
        if _pcr_count[0] != gid:
            _pcr_count[:] = [gid, 0]
        k = _pcr_count[1]
        _pcr_count[1] += 1
        if k not in _PCR_ONLY:
            stats["plain"] += 1
            return fn()
    dev = weight.device.index
    _pass_ids[dev][1].add(id(weight))
    side = _side(dev)
    cur = torch.cuda.current_stream(dev)
    ev = _event(dev)
    ev.record(cur)
    side.wait_event(ev)
    # (host cost matters here - ~135 calls per backward on the thread that also launches the chain: set_stream pairs instead of the
    # `torch.cuda.stream` context manager, 0.9 vs 5.7 us; no completion event per call, see the keep list below)
    torch.cuda.set_stream(side)
    try:
        out = fn()
        # AccumulateGrad adopts a gradient only if it obeys the parameter's layout; otherwise it CLONES it - a main-stream launch
        # before the join (and a copy kernel per layer and step on the chain: channels_last conv weights).  The first returned
        # tensor is the weight gradient: it is re-laid here, on the side stream.
        dw = out[0] if isinstance(out, (tuple, list)) else out
        if dw is not None and dw.shape == weight.shape and (dw.stride() != weight.stride() or dw.dtype != weight.dtype):
            dw = torch.empty_like(weight).copy_(dw)
            out = (dw,) + tuple(out[1:]) if isinstance(out, (tuple, list)) else dw
            stats["relaid"] = stats.get("relaid", 0) + 1
    finally:
        torch.cuda.set_stream(cur)
    if os.environ.get("S2D_SIDE_DEBUG_SYNC") == "1":   # debugging aid: same streams and allocation pattern, no concurrency
        torch.cuda.synchronize()
    if dw is not None and dw.shape != weight.shape:   # cannot be adopted: the main stream waits here
        cur.wait_stream(side)
        stats["waited"] = stats.get("waited", 0) + 1
    elif CHECK:
        _handed.append((weight, None if dw is None else dw.data_ptr()))
    _keep[dev].append(inputs)   # until the join (a few GB of gradient tensors at the benchmark's size; HBM is 288 GB)
    _pending[dev] = True
    # one join callback per backward pass, keyed on the engine's graph-task id (a flag "already queued" would go stale when a pass dies
    # before its callbacks run, and every later pass would go without its join)
    gid = torch._C._current_graph_task_id()
    if _queued_for.get(dev) != gid or gid < 0:
        _queued_for[dev] = gid
        torch.autograd.Variable._execution_engine.queue_callback(lambda: join(dev))
    stats["side"] += 1
    return out
