"""BEV 3x3 convolutions on the hand-written NHWC bf16 MFMA kernel (csrc/conv2d_nhwc.hip).

`Conv3x3(nn.Conv2d)` keeps nn.Conv2d's parameters and state_dict keys (it IS one), so the
reference's checkpoints and our golden fixtures load unchanged
(/root/reference/det3d/models/necks/rpn.py:126-145, bbox_heads/center_head.py:209-232).
The HIP path is taken for CUDA inputs under bf16 autocast when the layer is 3x3 / stride 1 /
padding 0|1 / no groups / no dilation and both channel counts are multiples of 64; forward and
data gradient run on the kernel, the weight gradient on MIOpen (aten.convolution_backward) or, where it measured
faster, on csrc/conv2d_wgrad.hip.  The stride-2 layer of a block runs its forward on the kernel too.
Everything else (fp32 runs, CPU goldens, the 2-channel output convs) is the stock layer.
"""
import ctypes

import torch

from . import collective as _collective
from torch import nn
import torch.nn.functional as F

from . import _lib
from ._lib import check

ENABLED = True   # bench / tests can switch the kernel off to A/B against MIOpen

_zero_pages = {}


def _zero_page(device):
    z = _zero_pages.get(device)
    if z is None:
        z = torch.zeros(64, dtype=torch.uint8, device=device)
        _zero_pages[device] = z
    return z


def _ptr(t):
    # a plain int: ctypes converts it for the c_void_p parameters (cheaper than building a c_void_p per argument)
    return None if t is None else t.data_ptr()


def _stream():
    # raw hipStream_t of torch's current stream on the current device (the private accessor is ~5x cheaper than
    # torch.cuda.current_stream().cuda_stream, and this runs once per kernel launch, ~500 times a step)
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def supported(cin, cout):
    return bool(_lib.load().s2d_conv2d3x3_supported(int(cin), int(cout)))


import weakref

_pack_cache = {}   # id(weight) -> [weakref(weight), data_ptr, version, {variant: packed}]
CAPTURE_PACKS = None   # set by graphed.GraphedSegment._capture for the duration of a capture: {(id(weight), variant): image built inside it}


def cached_pack(weight, variant, build):
    """Packed-weight images are rebuilt when the parameter changes (its autograd version counter moves on every in-place
    update such as an optimizer step or load_state_dict, its data_ptr on re-assignment), not on every call.  In-place
    edits through `weight.data` do not move the counter: call clear_pack_cache() after such surgery."""
    weight = getattr(weight, "_s2d_origin", weight)   # graphed.GraphedSegment captures with leaf aliases of the parameters: one image per parameter
    key = id(weight)
    ent = _pack_cache.get(key)
    if ent is None or ent[0]() is not weight or ent[1] != weight.data_ptr() or ent[2] != weight._version:
        ent = [weakref.ref(weight, lambda _r, k=key: _pack_cache.pop(k, None)), weight.data_ptr(), weight._version, {}]
        _pack_cache[key] = ent
    hit = ent[3].get(variant)
    if hit is not None and CAPTURE_PACKS is not None and (key, variant) not in _repack:
        # A capture is recording and this image has no registered in-place refresh: refresh_pack_cache() FREES it after the next optimizer
        # step (the fused Adam writes through raw pointers, so the capture's staleness check cannot see that), while the graph would go on
        # reading its address.  Only an image built inside this very capture (graph-pool memory, rebuilt by every replay) may be baked in:
        # a forward-only capture taken with a warm cache - eval / teacher calls between optimizer steps - builds its own.  ADVICE r05.
        own = CAPTURE_PACKS.get((key, variant))
        if own is None:
            own = CAPTURE_PACKS[(key, variant)] = build()
        return own
    if hit is None:
        hit = build()
        ent[3][variant] = hit
        if CAPTURE_PACKS is not None:
            CAPTURE_PACKS[(key, variant)] = hit
    return hit


def cached_pack_has(weight, variant):
    weight = getattr(weight, "_s2d_origin", weight)
    ent = _pack_cache.get(id(weight))
    return (ent is not None and ent[0]() is weight and ent[1] == weight.data_ptr() and ent[2] == weight._version and variant in ent[3])


def cached_pack_put(weight, variant, value):
    """store an image that was built together with another variant (one launch for a layer's forward and data-gradient operands)"""
    cached_pack(weight, variant, lambda: value)


def clear_pack_cache():
    _pack_cache.clear()
    _repack.clear()
    _repack_state.update(sig=None, graph=None, last=None, stable=0)


# Images whose build is ONE launch from the parameter's own storage into a fixed buffer register that launch here.  After an
# in-place update through raw pointers (the fused Adam, solver._step_hip) refresh_pack_cache() re-runs all registered launches of
# the trainable parameters into the SAME buffers - the cache entries stay valid, frozen parameters (a distillation teacher) keep
# their images - and, once the set of launches has been the same for two steps, replays them as one HIP graph: the ~50 pack
# launches that used to be scattered through the next forward become one graph launch at the end of the step.  Measured r03 (B=4 S2D
# student step): 24.6-24.8 ms with the graph, 24.6 ms with drop-and-rebuild - the packs were not on the critical path, so the graph is
# opt-in: S2D_PACK_GRAPH=1 graph, 0 (default) eager in-place re-launches at the end of the step, off = drop-and-rebuild (r02).
_repack = {}   # (id(weight), variant) -> closure
_repack_state = dict(sig=None, graph=None, last=None, stable=0)


def register_repack(weight, variants, src, launch, conv2d=None):
    """src: the tensor the launch reads (must alias the parameter's storage - a converted / re-laid copy would go stale).
    conv2d = (w_ptr_tensor, cin, cout, taps, nhwc, transpose_flip, packed_fwd, packed_dgrad | None): the launch's arguments, for launches
    that s2d_conv2d_pack_batch_bf16 can serve - refresh_pack_cache() then packs all such layers in ONE launch (r04)"""
    weight = getattr(weight, "_s2d_origin", weight)
    if src.data_ptr() != weight.data_ptr() or src.dtype != weight.dtype:
        return
    if conv2d is not None:
        launch.conv2d = conv2d
    for v in variants:
        if _repack.get((id(weight), v)) is not launch:   # a re-packed weight: a captured graph would replay into the OLD image
            _repack_state.update(sig=None, graph=None, last=None, stable=0)
        _repack[(id(weight), v)] = launch


def _run_pack_launches(fns):
    """the registered single-launch re-packs; those that carry conv2d arguments go through the batch entry, 64 layers per launch"""
    import ctypes
    batch = [fn for fn in fns if getattr(fn, "conv2d", None) is not None]
    for fn in fns:
        if getattr(fn, "conv2d", None) is None:
            fn()
    lib = _lib.load() if batch else None
    for i in range(0, len(batch), 64):
        part = [fn.conv2d for fn in batch[i:i + 64]]
        n = len(part)
        vp, i32 = ctypes.c_void_p * n, ctypes.c_int32 * n
        check(lib.s2d_conv2d_pack_batch_bf16(n, vp(*[d[0].data_ptr() for d in part]), i32(*[d[1] for d in part]), i32(*[d[2] for d in part]),
                                             i32(*[d[3] for d in part]), i32(*[int(d[4]) for d in part]), i32(*[int(d[5]) for d in part]),
                                             vp(*[d[6].data_ptr() for d in part]), vp(*[(None if d[7] is None else d[7].data_ptr()) for d in part]),
                                             _stream()), "s2d_conv2d_pack_batch_bf16")


def refresh_pack_cache():
    import os
    mode = os.environ.get("S2D_PACK_GRAPH", "0")
    if mode == "off":
        return clear_pack_cache()
    launches, sig_items = {}, []
    for key, ent in list(_pack_cache.items()):
        w = ent[0]()
        if w is None or ent[1] != w.data_ptr() or ent[2] != w._version:
            _pack_cache.pop(key, None)
            continue
        for v in list(ent[3]):
            fn = _repack.get((key, v))
            if fn is None:
                del ent[3][v]            # no single-launch rebuild registered (sparse images, derived matrices): rebuilt at the next use
            elif w.requires_grad:
                launches[id(fn)] = fn    # one launch may fill two variants (forward + data-gradient operand)
                sig_items.append((key, v, ent[1], id(fn)))   # (weight, variant, weight storage, closure): closure ids alone can be recycled
        if not ent[3]:
            _pack_cache.pop(key, None)
    for k in [k for k in _repack if k[0] not in _pack_cache]:
        del _repack[k]
    if not launches:
        return
    st = _repack_state
    sig = tuple(sorted(sig_items, key=repr))
    if st["graph"] is not None and st["sig"] == sig:
        st["graph"].replay()
        return
    _run_pack_launches(list(launches.values()))
    st["stable"] = st["stable"] + 1 if st["last"] == sig else 0
    st["last"] = sig
    if mode != "0" and st["stable"] >= 2 and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            _run_pack_launches(list(launches.values()))
        st.update(sig=sig, graph=g)


def pack_weights(weight, transpose_flip=False, with_dgrad=False):
    """weight fp32 [Cout,Cin,3,3] -> bf16 LDS image of the forward (or data-gradient) operand (cached, see cached_pack).  with_dgrad (a
    training forward whose input needs a gradient): the data-gradient image of the same step is packed in the same launch."""
    if with_dgrad and not transpose_flip and not cached_pack_has(weight, ("conv3x3", False)) and not cached_pack_has(weight, ("conv3x3", True)):
        lib = _lib.load()
        cout, cin = weight.shape[0], weight.shape[1]
        if lib.s2d_conv2d3x3_supported(cin, cout) and lib.s2d_conv2d3x3_supported(cout, cin):
            w = weight.detach().float()
            nhwc = (not w.is_contiguous()) and w.is_contiguous(memory_format=torch.channels_last)
            if not nhwc:
                w = w.contiguous()
            pf = torch.empty(9 * cin * cout, dtype=torch.bfloat16, device=weight.device)
            pd = torch.empty(9 * cin * cout, dtype=torch.bfloat16, device=weight.device)
            launch = lambda: check(lib.s2d_conv2d3x3_pack_weights_pair_bf16(_ptr(w), cin, cout, int(nhwc), _ptr(pf), _ptr(pd), _stream()),
                                   "s2d_conv2d3x3_pack_weights_pair_bf16")
            launch()
            cached_pack_put(weight, ("conv3x3", False), pf)
            cached_pack_put(weight, ("conv3x3", True), pd)
            register_repack(weight, [("conv3x3", False), ("conv3x3", True)], w, launch, conv2d=(w, cin, cout, 9, nhwc, 0, pf, pd))
    return cached_pack(weight, ("conv3x3", bool(transpose_flip)), lambda: _pack_weights(weight, transpose_flip))


def _pack_weights(weight, transpose_flip):
    lib = _lib.load()
    cout, cin = weight.shape[0], weight.shape[1]
    w = weight.detach().float()
    nhwc = (not w.is_contiguous()) and w.is_contiguous(memory_format=torch.channels_last)
    if not nhwc:
        w = w.contiguous()
    packed = torch.empty(9 * cin * cout, dtype=torch.bfloat16, device=weight.device)
    pc_in, pc_out = (cout, cin) if transpose_flip else (cin, cout)
    launch = lambda: check(lib.s2d_conv2d3x3_pack_weights_bf16(_ptr(w), pc_in, pc_out, int(transpose_flip), int(nhwc), _ptr(packed), _stream()),
                           "s2d_conv2d3x3_pack_weights_bf16")
    launch()
    register_repack(weight, [("conv3x3", bool(transpose_flip))], w, launch, conv2d=(w, pc_in, pc_out, 9, nhwc, int(transpose_flip), packed, None))
    return packed


BN_BWD_FOLD = 0   # (read from the environment below)
STATS = {"bn_bwd_folded": 0, "bn_bwd_reduced": 0}   # batch-norm backward passes that took the conv epilogue's sums / their own reduction pass


def _bn_src_of(x):
    """the (z, fin, act) tag a training-mode FastBatchNorm2d leaves on its output (see _BNRowFn.forward), if x still is that tensor"""
    src = getattr(x, "_s2d_bn_src", None) if BN_BWD_FOLD else None
    if src is None or src[0].shape != x.shape or src[3] != x._version or (src[2] == 2 and int(BN_BWD_FOLD) < 2):
        return None
    return src


def _tag_bn_bwd(dx, partial):
    """dx is the output gradient of a batch norm and `partial` holds that layer's backward sums (conv epilogue): _BNRowFn.backward reads the
    tag - valid only while dx is the very tensor the kernel wrote (autograd accumulating a second consumer's gradient into it bumps
    the version, a fresh sum tensor has no tag)"""
    dx._s2d_bnbwd = (partial, dx._version)
    return dx


def conv3x3_nhwc(x, packed, bias, cin, cout, pad, stride=1, bn_stats=False, bn_bwd=None):
    """x: bf16 [N,cin,H,W] in channels_last memory -> bf16 [N,cout,Ho,Wo] channels_last.  bn_stats: also returns the
    per-tile (sum, sumsq) slabs [tiles, 2, cout] of the output (the statistics pass of a following batch norm).
    bn_bwd = (z, fin, act, ...): this launch is the DATA GRADIENT of a conv that read act(bn(z)); returns (y, partial) with the batch norm's
    backward sums per tile (csrc/conv2d_nhwc.hip `BnBwd`)."""
    lib = _lib.load()
    assert x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last) and x.shape[1] == cin
    n, _, h, w = x.shape
    ho, wo = (h + 2 * pad - 3) // stride + 1, (w + 2 * pad - 3) // stride + 1
    y = torch.empty((n, cout, ho, wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    from . import hip_ops as H
    rec = None
    if H.PROFILE is not None:   # bench.py roofline pass: HIP events on the launch stream
        rec = dict(kernel="conv3x3_nhwc_bf16", tag="dense", cin=cin, cout=cout, n_out=n * ho * wo, kvol=9, pairs=None,
                   dense=True, in_pixels=n * h * w, pad=pad, stride=stride,
                   tile_rows=lib.s2d_conv2d3x3_tile_rows(n, h, w, cin, cout, pad, stride),
                   start=torch.cuda.Event(enable_timing=True),
                   end=torch.cuda.Event(enable_timing=True))
        rec["start"].record()
    partial = (torch.empty((lib.s2d_conv2d3x3_stats_tiles(n, h, w, cin, cout, pad, stride), 2, cout), dtype=torch.float32,
                           device=x.device) if (bn_stats or bn_bwd is not None) else None)
    if bn_bwd is not None:
        z, fin, act = bn_bwd[0], bn_bwd[1], bn_bwd[2]
        assert bias is None and not bn_stats and stride == 1 and z.shape == y.shape and z.dtype == torch.bfloat16 \
            and z.is_contiguous(memory_format=torch.channels_last) and fin.shape == (4, cout)
        fp = fin.data_ptr()
        check(lib.s2d_conv2d3x3_nhwc_bf16_bnbwd(_ptr(x), _ptr(packed), _ptr(_zero_page(x.device)), n, h, w, cin, cout, pad, _ptr(y), _ptr(z),
                                                fp + 8 * cout, fp + 12 * cout, int(act), _ptr(partial), _stream()),
              "s2d_conv2d3x3_nhwc_bf16_bnbwd")
    else:
        check(lib.s2d_conv2d3x3_nhwc_bf16(_ptr(x), _ptr(packed), _ptr(bias), _ptr(_zero_page(x.device)), n, h, w, cin, cout,
                                          pad, stride, _ptr(y), _ptr(partial), _stream()), "s2d_conv2d3x3_nhwc_bf16")
    if rec is not None:
        rec["end"].record()
        H.PROFILE.append(rec)
    return (y, partial) if (bn_stats or bn_bwd is not None) else y


# which weight gradients take the hand-written transpose-read kernel (csrc/conv2d_wgrad.hip) instead of MIOpen:
# "auto" = every stride-1 3x3 conv the kernel supports (r01 end state: 111-116 us against MIOpen's 114-120 us on the
# 128->128 / 256->256 BEV layers, 181 vs 224 us on 512->64 - and no SubTensorOp / cast launches or MIOpen host-side
# solver lookup around it); True / False = all / none
import os as _os
from . import side as _side
WGRAD_HIP = {"0": False, "1": True}.get(_os.environ.get("S2D_WGRAD_HIP", ""), "auto")
# r06: the data-gradient conv behind a conv -> BatchNorm2d -> ReLU layer can write that batch norm's backward sums from its epilogue (the reduction
# pass over (dY, z) of the batch-norm backward is skipped): S2D_BN_BWD_FOLD=1 (ReLU / no activation) or 2 (also GELU).  Measured on the benchmarked
# step (4 x 150 k points): the ten ReLU layers of the RPN trunk save 137 us of `row_reduce` and ten launches for ~90 us of longer conv epilogues (z is
# read once more, one exposed load latency per tile); behind a GELU the erf / exp per element cost +40 us per launch against the 30 us pass they
# replace.  The step's wall time does not move (218.4 vs 218.5 frames/s: the eager step is bound by the launch thread, 17.9 ms of enqueue per
# 18.0 ms step) while the dominant kernel's own launches get 10 % longer, so the default stays the separate pass ("0").
BN_BWD_FOLD = {"1": 1, "2": 2}.get(_os.environ.get("S2D_BN_BWD_FOLD", "0"), 0)


def _wgrad_hip(cin, cout):
    if WGRAD_HIP == "auto":
        return True
    return bool(WGRAD_HIP)


def conv3x3_wgrad(x, dy, pad, want_db=False):
    """x bf16 NHWC [N,cin,H,W] (forward input), dy bf16 NHWC [N,cout,Ho,Wo] -> fp32 [cout,cin,3,3]; want_db: also the per-channel sums
    of dy (the bias gradient), accumulated by the same launch -> (dw, db)"""
    lib = _lib.load()
    n, cin, h, w = x.shape
    cout = dy.shape[1]
    assert x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16
    assert x.is_contiguous(memory_format=torch.channels_last) and dy.is_contiguous(memory_format=torch.channels_last)
    assert dy.shape[2] == h + 2 * pad - 2 and dy.shape[3] == w + 2 * pad - 2
    dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=x.device)
    db = torch.empty((cout,), dtype=torch.float32, device=x.device) if want_db else None
    ws = _ws(lib.s2d_conv2d3x3_wgrad_workspace_bytes(n, h, w, cin, cout, pad), x.device)
    from . import hip_ops as H
    rec = None
    if H.PROFILE is not None:   # bench.py roofline pass (`roofline_wgrad`): the transpose-read contraction + its split-K fold, one event pair
        rec = dict(kernel="conv3x3_wgrad", tag="dense_wgrad", cin=cin, cout=cout, n_out=n * dy.shape[2] * dy.shape[3], kvol=9, pairs=None, dense=True,
                   in_pixels=n * h * w, pad=pad, stride=1, kname="conv3x3_wgrad_kernel + conv3x3_wgrad_reduce_kernel",
                   start=torch.cuda.Event(enable_timing=True), end=torch.cuda.Event(enable_timing=True))
        rec["start"].record()
    check(lib.s2d_conv2d3x3_wgrad_nhwc_bf16(_ptr(x), _ptr(dy), _ptr(_zero_page(x.device)), n, h, w, cin, cout, pad, _ptr(dw), _ptr(db), _ptr(ws),
                                            ws.numel(), _stream()), "s2d_conv2d3x3_wgrad_nhwc_bf16")
    if rec is not None:
        rec["end"].record()
        H.PROFILE.append(rec)
    return (dw, db) if want_db else dw


def _nhwc_bf16(t):
    return t.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


class _Conv3x3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, pad, stride, bn_stats, bn_src=None):
        xb = _nhwc_bf16(x)
        cout, cin = weight.shape[0], weight.shape[1]
        ctx.save_for_backward(xb, weight)
        ctx.pad, ctx.stride = pad, stride
        # x is the output of a training-mode batch norm (+ activation): the data gradient below is that layer's dY
        ctx.bn_src = bn_src if (bn_src is not None and xb is x and stride == 1
                                and _lib.load().s2d_conv2d3x3_bnbwd_supported(cout, cin, 1, 1)) else None
        ctx.has_bias = bias is not None
        ctx.bias_p = bias   # (the parameter itself: side.run hands a deferred bias gradient to it)
        b = None if bias is None else bias.detach().float().contiguous()
        both = bool(ctx.needs_input_grad[0] and stride == 1)   # this step's backward will ask for the data-gradient operand
        if bn_stats:
            y, partial = conv3x3_nhwc(xb, pack_weights(weight, with_dgrad=both), b, cin, cout, pad, stride, bn_stats=True)
            ctx.mark_non_differentiable(partial)
            ctx.set_materialize_grads(False)   # no zero-filled gradient tensor for the statistics output in every backward
            return y, partial
        return conv3x3_nhwc(xb, pack_weights(weight, with_dgrad=both), b, cin, cout, pad, stride)

    @staticmethod
    def backward(ctx, dy, *_unused):
        xb, weight = ctx.saved_tensors
        pad, stride = ctx.pad, ctx.stride
        cout, cin = weight.shape[0], weight.shape[1]
        dyb = _nhwc_bf16(dy)
        dx = dw = db = None
        if stride == 2 and _s2_backward_ok(xb, dyb, cin, cout, pad):
            # the strided layer (one per RPN block): data gradient = the transposed form split into its four output parity classes
            # (1 + 2 + 2 + 4 dense taps), weight gradient = the stride-2 contraction (csrc/conv2d_nhwc.hip, conv2d_wgrad.hip)
            if ctx.needs_input_grad[0]:
                dx = conv_up(dyb, weight, None, 3)
            if ctx.needs_input_grad[1]:
                dw = _side.run(weight, lambda: conv_s2_wgrad(dyb, xb, 3).to(weight.dtype), xb, dyb)
        elif stride != 1:   # shapes the kernels above do not take: both gradients through MIOpen
            wb = weight.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            dx, dwb, _ = torch.ops.aten.convolution_backward(dyb, xb, wb, None, [stride, stride], [pad, pad], [1, 1], False,
                                                             [0, 0], 1, [ctx.needs_input_grad[0], ctx.needs_input_grad[1], False])
            dw = None if dwb is None else dwb.to(weight.dtype)
        else:
            if ctx.needs_input_grad[0]:
                # for a 3x3 stride-1 conv the data gradient is the correlation of dY with flip(W)^T at padding 2 - pad:
                # pad=1 -> 1; pad=0 -> 2 is not a kernel mode: pad dY by one ring of zeros and run pad=1.
                if pad == 1:
                    src = dyb
                else:
                    src = torch.nn.functional.pad(dyb, (1, 1, 1, 1)).contiguous(memory_format=torch.channels_last)
                if ctx.bn_src is not None:
                    dx = _tag_bn_bwd(*conv3x3_nhwc(src, pack_weights(weight, transpose_flip=True), None, cout, cin, 1, bn_bwd=ctx.bn_src))
                else:
                    dx = conv3x3_nhwc(src, pack_weights(weight, transpose_flip=True), None, cout, cin, 1)
            if ctx.needs_input_grad[1] and _wgrad_hip(cin, cout):   # off the chain to the next layer: second stream when enabled (side.py)
                if ctx.has_bias and ctx.needs_input_grad[2]:   # the bias gradient rides on the weight-gradient launch
                    def both():
                        dwf, dbf = conv3x3_wgrad(xb, dyb, pad, want_db=True)
                        return dwf.to(weight.dtype), dbf
                    dw, db = _side.run(weight, both, xb, dyb, bias=ctx.bias_p, pair=True)
                else:
                    dw = _side.run(weight, lambda: conv3x3_wgrad(xb, dyb, pad).to(weight.dtype), xb, dyb)
            elif ctx.needs_input_grad[1]:
                wb = weight.detach().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
                _, dwb, _ = torch.ops.aten.convolution_backward(dyb, xb, wb, None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1,
                                                                [False, True, False])
                dw = dwb.to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2] and db is None:   # per-channel sum of dY: the row-reduce kernel's first output half
            lib = _lib.load()
            rows = dyb.shape[0] * dyb.shape[2] * dyb.shape[3]
            stats = torch.empty((2 * cout,), dtype=torch.float32, device=dyb.device)
            ws = _ws(lib.s2d_bnrow_workspace_bytes(rows, cout), dyb.device)
            check(lib.s2d_bnrow_stats_bf16(_ptr(dyb), rows, cout, _ptr(stats), 0, _ptr(ws), ws.numel(), _stream()),
                  "s2d_bnrow_stats_bf16")
            db = stats[:cout]
        return dx, _side.undefer(dw), _side.undefer(db), None, None, None, None


class Conv3x3(nn.Conv2d):
    def _hip_ok(self, x):
        return (ENABLED and x.is_cuda and x.dim() == 4 and torch.is_autocast_enabled()
                and torch.get_autocast_gpu_dtype() == torch.bfloat16
                and self.kernel_size == (3, 3) and self.stride in ((1, 1), (2, 2)) and self.dilation == (1, 1) and self.groups == 1
                and self.padding in ((0, 0), (1, 1)) and self.padding_mode == "zeros"
                and self.in_channels % 64 == 0 and self.out_channels % 64 == 0)

    emit_bn_stats = False   # set by fuse_bn_relu() when a FastBatchNorm2d follows: its statistics come out of our epilogue
    absorbed_pad = 0        # set by fuse_bn_relu(): an nn.ZeroPad2d(1) in front of this padding-0 conv is folded into the kernel

    def forward(self, x):
        if self._hip_ok(x):
            pad = self.padding[0] + self.absorbed_pad
            src = _bn_src_of(x) if torch.is_grad_enabled() else None
            if self.emit_bn_stats and self.training and torch.is_grad_enabled():
                y, partial = _Conv3x3Fn.apply(x, self.weight, self.bias, pad, self.stride[0], True, src)
                y._s2d_bn_partial = partial   # read (forward pass only) by the FastBatchNorm2d that follows
                return y
            return _Conv3x3Fn.apply(x, self.weight, self.bias, pad, self.stride[0], False, src)
        if self.absorbed_pad:
            x = torch.nn.functional.pad(x, (self.absorbed_pad,) * 4)
        return super().forward(x)


# --------------------------------------------------------------------------------------------------
# 1x1 convolutions of the S2D module (csrc/conv2d_nhwc.hip / conv2d_wgrad.hip with one tap)
# --------------------------------------------------------------------------------------------------
def _pack_weights_1x1(weight, transpose, with_dgrad=False):
    if with_dgrad and not transpose and not cached_pack_has(weight, ("conv1x1", False)) and not cached_pack_has(weight, ("conv1x1", True)):
        lib = _lib.load()
        cout, cin = weight.shape[0], weight.shape[1]
        if lib.s2d_conv2d3x3_supported(cin, cout) and lib.s2d_conv2d3x3_supported(cout, cin):
            w = weight.detach().float().reshape(cout, cin).contiguous()
            pf = torch.empty(cin * cout, dtype=torch.bfloat16, device=weight.device)
            pd = torch.empty(cin * cout, dtype=torch.bfloat16, device=weight.device)
            launch = lambda: check(lib.s2d_conv2d1x1_pack_weights_pair_bf16(_ptr(w), cin, cout, _ptr(pf), _ptr(pd), _stream()),
                                   "s2d_conv2d1x1_pack_weights_pair_bf16")
            launch()
            cached_pack_put(weight, ("conv1x1", False), pf)
            cached_pack_put(weight, ("conv1x1", True), pd)
            register_repack(weight, [("conv1x1", False), ("conv1x1", True)], w, launch, conv2d=(w, cin, cout, 1, 0, 0, pf, pd))

    def build():
        lib = _lib.load()
        cout, cin = weight.shape[0], weight.shape[1]
        w = weight.detach().float().reshape(cout, cin).contiguous()
        packed = torch.empty(cin * cout, dtype=torch.bfloat16, device=weight.device)
        pc_in, pc_out = (cout, cin) if transpose else (cin, cout)
        launch = lambda: check(lib.s2d_conv2d1x1_pack_weights_bf16(_ptr(w), pc_in, pc_out, int(transpose), _ptr(packed), _stream()),
                               "s2d_conv2d1x1_pack_weights_bf16")
        launch()
        register_repack(weight, [("conv1x1", bool(transpose))], w, launch, conv2d=(w, pc_in, pc_out, 1, 0, int(transpose), packed, None))
        return packed
    return cached_pack(weight, ("conv1x1", bool(transpose)), build)


def conv1x1_nhwc(x, packed, bias, cin, cout, bn_stats=False, bn_bwd=None):
    """x: bf16 [N,cin,H,W] channels_last -> bf16 [N,cout,H,W] channels_last (+ the per-tile batch-norm statistics slabs); bn_bwd: see
    conv3x3_nhwc"""
    lib = _lib.load()
    assert x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last) and x.shape[1] == cin
    n, _, h, w = x.shape
    y = torch.empty((n, cout, h, w), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    partial = (torch.empty((lib.s2d_conv2d1x1_stats_tiles(n, h, w, cin, cout), 2, cout), dtype=torch.float32, device=x.device)
               if (bn_stats or bn_bwd is not None) else None)
    from . import hip_ops as H
    rec = None
    if H.PROFILE is not None:   # bench.py roofline pass (the one-tap instantiation of the 32-deep tile kernel, or the 64-deep one)
        tiles = lib.s2d_conv2d1x1_stats_tiles(n, h, w, cin, cout)
        rows = next(bm for bm in (128, 96, 64) if -(-n * h * w // bm) == tiles)
        bn = 128 if cout % 128 == 0 else 64
        k32 = cin >= 128 and _os.environ.get("S2D_CONV1X1", "")[:2] != "k6"
        rec = dict(kernel="conv1x1_nhwc_bf16", tag="dense1x1", cin=cin, cout=cout, n_out=n * h * w, kvol=1, pairs=None, dense=True,
                   in_pixels=n * h * w, pad=0, stride=1, tile_rows=rows,
                   kname=(f"conv3x3_k32_nhwc_bf16_kernel<{bn}, {rows // 32}, 1, false>" if k32 else f"conv3x3_nhwc_bf16_kernel<{bn}, 2, 1>"),
                   start=torch.cuda.Event(enable_timing=True), end=torch.cuda.Event(enable_timing=True))
        rec["start"].record()
    if bn_bwd is not None:
        z, fin, act = bn_bwd[0], bn_bwd[1], bn_bwd[2]
        assert bias is None and not bn_stats and z.shape == y.shape and z.dtype == torch.bfloat16 \
            and z.is_contiguous(memory_format=torch.channels_last) and fin.shape == (4, cout)
        fp = fin.data_ptr()
        check(lib.s2d_conv2d1x1_nhwc_bf16_bnbwd(_ptr(x), _ptr(packed), _ptr(_zero_page(x.device)), n, h, w, cin, cout, _ptr(y), _ptr(z),
                                                fp + 8 * cout, fp + 12 * cout, int(act), _ptr(partial), _stream()),
              "s2d_conv2d1x1_nhwc_bf16_bnbwd")
    else:
        check(lib.s2d_conv2d1x1_nhwc_bf16(_ptr(x), _ptr(packed), _ptr(bias), _ptr(_zero_page(x.device)), n, h, w, cin, cout, _ptr(y),
                                          _ptr(partial), _stream()), "s2d_conv2d1x1_nhwc_bf16")
    if rec is not None:
        rec["end"].record()
        H.PROFILE.append(rec)
    return (y, partial) if (bn_stats or bn_bwd is not None) else y


def _pack_matrix_1x1(owner, tag, make, transpose=False):
    """fp32 [out, in] matrix derived from parameter `owner` (a 2x2 kernel seen as a 1x1 conv over 4x the channels) -> the 1x1 kernels'
    weight image, cached per version of the parameter"""
    def build():
        lib = _lib.load()
        m = make().detach().float().contiguous()
        out_c, in_c = m.shape
        packed = torch.empty(out_c * in_c, dtype=torch.bfloat16, device=m.device)
        pc_in, pc_out = (out_c, in_c) if transpose else (in_c, out_c)
        check(lib.s2d_conv2d1x1_pack_weights_bf16(_ptr(m), pc_in, pc_out, int(transpose), _ptr(packed), _stream()),
              "s2d_conv2d1x1_pack_weights_bf16")
        return packed
    return cached_pack(owner, tag, build)


def _space_depth_hip(t):
    return (ENABLED and t.is_cuda and t.dtype == torch.bfloat16 and t.dim() == 4 and t.shape[1] % 8 == 0 and t.numel() > 0
            and t.is_contiguous(memory_format=torch.channels_last) and not (t.requires_grad and torch.is_grad_enabled()))


def _space_to_depth(t):
    """NHWC [n, c, 2h, 2w] -> NHWC [n, 4c, h, w], channel = (py, px, c): the four pixels of a 2x2 block side by side"""
    n, c, hh, ww = t.shape
    if _space_depth_hip(t) and hh % 2 == 0 and ww % 2 == 0:   # one 16-byte-per-lane pass (csrc/layout.hip) instead of torch's strided copy of the view
        y = torch.empty((n, 4 * c, hh // 2, ww // 2), dtype=torch.bfloat16, device=t.device, memory_format=torch.channels_last)
        check(_lib.load().s2d_space_depth2_nhwc_bf16(_ptr(t), n, hh // 2, ww // 2, c, 1, _ptr(y), _stream()), "s2d_space_depth2_nhwc_bf16")
        return y
    v = t.permute(0, 2, 3, 1).reshape(n, hh // 2, 2, ww // 2, 2, c).permute(0, 1, 3, 2, 4, 5).reshape(n, hh // 2, ww // 2, 4 * c)
    return v.permute(0, 3, 1, 2)


def _depth_to_space(t, c):
    """inverse of _space_to_depth: NHWC [n, 4c, h, w] -> NHWC [n, c, 2h, 2w]"""
    n, _, h, w = t.shape
    if _space_depth_hip(t) and t.shape[1] == 4 * c and c % 8 == 0:
        y = torch.empty((n, c, 2 * h, 2 * w), dtype=torch.bfloat16, device=t.device, memory_format=torch.channels_last)
        check(_lib.load().s2d_space_depth2_nhwc_bf16(_ptr(t), n, h, w, c, 0, _ptr(y), _stream()), "s2d_space_depth2_nhwc_bf16")
        return y
    v = t.permute(0, 2, 3, 1).reshape(n, h, w, 2, 2, c).permute(0, 1, 3, 2, 4, 5).reshape(n, 2 * h, 2 * w, c)
    return v.permute(0, 3, 1, 2)


def _wgrad_1x1(xb, dyb, cin, cout):
    """fp32 [cout, cin] = sum over pixels of dy (x) x, both bf16 NHWC"""
    lib = _lib.load()
    n, _, h, w = xb.shape
    dwf = torch.empty((cout, cin), dtype=torch.float32, device=xb.device)
    ws = _ws(lib.s2d_conv2d1x1_wgrad_workspace_bytes(n, h, w, cin, cout), xb.device)
    check(lib.s2d_conv2d1x1_wgrad_nhwc_bf16(_ptr(xb), _ptr(dyb), _ptr(_zero_page(xb.device)), n, h, w, cin, cout, _ptr(dwf), None, _ptr(ws),
                                            ws.numel(), _stream()), "s2d_conv2d1x1_wgrad_nhwc_bf16")
    return dwf


def _channel_sums(dyb):
    """per-channel sum of a bf16 NHWC map (bias gradient): first half of the row-reduce kernel's output"""
    lib = _lib.load()
    n, c, h, w = dyb.shape
    stats = torch.empty((2 * c,), dtype=torch.float32, device=dyb.device)
    ws = _ws(lib.s2d_bnrow_workspace_bytes(n * h * w, c), dyb.device)
    check(lib.s2d_bnrow_stats_bf16(_ptr(dyb), n * h * w, c, _ptr(stats), 0, _ptr(ws), ws.numel(), _stream()), "s2d_bnrow_stats_bf16")
    return stats[:c]


class _Conv1x1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, bn_stats, bn_src=None):
        xb = _nhwc_bf16(x)
        cout, cin = weight.shape[0], weight.shape[1]
        ctx.save_for_backward(xb, weight)
        ctx.bn_src = bn_src if (bn_src is not None and xb is x and _lib.load().s2d_conv2d1x1_bnbwd_supported(cout, cin)) else None
        ctx.has_bias = bias is not None
        ctx.bias_p = bias   # (the parameter itself: side.run hands a deferred bias gradient to it)
        b = None if bias is None else bias.detach().float().contiguous()
        if bn_stats:
            y, partial = conv1x1_nhwc(xb, _pack_weights_1x1(weight, False, with_dgrad=ctx.needs_input_grad[0]), b, cin, cout, bn_stats=True)
            ctx.mark_non_differentiable(partial)
            ctx.set_materialize_grads(False)   # no zero-filled gradient tensor for the statistics output in every backward
            return y, partial
        return conv1x1_nhwc(xb, _pack_weights_1x1(weight, False, with_dgrad=ctx.needs_input_grad[0]), b, cin, cout)

    @staticmethod
    def backward(ctx, dy, *_unused):
        lib = _lib.load()
        xb, weight = ctx.saved_tensors
        cout, cin = weight.shape[0], weight.shape[1]
        dyb = _nhwc_bf16(dy)
        n, _, h, w = xb.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if ctx.bn_src is not None:
                dx = _tag_bn_bwd(*conv1x1_nhwc(dyb, _pack_weights_1x1(weight, True), None, cout, cin, bn_bwd=ctx.bn_src))
            else:
                dx = conv1x1_nhwc(dyb, _pack_weights_1x1(weight, True), None, cout, cin)
        if ctx.needs_input_grad[1]:
            want_db = bool(ctx.has_bias and ctx.needs_input_grad[2])   # the bias gradient rides on the weight-gradient launch

            def wgrad():
                dwf = torch.empty((cout, cin), dtype=torch.float32, device=xb.device)
                dbf = torch.empty((cout,), dtype=torch.float32, device=xb.device) if want_db else None
                ws = _ws(lib.s2d_conv2d1x1_wgrad_workspace_bytes(n, h, w, cin, cout), xb.device)
                check(lib.s2d_conv2d1x1_wgrad_nhwc_bf16(_ptr(xb), _ptr(dyb), _ptr(_zero_page(xb.device)), n, h, w, cin, cout, _ptr(dwf), _ptr(dbf),
                                                        _ptr(ws), ws.numel(), _stream()), "s2d_conv2d1x1_wgrad_nhwc_bf16")
                return dwf.reshape(weight.shape).to(weight.dtype), dbf
            dw, db = _side.run(weight, wgrad, xb, dyb, bias=ctx.bias_p, pair=True)
        if ctx.has_bias and ctx.needs_input_grad[2] and db is None:   # per-channel sum of dY: the row-reduce kernel's first output half
            rows = n * h * w
            stats = torch.empty((2 * cout,), dtype=torch.float32, device=dyb.device)
            ws = _ws(lib.s2d_bnrow_workspace_bytes(rows, cout), dyb.device)
            check(lib.s2d_bnrow_stats_bf16(_ptr(dyb), rows, cout, _ptr(stats), 0, _ptr(ws), ws.numel(), _stream()), "s2d_bnrow_stats_bf16")
            db = stats[:cout]
        return dx, _side.undefer(dw), _side.undefer(db), None, None


class Conv1x1(nn.Conv2d):
    """nn.Conv2d with a 1x1 kernel, stride 1, no padding (same parameters / state_dict keys).  CUDA inputs under bf16 autocast with
    channel counts that are multiples of 64 run the NHWC tile kernels (forward, data and weight gradient); anything else is the
    stock layer."""

    emit_bn_stats = False   # set by fuse_bn_relu() when a FastBatchNorm2d follows

    def _hip_ok(self, x):
        return (ENABLED and x.is_cuda and x.dim() == 4 and torch.is_autocast_enabled()
                and torch.get_autocast_gpu_dtype() == torch.bfloat16
                and self.kernel_size == (1, 1) and self.stride == (1, 1) and self.padding == (0, 0) and self.dilation == (1, 1)
                and self.groups == 1 and self.in_channels % 64 == 0 and self.out_channels % 64 == 0)

    def forward(self, x):
        if self._hip_ok(x):
            src = _bn_src_of(x) if torch.is_grad_enabled() else None
            if self.emit_bn_stats and self.training and torch.is_grad_enabled():
                y, partial = _Conv1x1Fn.apply(x, self.weight, self.bias, True, src)
                y._s2d_bn_partial = partial   # read (forward pass only) by the FastBatchNorm2d that follows
                return y
            return _Conv1x1Fn.apply(x, self.weight, self.bias, False, src)
        return super().forward(x)


class _SmallConv3x3Fn(torch.autograd.Function):
    """3x3 / padding 1 conv with <= 4 output channels from an NHWC bf16 map to fp32 planar predictions (csrc/smallconv.hip)"""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        xb = _nhwc_bf16(x)
        cout, cin = weight.shape[0], weight.shape[1]
        n, _, h, w = xb.shape
        wf = weight.detach().float().contiguous()
        b = None if bias is None else bias.detach().float().contiguous()
        y = torch.empty((n, cout, h, w), dtype=torch.float32, device=xb.device)
        check(lib.s2d_smallconv3x3_fwd(_ptr(xb), _ptr(wf), _ptr(b), n, h, w, cin, cout, _ptr(y), _stream()), "s2d_smallconv3x3_fwd")
        ctx.save_for_backward(xb, weight)
        ctx.has_bias = bias is not None
        ctx.bias_p = bias   # (the parameter itself: side.run hands a deferred bias gradient to it)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        xb, weight = ctx.saved_tensors
        cout, cin = weight.shape[0], weight.shape[1]
        n, _, h, w = xb.shape
        dyf = dy.float().contiguous()
        wf = weight.detach().float().contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((n, cin, h, w), dtype=torch.bfloat16, device=xb.device, memory_format=torch.channels_last)
            check(lib.s2d_smallconv3x3_dgrad(_ptr(dyf), _ptr(wf), n, h, w, cin, cout, _ptr(dx), _stream()), "s2d_smallconv3x3_dgrad")
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            def wgrad():
                dwf = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=xb.device)
                dbf = torch.empty((cout,), dtype=torch.float32, device=xb.device)
                ws = _ws(lib.s2d_smallconv3x3_wgrad_workspace_bytes(cin, cout), xb.device)
                check(lib.s2d_smallconv3x3_wgrad(_ptr(xb), _ptr(dyf), n, h, w, cin, cout, _ptr(dwf), _ptr(dbf), _ptr(ws), ws.numel(), _stream()),
                      "s2d_smallconv3x3_wgrad")
                return dwf.to(weight.dtype), (dbf if ctx.has_bias else None)
            dw, db = _side.run(weight, wgrad, xb, dyf, kind="aux", bias=ctx.bias_p, pair=True)
        return dx, _side.undefer(dw), _side.undefer(db)


class SmallConv3x3(nn.Conv2d):
    """nn.Conv2d(cin, 1..4, 3, padding=1): the last conv of every CenterHead branch (same parameters / state_dict keys).  CUDA
    inputs under bf16 autocast run the streaming kernels and return fp32 NCHW predictions (what the losses and the decoder read);
    anything else is the stock layer."""

    def _hip_ok(self, x):
        return (ENABLED and x.is_cuda and x.dim() == 4 and torch.is_autocast_enabled()
                and torch.get_autocast_gpu_dtype() == torch.bfloat16
                and self.kernel_size == (3, 3) and self.stride == (1, 1) and self.padding == (1, 1) and self.dilation == (1, 1)
                and self.groups == 1 and self.in_channels in (8, 16, 32, 64, 128) and self.out_channels <= 4)

    def forward(self, x):
        if self._hip_ok(x):
            return _SmallConv3x3Fn.apply(x, self.weight, self.bias)
        return super().forward(x)


class _Conv2x2S2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, bn_stats):
        lib = _lib.load()
        xb = _nhwc_bf16(x)
        cout, cin = weight.shape[0], weight.shape[1]
        n, _, h, w = xb.shape

        def build():
            wf = weight.detach().float()
            nhwc = (not wf.is_contiguous()) and wf.is_contiguous(memory_format=torch.channels_last)
            if not nhwc:
                wf = wf.contiguous()
            packed = torch.empty(4 * cin * cout, dtype=torch.bfloat16, device=weight.device)
            launch = lambda: check(lib.s2d_conv2d2x2s2_pack_weights_bf16(_ptr(wf), cin, cout, int(nhwc), _ptr(packed), _stream()),
                                   "s2d_conv2d2x2s2_pack_weights_bf16")
            launch()
            register_repack(weight, [("conv2x2s2",)], wf, launch)
            return packed
        packed = cached_pack(weight, ("conv2x2s2",), build)
        y = torch.empty((n, cout, h // 2, w // 2), dtype=torch.bfloat16, device=xb.device, memory_format=torch.channels_last)
        partial = torch.empty((lib.s2d_conv2d2x2s2_stats_tiles(n, h, w), 2, cout), dtype=torch.float32, device=xb.device) if bn_stats else None
        b = None if bias is None else bias.detach().float().contiguous()
        check(lib.s2d_conv2d2x2s2_nhwc_bf16(_ptr(xb), _ptr(packed), _ptr(b), _ptr(_zero_page(xb.device)), n, h, w, cin, cout, _ptr(y), _ptr(partial),
                                            _stream()), "s2d_conv2d2x2s2_nhwc_bf16")
        ctx.save_for_backward(xb, weight)
        ctx.has_bias = bias is not None
        ctx.bias_p = bias   # (the parameter itself: side.run hands a deferred bias gradient to it)
        if bn_stats:
            ctx.mark_non_differentiable(partial)
            ctx.set_materialize_grads(False)   # no zero-filled gradient tensor for the statistics output in every backward
            return y, partial
        return y

    @staticmethod
    def backward(ctx, dy, *_unused):
        # a 2x2 / stride-2 conv is a 1x1 conv over the space-to-depth image: both gradients run the 1x1 tile kernels
        xb, weight = ctx.saved_tensors
        cout, cin = weight.shape[0], weight.shape[1]
        dyb = _nhwc_bf16(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:   # dy [.., cout] x Wd [(py, px, ci), cout] -> the four pixels of every 2x2 block, then depth-to-space
            packed = _pack_matrix_1x1(weight, ("conv2x2s2", "dgrad"), lambda: weight.permute(2, 3, 1, 0).reshape(4 * cin, cout))
            dx = _depth_to_space(conv1x1_nhwc(dyb, packed, None, cout, 4 * cin), cin)
        if ctx.needs_input_grad[1]:
            dw = _side.run(weight, lambda: _wgrad_1x1(_space_to_depth(xb), dyb, 4 * cin, cout).reshape(cout, 2, 2, cin).permute(0, 3, 1, 2)
                           .to(weight.dtype), xb, dyb, kind="aux")   # _wgrad_1x1: [cout, (py, px, ci)]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _channel_sums(dyb)
        return dx, _side.undefer(dw), _side.undefer(db), None


class Conv2x2S2(nn.Conv2d):
    """nn.Conv2d(cin, cout, 2, stride 2) (same parameters / state_dict keys): the forward (with the batch-norm statistics of a following
    FastBatchNorm2d) runs the NHWC tile kernel with 4 taps; the backward = 1x1 kernels over the space-to-depth image."""

    emit_bn_stats = False

    def _hip_ok(self, x):
        return (ENABLED and x.is_cuda and x.dim() == 4 and torch.is_autocast_enabled()
                and torch.get_autocast_gpu_dtype() == torch.bfloat16
                and self.kernel_size == (2, 2) and self.stride == (2, 2) and self.padding == (0, 0) and self.dilation == (1, 1)
                and self.groups == 1 and self.in_channels % 64 == 0 and self.out_channels % 64 == 0
                and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0)

    def forward(self, x):
        if self._hip_ok(x):
            if self.emit_bn_stats and self.training and torch.is_grad_enabled():
                y, partial = _Conv2x2S2Fn.apply(x, self.weight, self.bias, True)
                y._s2d_bn_partial = partial
                return y
            return _Conv2x2S2Fn.apply(x, self.weight, self.bias, False)
        return super().forward(x)


# --------------------------------------------------------------------------------------------------
# depth-wise 7x7 convolution of the S2D ConvNeXt blocks (csrc/dwconv.hip)
# --------------------------------------------------------------------------------------------------
class _DwConv7Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        xb = _nhwc_bf16(x)
        n, c, h, w = xb.shape
        wf = weight.detach().float().contiguous()
        y = torch.empty_like(xb)
        check(lib.s2d_dwconv7_nhwc_bf16(_ptr(xb), _ptr(wf), _ptr(None if bias is None else bias.detach().float().contiguous()),
                                        n, h, w, c, 0, _ptr(y), _stream()), "s2d_dwconv7_nhwc_bf16")
        ctx.save_for_backward(xb, weight)
        ctx.has_bias = bias is not None
        ctx.bias_p = bias   # (the parameter itself: side.run hands a deferred bias gradient to it)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        xb, weight = ctx.saved_tensors
        n, c, h, w = xb.shape
        dyb = _nhwc_bf16(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(xb)
            check(lib.s2d_dwconv7_nhwc_bf16(_ptr(dyb), _ptr(weight.detach().float().contiguous()), None, n, h, w, c, 1, _ptr(dx),
                                            _stream()), "s2d_dwconv7_nhwc_bf16 (data gradient)")
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            def wgrad():
                dwf = torch.empty((c, 1, 7, 7), dtype=torch.float32, device=xb.device)
                dbf = torch.empty((c,), dtype=torch.float32, device=xb.device) if ctx.has_bias else None
                ws = _ws(lib.s2d_dwconv7_wgrad_workspace_bytes(n, h, w, c), xb.device)
                check(lib.s2d_dwconv7_wgrad_nhwc_bf16(_ptr(xb), _ptr(dyb), n, h, w, c, _ptr(dwf), _ptr(dbf), _ptr(ws), ws.numel(),
                                                      _stream()), "s2d_dwconv7_wgrad_nhwc_bf16")
                return dwf.to(weight.dtype), dbf
            dw, db = _side.run(weight, wgrad, xb, dyb, kind="aux", bias=ctx.bias_p, pair=True)
        return dx, _side.undefer(dw), _side.undefer(db)


class DepthwiseConv7(nn.Conv2d):
    """nn.Conv2d(C, C, 7, padding=3, groups=C) (same parameters / state_dict keys).  CUDA inputs under bf16 autocast run
    the NHWC stencil kernels; anything else is the stock layer."""

    def _hip_ok(self, x):
        return (ENABLED and x.is_cuda and x.dim() == 4 and torch.is_autocast_enabled()
                and torch.get_autocast_gpu_dtype() == torch.bfloat16
                and self.kernel_size == (7, 7) and self.stride == (1, 1) and self.padding == (3, 3) and self.dilation == (1, 1)
                and self.groups == self.in_channels == self.out_channels and self.padding_mode == "zeros"
                and self.in_channels % 8 == 0 and self.in_channels <= 640)

    def forward(self, x):
        if self._hip_ok(x):
            return _DwConv7Fn.apply(x, self.weight, self.bias)
        return super().forward(x)


class AbsorbedZeroPad2d(nn.ZeroPad2d):
    """nn.ZeroPad2d(1) whose zero border is produced inside the 3x3 conv that follows it (ZeroPad2d(1) + conv(padding=0)
    == conv(padding=1), rpn.py:129-131): keeps the reference's Sequential slot, moves no data."""

    def forward(self, x):
        return x


# --------------------------------------------------------------------------------------------------
# BatchNorm2d (+ReLU) on NHWC bf16 (csrc/features.hip, s2d_bnrow_*)
# --------------------------------------------------------------------------------------------------
_ws_cache = {}
_WS_POISON = None if not _os.environ.get("S2D_WS_POISON") else float(_os.environ["S2D_WS_POISON"])
WS_PRIVATE = False   # set by graphed.GraphedSegment while it captures: a HIP graph must not reference a buffer that a later, larger request re-allocates


def _ws(nbytes, device):
    """reduction workspace: one grow-only buffer per device and stream.  Every user writes it before reading it inside
    one entry point, and launches on a stream are ordered, so consecutive calls can share it."""
    if WS_PRIVATE:   # graph capture: a plain allocation from the graph's memory pool, owned by the graph
        return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
    key = (device.index, _stream())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    if _WS_POISON is not None:   # debugging aid (S2D_WS_POISON=<float>): a kernel that reads workspace words it did not write shows up in the results
        buf.view(torch.float32).fill_(_WS_POISON)
    return buf


_dist_sync = _collective.sync_on


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def channel_slice_ld(t, c):
    """row stride (elements) if t is a [N, c, H, W] bf16 channel slice of a WIDER channels_last tensor whose rows the *_ld kernels can
    address (16-byte aligned start, stride a multiple of 8), else 0"""
    if not (torch.is_tensor(t) and t.dim() == 4 and t.dtype == torch.bfloat16 and t.is_cuda and t.shape[1] == c):
        return 0
    n, _, h, w = t.shape
    s = t.stride()
    ld = s[3]
    if s[1] != 1 or ld <= c or ld % 8 or s[2] != w * ld or (n > 1 and s[0] != h * w * ld) or (t.storage_offset() * 2) % 16:
        return 0
    return ld


class _CatSlicesFn(torch.autograd.Function):
    """torch.cat(parts, 1) whose parts were WRITTEN as channel slices of `buf` by their producers (FastBatchNorm2d(out=slice)): the forward
    hands out the filled buffer, the backward hands every producer its channel slice of the gradient as a strided view - no copy
    either way (the *_ld batch-norm kernels read it in place)."""

    @staticmethod
    def forward(ctx, buf, *parts):
        ctx.widths = [p.shape[1] for p in parts]
        return buf.detach()

    @staticmethod
    def backward(ctx, dcat):
        if dcat.dtype != torch.bfloat16 or not dcat.is_contiguous(memory_format=torch.channels_last):
            dcat = _nhwc_bf16(dcat)
        outs, off = [], 0
        for wd in ctx.widths:
            outs.append(dcat[:, off:off + wd])
            off += wd
        return (None, *outs)


class _BNRowFn(torch.autograd.Function):
    """Batch norm over the rows of a row-major bf16 matrix: x is bf16 channels_last [N,C,H,W] (rows = N*H*W) or a
    sparse feature matrix [n,C].  y = relu?(bn(x) + residual?).  Training statistics over all rows (of all ranks when
    `sync`).  This runs ~80 times per training step: per-channel vectors live in one [4,C] / [5,C] buffer each and are
    passed as base pointer + offset (no tensor views on the hot path)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, relu, eps, sync, module, training, partial=None, out=None):
        """out (r04): a channel slice [N, c, H, W] of a wider bf16 channels_last tensor to write y into (`channel_slice_ld`) - the RPN's
        up-sampling branches fill the concatenated tensor directly; returned as the function's output"""
        lib = _lib.load()
        c = x.shape[1]
        rows = x.numel() // c
        dev = x.device
        gamma, beta = _f32c(gamma), _f32c(beta)
        track = module.track_running_stats
        ws = _ws(lib.s2d_bnrow_workspace_bytes(rows, c), dev)
        stream = _stream()
        count = None
        rm, rv, nbt = (module.running_mean, module.running_var, module.num_batches_tracked) if track else (None, None, None)
        mom = float(module.momentum if track else 0.0)
        if training:
            fin = torch.empty((4, c), dtype=torch.float32, device=dev)   # rows: mean, invstd, scale, shift
            fp, rb = fin.data_ptr(), 4 * c
            if track:   # the finalize kernels below update the running statistics through raw pointers (no version bump): see the eval cache
                module._s2d_stats_epoch = getattr(module, "_s2d_stats_epoch", 0) + 1
            if partial is not None and not sync:   # statistics pass already done in the producing conv's epilogue
                pws = ws
                if lib.s2d_bn_partials_sum_workspace_bytes(partial.shape[0], c) > ws.numel():   # long list (sparse conv tiles): two stages
                    pws = _ws(lib.s2d_bn_partials_sum_workspace_bytes(partial.shape[0], c), dev)
                check(lib.s2d_bn_partials_finalize_ws_f32(partial.data_ptr(), partial.shape[0], rows, c, gamma.data_ptr(),
                                                          beta.data_ptr(), float(eps), mom, fp, fp + rb, fp + 2 * rb, fp + 3 * rb,
                                                          _ptr(rm), _ptr(rv), _ptr(nbt), pws.data_ptr(), pws.numel(), stream),
                      "s2d_bn_partials_finalize_ws_f32")
            elif not sync:
                check(lib.s2d_bnrow_stats_finalize_bf16(x.data_ptr(), rows, c, gamma.data_ptr(), beta.data_ptr(), float(eps), mom,
                                                        fp, fp + rb, fp + 2 * rb, fp + 3 * rb, _ptr(rm), _ptr(rv), _ptr(nbt),
                                                        ws.data_ptr(), ws.numel(), stream), "s2d_bnrow_stats_finalize_bf16")
            else:
                import torch.distributed as dist
                packed = torch.empty((2 * c + 1,), dtype=torch.float32, device=dev)   # [sum, sumsq, row count]
                if partial is not None:
                    pws = _ws(max(lib.s2d_bn_partials_sum_workspace_bytes(partial.shape[0], c), 256), dev)
                    check(lib.s2d_bn_partials_sum_ws_f32(partial.data_ptr(), partial.shape[0], rows, c, packed.data_ptr(), 1, pws.data_ptr(),
                                                         pws.numel(), stream), "s2d_bn_partials_sum_ws_f32")
                else:
                    check(lib.s2d_bnrow_stats_bf16(x.data_ptr(), rows, c, packed.data_ptr(), 1, ws.data_ptr(), ws.numel(), stream),
                          "s2d_bnrow_stats_bf16")
                _collective.allreduce_sum_(packed)
                count = packed   # kept alive for the backward: the count is its last element
                pp = packed.data_ptr()
                check(lib.s2d_bn1d_finalize_fwd_f32(pp, pp + 8 * c, gamma.data_ptr(), beta.data_ptr(), float(eps), mom, c, fp,
                                                    fp + rb, fp + 2 * rb, fp + 3 * rb, _ptr(rm), _ptr(rv), _ptr(nbt), stream),
                      "s2d_bn1d_finalize_fwd_f32")
        else:
            # running statistics: [mean, invstd, scale, shift].  Composed from six tiny torch launches per layer - for a FROZEN network
            # (the distillation teacher: 62 such layers, ~370 launches per step) the result is cached on the module, keyed on every input's
            # storage and version and on the count of raw-pointer statistics updates a training-mode forward of this layer has made
            frozen = not module.weight.requires_grad and not module.bias.requires_grad   # (the fused Adam moves trainable ones through raw pointers)
            rm_, rv_ = module.running_mean, module.running_var
            key = (gamma.data_ptr(), gamma._version, beta.data_ptr(), beta._version, rm_.data_ptr(), rm_._version, rv_.data_ptr(), rv_._version,
                   float(eps), getattr(module, "_s2d_stats_epoch", 0)) if frozen else None
            hit = getattr(module, "_s2d_eval_fin", None) if frozen else None
            if hit is not None and hit[0] == key:
                fin = hit[1]
            else:
                invstd = torch.rsqrt(rv_.float() + eps)
                scale = gamma * invstd
                fin = torch.stack([rm_.float(), invstd, scale, beta - rm_.float() * scale])
                if frozen:
                    module._s2d_eval_fin = (key, fin)
            fp, rb = fin.data_ptr(), 4 * c
        if out is not None:
            y, y_ld = out, channel_slice_ld(out, c)
            assert y_ld and out.shape == x.shape and not out.requires_grad, "out must be a channel slice of a bf16 channels_last tensor, shaped like x"
        else:
            y, y_ld = torch.empty_like(x), c   # preserves channels_last
        check(lib.s2d_bnrow_apply_ld_bf16(x.data_ptr(), fp + 2 * rb, fp + 3 * rb, _ptr(residual), int(relu), rows, c, y.data_ptr(), y_ld,
                                          stream), "s2d_bnrow_apply_ld_bf16")
        has_res = residual is not None
        assert not (has_res and int(relu) == 2), "fused GELU behind a residual add is not supported (its derivative needs bn(x)+res)"
        # with a residual the ReLU mask cannot be recomputed from x alone: keep y
        assert out is None or not (has_res and relu), "a ReLU behind a residual keeps y for its backward: not with a strided destination"
        ctx.save_for_backward(x, gamma, fin, count, y if (has_res and relu) else None)
        ctx.relu, ctx.sync, ctx.training, ctx.has_res = relu, sync, training, has_res
        if training and not sync and not has_res and out is None and x.dim() == 4 and BN_BWD_FOLD:
            # read by the conv that consumes y (Conv3x3 / Conv1x1 forward): its data gradient is this layer's dY and can emit our backward sums
            y._s2d_bn_src = (x, fin, int(relu), y._version)
        # (a destination slice is a plain no-grad buffer written in place; the output is a fresh alias of it, so autograd neither sees an
        # in-place operation on a view - which would put a CopySlices with its copies into the backward - nor an input returned as is)
        return y if out is None else y.detach()

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, gamma, fin, count, y = ctx.saved_tensors
        c = x.shape[1]
        rows = x.numel() // c
        dev = x.device
        fp, rb = fin.data_ptr(), 4 * c          # mean, invstd, scale, shift
        dy_ld = channel_slice_ld(dy, c) if x.dim() == 4 else 0   # a channel slice of the concatenated tensor's gradient is read in place
        if not dy_ld:
            dy_ld = c
            if dy.dtype != torch.bfloat16 or not (dy.is_contiguous(memory_format=torch.channels_last) if x.dim() == 4
                                                  else dy.is_contiguous()):
                dy = _nhwc_bf16(dy) if x.dim() == 4 else dy.to(torch.bfloat16).contiguous()
        relu = int(ctx.relu)
        ws = _ws(lib.s2d_bnrow_workspace_bytes(rows, c), dev)
        stream = _stream()
        out = torch.empty((5, c), dtype=torch.float32, device=dev)   # rows: dgamma, dbeta, a, b, d
        op = out.data_ptr()
        tag = getattr(dy, "_s2d_bnbwd", None) if (ctx.training and not ctx.sync and not ctx.has_res and dy_ld == c and x.dim() == 4) else None
        if tag is not None and (tag[1] != dy._version or tag[0].shape[2] != c):
            tag = None
        if tag is not None:   # the conv that produced dy summed (g, g x) per tile in its epilogue: fold the slabs, no pass over (dy, x)
            part = tag[0]
            pws = _ws(max(lib.s2d_bn_partials_sum_workspace_bytes(part.shape[0], c), 256), dev)
            check(lib.s2d_bn_partials_bwd_finalize_ws_f32(part.data_ptr(), part.shape[0], rows, c, gamma.data_ptr(), fp, fp + rb, op, op + rb,
                                                          op + 2 * rb, op + 3 * rb, op + 4 * rb, pws.data_ptr(), pws.numel(), stream),
                  "s2d_bn_partials_bwd_finalize_ws_f32")
            STATS["bn_bwd_folded"] += 1
        elif ctx.training and not ctx.sync:
            STATS["bn_bwd_reduced"] += 1
            check(lib.s2d_bnrow_bwd_reduce_finalize_ld_bf16(dy.data_ptr(), dy_ld, x.data_ptr(), _ptr(y), fp + 2 * rb, fp + 3 * rb, relu, rows,
                                                            c, gamma.data_ptr(), fp, fp + rb, op, op + rb, op + 2 * rb, op + 3 * rb,
                                                            op + 4 * rb, ws.data_ptr(), ws.numel(), stream),
                  "s2d_bnrow_bwd_reduce_finalize_ld_bf16")
        else:
            sums = torch.empty((2, 2 * c), dtype=torch.float32, device=dev)   # row 0: local sums, row 1: the copy that is all-reduced
            sp = sums.data_ptr()
            check(lib.s2d_bnrow_bwd_reduce_ld_bf16(dy.data_ptr(), dy_ld, x.data_ptr(), _ptr(y), fp + 2 * rb, fp + 3 * rb, relu, rows, c, sp,
                                                   sp + 8 * c if ctx.training else None, ws.data_ptr(), ws.numel(), stream),
                  "s2d_bnrow_bwd_reduce_ld_bf16")
            if ctx.training:
                import torch.distributed as dist
                _collective.allreduce_sum_(sums[1])
                check(lib.s2d_bn1d_finalize_bwd_f32(sp, sp + 8 * c, count.data_ptr() + 8 * c, gamma.data_ptr(), fp, fp + rb, c, op,
                                                    op + rb, op + 2 * rb, op + 3 * rb, op + 4 * rb, stream),
                      "s2d_bn1d_finalize_bwd_f32")
            else:   # eval: y = x*scale + shift with constant scale
                mean, invstd, scale = fin[0], fin[1], fin[2]
                zeros = torch.zeros_like(scale)
                out = torch.stack([invstd * (sums[0, c:] - mean * sums[0, :c]), sums[0, :c], scale, zeros, zeros])
                op = out.data_ptr()
        dx = dres = None
        want_res = ctx.has_res and ctx.needs_input_grad[3]
        if ctx.needs_input_grad[0] or want_res:
            dx = torch.empty_like(x)
            if want_res:
                dres = torch.empty_like(x) if relu else dy   # without a ReLU the residual gradient is dy itself
            check(lib.s2d_bnrow_bwd_apply_ld_bf16(dy.data_ptr(), dy_ld, x.data_ptr(), _ptr(y), fp + 2 * rb, fp + 3 * rb, relu, op + 2 * rb,
                                                  op + 3 * rb, op + 4 * rb, rows, c, dx.data_ptr(),
                                                  dres.data_ptr() if (want_res and relu) else None, stream), "s2d_bnrow_bwd_apply_ld_bf16")
        return dx, out[0], out[1], dres, None, None, None, None, None, None, None


class FastBatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d (same parameters / buffers / state_dict keys) with an optional fused ReLU.  bf16 channels_last
    CUDA inputs run the row-major HIP kernels, which synchronise their statistics across ranks themselves when
    torch.distributed is initialised (so dp.convert_syncbn leaves this class alone).  Other inputs take the stock
    batch norm — torch's SyncBatchNorm function when a process group is up — followed by the ReLU."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, fused_relu=False):
        super().__init__(num_features, eps, momentum, affine, track_running_stats)
        self.fused_relu = fused_relu

    def _hip_ok(self, x):
        return (ENABLED and x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and self.affine
                and self.num_features % 8 == 0 and self.num_features <= 1024 and x.numel() > 0
                and x.is_contiguous(memory_format=torch.channels_last) and self.momentum is not None)

    def forward(self, x, relu=None, out=None):
        """out: see _BNRowFn.forward (only on the HIP path: callers check `_hip_ok` first)"""
        relu = self.fused_relu if relu is None else relu
        training = self.training or not self.track_running_stats
        sync = training and _dist_sync()
        if self._hip_ok(x):
            partial = getattr(x, "_s2d_bn_partial", None) if training else None
            if partial is not None and partial.shape[2] != self.num_features:
                partial = None
            return _BNRowFn.apply(x, self.weight, self.bias, None, relu, self.eps, sync, self, training, partial, out)
        assert out is None, "FastBatchNorm2d(out=...) needs the HIP path"
        if sync and x.is_cuda:
            import torch.distributed as dist
            from torch.nn.modules._functions import SyncBatchNorm as _SyncFn
            if self.track_running_stats:
                self.num_batches_tracked += 1
            y = _SyncFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps, self.momentum,
                              dist.group.WORLD, dist.get_world_size())
        else:
            y = super().forward(x)
        if int(relu) == 2:
            return torch.nn.functional.gelu(y)
        return torch.relu(y) if relu else y


class _ConvT2x2S2Fn(torch.autograd.Function):
    """ConvTranspose2d(k=2, s=2): a 1x1 conv to 4x the channels (py, px, co) followed by depth-to-space"""

    @staticmethod
    def forward(ctx, x, weight, bias):
        xb = _nhwc_bf16(x)
        cin, cout = weight.shape[0], weight.shape[1]
        packed = _pack_matrix_1x1(weight, ("convt2x2s2", "fwd"), lambda: weight.permute(2, 3, 1, 0).reshape(4 * cout, cin))
        b4 = None if bias is None else bias.detach().float().repeat(4).contiguous()
        y = _depth_to_space(conv1x1_nhwc(xb, packed, b4, cin, 4 * cout), cout)
        ctx.save_for_backward(xb, weight)
        ctx.has_bias = bias is not None
        ctx.bias_p = bias   # (the parameter itself: side.run hands a deferred bias gradient to it)
        return y

    @staticmethod
    def backward(ctx, dy):
        xb, weight = ctx.saved_tensors
        cin, cout = weight.shape[0], weight.shape[1]
        dyb = _nhwc_bf16(dy)
        dys = _space_to_depth(dyb)   # [n, (py, px, co), h, w]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            packed = _pack_matrix_1x1(weight, ("convt2x2s2", "dgrad"), lambda: weight.permute(2, 3, 1, 0).reshape(4 * cout, cin), transpose=True)
            dx = conv1x1_nhwc(dys, packed, None, 4 * cout, cin)
        if ctx.needs_input_grad[1]:   # (the permuted view is re-laid to the parameter's layout on the side stream: side.run)
            dw = _side.run(weight, lambda: _wgrad_1x1(xb, dys, cin, 4 * cout).reshape(2, 2, cout, cin).permute(3, 2, 0, 1).to(weight.dtype),
                           xb, dys, kind="aux")   # _wgrad_1x1: [(py, px, co), ci]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = _channel_sums(dyb)
        return dx, _side.undefer(dw), _side.undefer(db)


class ConvT2x2S2(nn.ConvTranspose2d):
    """nn.ConvTranspose2d(cin, cout, 2, stride 2) (same parameters / state_dict keys): the RPN's up-sampling deblock
    (/root/reference/det3d/models/necks/rpn.py:84-113).  CUDA inputs under bf16 autocast with channel counts that are multiples of 64
    run as a 1x1 conv on the NHWC tile kernels plus a depth-to-space pass; anything else is the stock layer."""

    def _hip_ok(self, x):
        return (ENABLED and x.is_cuda and x.dim() == 4 and torch.is_autocast_enabled()
                and torch.get_autocast_gpu_dtype() == torch.bfloat16
                and self.kernel_size == (2, 2) and self.stride == (2, 2) and self.padding == (0, 0) and self.output_padding == (0, 0)
                and self.dilation == (1, 1) and self.groups == 1 and self.in_channels % 64 == 0 and self.out_channels % 64 == 0)

    def forward(self, x, output_size=None):
        if output_size is None and self._hip_ok(x):
            return _ConvT2x2S2Fn.apply(x, self.weight, self.bias)
        return super().forward(x, output_size)


# --------------------------------------------------------------------------------------------------
# stride-2 transposed forms: ConvTranspose2d(4,2,1) (decoder_1 / decoder_2 of the S2D module, rpn.py:217-231) and the backward of the
# stride-2 3x3 convs (rpn.py:126-133) on the tile kernels (csrc/conv2d_nhwc.hip "UP" mode, conv2d_wgrad.hip STRIDE = 2)
# --------------------------------------------------------------------------------------------------
def _s2_backward_ok(xb, dyb, cin, cout, pad):
    lib = _lib.load()
    return bool(pad == 1 and xb.shape[2] == 2 * dyb.shape[2] and xb.shape[3] == 2 * dyb.shape[3]
                and lib.s2d_convup_supported(cout, cin, 3) and lib.s2d_conv2d_s2_wgrad_supported(cout, cin, 3))


def conv_up(x, weight, bias, ks, bn_stats=False):
    """x bf16 NHWC [n, kc, h, w], weight fp32 [kc, nc, ks, ks] -> bf16 NHWC [n, nc, 2h, 2w]: ks = 4 the forward of ConvTranspose2d(4,2,1),
    ks = 3 the data gradient of a stride-2 / pad-1 3x3 conv (x = dY, weight = the conv's [cout, cin, 3, 3])."""
    lib = _lib.load()
    kc, nc = weight.shape[0], weight.shape[1]
    assert x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last) and x.shape[1] == kc and weight.shape[2] == ks
    n, _, h, w = x.shape

    def build():
        packed = torch.empty(ks * ks * kc * nc, dtype=torch.bfloat16, device=weight.device)
        wsrc = weight.detach().float().contiguous()
        launch = lambda: check(lib.s2d_convup_pack_weights_bf16(_ptr(wsrc), kc, nc, ks, _ptr(packed), _stream()), "s2d_convup_pack_weights_bf16")
        launch()
        register_repack(weight, [("convup", ks)], wsrc, launch)
        return packed
    packed = cached_pack(weight, ("convup", ks), build)
    y = torch.empty((n, nc, 2 * h, 2 * w), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    partial = (torch.empty((lib.s2d_convup_stats_tiles(n, h, w, kc, nc, ks), 2, nc), dtype=torch.float32, device=x.device)
               if bn_stats else None)
    from . import hip_ops as H
    rec = None
    if H.PROFILE is not None:
        tiles = lib.s2d_convup_stats_tiles(n, h, w, kc, nc, ks) // 4
        rows = next(bm for bm in (128, 96, 64) if -(-n * h * w // bm) == tiles)
        rec = dict(kernel="conv_up_nhwc_bf16", tag="dense_up", cin=kc, cout=nc, n_out=4 * n * h * w, kvol=ks * ks / 4.0, pairs=None, dense=True,
                   in_pixels=n * h * w, pad=1, stride=2, tile_rows=rows,
                   kname=f"conv3x3_k32_nhwc_bf16_kernel<{128 if nc % 128 == 0 else 64}, {rows // 32}, {ks}, true>",
                   start=torch.cuda.Event(enable_timing=True), end=torch.cuda.Event(enable_timing=True))
        rec["start"].record()
    check(lib.s2d_convup_nhwc_bf16(_ptr(x), _ptr(packed), _ptr(bias), _ptr(_zero_page(x.device)), n, h, w, kc, nc, ks, _ptr(y), _ptr(partial),
                                   _stream()), "s2d_convup_nhwc_bf16")
    if rec is not None:
        rec["end"].record()
        H.PROFILE.append(rec)
    return (y, partial) if bn_stats else y


def conv4x4s2(x, weight):
    """x bf16 NHWC [n, cin, h, w], weight fp32 [cout, cin, 4, 4] -> bf16 NHWC [n, cout, h/2, w/2] (stride 2, pad 1): the data gradient of
    ConvTranspose2d(cout -> cin, 4, 2, 1), whose stored weight [its Cin, its Cout, 4, 4] is exactly that array."""
    lib = _lib.load()
    cout, cin = weight.shape[0], weight.shape[1]
    assert x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last) and x.shape[1] == cin
    n, _, h, w = x.shape

    def build():
        packed = torch.empty(16 * cin * cout, dtype=torch.bfloat16, device=weight.device)
        wsrc = weight.detach().float().contiguous()
        launch = lambda: check(lib.s2d_conv2d4x4s2_pack_weights_bf16(_ptr(wsrc), cin, cout, _ptr(packed), _stream()), "s2d_conv2d4x4s2_pack_weights_bf16")
        launch()
        register_repack(weight, [("conv4x4s2",)], wsrc, launch)
        return packed
    packed = cached_pack(weight, ("conv4x4s2",), build)
    y = torch.empty((n, cout, h // 2, w // 2), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    check(lib.s2d_conv2d4x4s2_nhwc_bf16(_ptr(x), _ptr(packed), _ptr(_zero_page(x.device)), n, h, w, cin, cout, _ptr(y), _stream()),
          "s2d_conv2d4x4s2_nhwc_bf16")
    return y


def conv_s2_wgrad(a, b, ks):
    """a bf16 NHWC [n, ca, h/2, w/2], b bf16 NHWC [n, cb, h, w] -> fp32 [ca, cb, ks, ks] = sum a[i, j] * b[2i - 1 + ky, 2j - 1 + kx]"""
    lib = _lib.load()
    n, ca = a.shape[0], a.shape[1]
    cb, h, w = b.shape[1], b.shape[2], b.shape[3]
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.shape[2] * 2 == h and a.shape[3] * 2 == w
    assert a.is_contiguous(memory_format=torch.channels_last) and b.is_contiguous(memory_format=torch.channels_last)
    dw = torch.empty((ca, cb, ks, ks), dtype=torch.float32, device=a.device)
    ws = _ws(lib.s2d_conv2d_s2_wgrad_workspace_bytes(n, h, w, ca, cb, ks), a.device)
    check(lib.s2d_conv2d_s2_wgrad_nhwc_bf16(_ptr(a), _ptr(b), _ptr(_zero_page(a.device)), n, h, w, ca, cb, ks, _ptr(dw), _ptr(ws), ws.numel(),
                                            _stream()), "s2d_conv2d_s2_wgrad_nhwc_bf16")
    return dw


class _ConvT4x4S2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, bn_stats):
        xb = _nhwc_bf16(x)
        ctx.save_for_backward(xb, weight)
        ctx.has_bias = bias is not None
        ctx.bias_p = bias   # (the parameter itself: side.run hands a deferred bias gradient to it)
        b = None if bias is None else bias.detach().float().contiguous()
        if bn_stats:
            y, partial = conv_up(xb, weight, b, 4, bn_stats=True)
            ctx.mark_non_differentiable(partial)
            ctx.set_materialize_grads(False)
            return y, partial
        return conv_up(xb, weight, b, 4)

    @staticmethod
    def backward(ctx, dy, *_unused):
        xb, weight = ctx.saved_tensors
        dyb = _nhwc_bf16(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = conv4x4s2(dyb, weight)
        want_db = bool(ctx.has_bias and ctx.needs_input_grad[2])
        if ctx.needs_input_grad[1]:   # weight and bias gradient: off the chain (side.py)
            dw, db = _side.run(weight, lambda: (conv_s2_wgrad(xb, dyb, 4).to(weight.dtype), _channel_sums(dyb) if want_db else None), xb, dyb,
                               kind="aux", bias=ctx.bias_p, pair=True)
        elif want_db:
            db = _channel_sums(dyb)
        return dx, _side.undefer(dw), _side.undefer(db), None


class ConvT4x4S2(nn.ConvTranspose2d):
    """nn.ConvTranspose2d(cin, cout, 4, 2, 1) (same parameters / state_dict keys): decoder_1 / decoder_2 of the S2D module
    (/root/reference/det3d/models/necks/rpn.py:217-231; 64 channels in the pillar S2D module, readers/pillar_encoder.py:337-394).  CUDA inputs
    under bf16 autocast with channel counts that are multiples of 64:
    forward = four parity-class 2x2 convs in one launch (with the batch-norm statistics of a following FastBatchNorm2d), data
    gradient = a 4x4 stride-2 conv, weight gradient = the stride-2 contraction; anything else is the stock layer."""

    emit_bn_stats = False

    def _hip_ok(self, x):
        if not (ENABLED and x.is_cuda and x.dim() == 4 and torch.is_autocast_enabled()
                and torch.get_autocast_gpu_dtype() == torch.bfloat16
                and self.kernel_size == (4, 4) and self.stride == (2, 2) and self.padding == (1, 1) and self.output_padding == (0, 0)
                and self.dilation == (1, 1) and self.groups == 1):
            return False
        lib = _lib.load()
        cin, cout = self.in_channels, self.out_channels
        return bool(lib.s2d_convup_supported(cin, cout, 4) and lib.s2d_conv2d_s2_wgrad_supported(cin, cout, 4)
                    and lib.s2d_conv2d3x3_supported(cout, cin))

    def forward(self, x, output_size=None):
        if output_size is None and self._hip_ok(x):
            if self.emit_bn_stats and self.training and torch.is_grad_enabled():
                y, partial = _ConvT4x4S2Fn.apply(x, self.weight, self.bias, True)
                y._s2d_bn_partial = partial
                return y
            return _ConvT4x4S2Fn.apply(x, self.weight, self.bias, False)
        return super().forward(x, output_size)


def fuse_bn_relu(layers):
    """[.., FastBatchNorm2d, nn.ReLU, ..] -> [.., FastBatchNorm2d(fused_relu), nn.Identity, ..]: same indices (state_dict
    keys of the reference checkpoints), one kernel instead of two."""
    layers = list(layers)
    for i in range(len(layers) - 1):
        if isinstance(layers[i], FastBatchNorm2d) and isinstance(layers[i + 1], nn.ReLU):
            layers[i].fused_relu = True
            layers[i + 1] = nn.Identity()
        if isinstance(layers[i], FastBatchNorm2d) and isinstance(layers[i + 1], nn.GELU) and layers[i + 1].approximate == "none":
            layers[i].fused_relu = 2   # activation code 2: exact GELU behind the normalisation (csrc/features.hip)
            layers[i + 1] = nn.Identity()
        if isinstance(layers[i], (Conv3x3, Conv1x1, Conv2x2S2, ConvT4x4S2)) and isinstance(layers[i + 1], FastBatchNorm2d):
            layers[i].emit_bn_stats = True   # the conv epilogue produces the batch-norm statistics partials
        if type(layers[i]) is nn.ZeroPad2d and tuple(layers[i].padding) == (1, 1, 1, 1) and isinstance(layers[i + 1], Conv3x3) \
                and layers[i + 1].padding == (0, 0) and layers[i + 1].kernel_size == (3, 3):
            layers[i] = AbsorbedZeroPad2d(1)
            layers[i + 1].absorbed_pad = 1
    return layers


# --------------------------------------------------------------------------------------------------
# LayerNorm over a whole [C,H,W] map (ConvNeXt blocks of the S2D module, rpn.py:186-259)
# --------------------------------------------------------------------------------------------------
class _WideLNFn(torch.autograd.Function):
    """LayerNorm over whole [C,H,W] maps of a bf16 batch (csrc/layernorm.hip): everything runs in the memory order of x; weight
    and bias are handed over permuted to that order (cached per parameter version) and their gradients permuted back."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        lib = _lib.load()
        b = x.shape[0]
        row = x.numel() // b
        nhwc = x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
        to_mem = (lambda t: t.detach().float().permute(1, 2, 0).contiguous()) if nhwc else (lambda t: t.detach().float().contiguous())
        w_t = cached_pack(weight, ("lnwide", nhwc), lambda: to_mem(weight))
        b_t = cached_pack(bias, ("lnwide", nhwc), lambda: to_mem(bias))
        y = torch.empty_like(x)   # same strides
        stats = torch.empty((b, 2), dtype=torch.float32, device=x.device)
        ws = _ws(lib.s2d_lnwide_workspace_bytes(b), x.device)
        check(lib.s2d_lnwide_fwd_bf16(x.data_ptr(), _ptr(w_t), _ptr(b_t), b, row, float(eps), y.data_ptr(), _ptr(stats), _ptr(ws), ws.numel(),
                                      _stream()), "s2d_lnwide_fwd_bf16")
        ctx.save_for_backward(x, w_t, stats)
        ctx.nhwc, ctx.pshape = nhwc, weight.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w_t, stats = ctx.saved_tensors
        b = x.shape[0]
        row = x.numel() // b
        if dy.stride() != x.stride():
            dy = torch.empty_like(x).copy_(dy)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty(row, dtype=torch.float32, device=x.device) if ctx.needs_input_grad[1] else None
        db = torch.empty(row, dtype=torch.float32, device=x.device) if ctx.needs_input_grad[2] else None
        ws = _ws(lib.s2d_lnwide_workspace_bytes(b), x.device)
        check(lib.s2d_lnwide_bwd_bf16(dy.data_ptr(), x.data_ptr(), _ptr(w_t), _ptr(stats), b, row, _ptr(dx), _ptr(dw), _ptr(db), _ptr(ws),
                                      ws.numel(), _stream()), "s2d_lnwide_bwd_bf16")

        def back(t):
            if t is None:
                return None
            if ctx.nhwc:
                c, h, w = ctx.pshape
                return t.view(h, w, c).permute(2, 0, 1).contiguous()
            return t.view(ctx.pshape)
        return dx, back(dw), back(db), None


class WideLayerNorm(nn.LayerNorm):
    """nn.LayerNorm (same parameters / state_dict keys).  The S2D blocks normalise over the whole [C, 47, 47] map, i.e.
    a few rows of ~5e5 elements: torch's row-per-workgroup kernels then run on B (= 2-4) workgroups of a 256-CU GPU
    (measured 0.44 ms forward + 1.13 ms backward per layer).  bf16 CUDA maps take the multi-workgroup kernels of
    csrc/layernorm.hip; other large-row inputs compose the statistics from torch's multi-workgroup reductions; anything else is
    the stock path."""

    def forward(self, x):
        nd = len(self.normalized_shape)
        row = 1
        for d in self.normalized_shape:
            row *= int(d)
        big = (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16, torch.float16) and row >= (1 << 16) and x.numel() // row <= 64
               and self.elementwise_affine)
        if big and ENABLED and x.dtype == torch.bfloat16 and nd == 3 and x.dim() == 4 and row % 8 == 0 and self.bias is not None and (
                x.is_contiguous() or x.is_contiguous(memory_format=torch.channels_last)):
            return _WideLNFn.apply(x, self.weight, self.bias, self.eps)
        if big:
            dims = tuple(range(x.dim() - nd, x.dim()))
            xf = x.float()
            var, mean = torch.var_mean(xf, dim=dims, unbiased=False, keepdim=True)
            y = (xf - mean) * torch.rsqrt(var + self.eps) * self.weight.float() + self.bias.float()
            return y.to(x.dtype)
        return super().forward(x)
