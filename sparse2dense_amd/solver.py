"""Optimizer + learning-rate schedule of the reference's training step.

  build_one_cycle_optimizer      /root/reference/det3d/torchie/apis/train.py:168-186
  OptimWrapper (fastai)          /root/reference/det3d/solver/fastai_optim.py:121-270   (Adam, "true" weight decay, bn_wd)
  OneCycle / LRSchedulerStep     /root/reference/det3d/solver/learning_schedules_fastai.py:8-95
  OptimizerHook.after_train_iter /root/reference/det3d/torchie/trainer/hooks/optimizer.py:15-21 (zero_grad, backward, clip 35, step)

What the reference's wrapper does per iteration, for every trainable parameter (its layer groups split BN from non-BN
parameters, but with `bn_wd=True` and one learning rate both halves get identical treatment):
    p <- p * (1 - wd * lr)                                  (decoupled decay, fastai_optim.py:161-170)
    Adam(betas=(mom, 0.99), eps=1e-8, weight_decay=0).step   (torch.optim.Adam; `mom` is scheduled, fastai_optim.py:206-213)
with lr and mom set by `OneCycle.step(global_step)` before the iteration (trainer.py:866-868).

`OneCycleAdam` keeps that arithmetic and runs it as ONE fused multi-tensor kernel per <= 48 tensors (csrc/optim.hip) with the
gradient-clip coefficient folded in as a device scalar: the norm is reduced on the device, never read by the host, and the
clipped gradients are never written back.  CPU tensors (host-logic tests) take an equivalent torch implementation.
State layout = torch.optim.Adam's (`state[p] = {step, exp_avg, exp_avg_sq}`).  `load_state_dict` accepts (a) this class's own
single-group state_dict and (b) the reference's: torch.optim.Adam over the TWO param groups `OptimWrapper.create` makes from
`split_bn_bias(get_layer_groups(model))` (fastai_optim.py:17-28, apis/train.py:159-164) - every non-batch-norm leaf module's
parameters first, then every batch-norm leaf's - which needs the model to rebuild the index map (`reference_param_order`).
Every moment tensor is shape-checked against its parameter and re-laid to the parameter's strides before the raw-pointer
kernel may see it; anything else is refused with a ValueError.  `state_dict` writes layout (b) when the model is attached (the
reference's torch.optim.Adam resumes from it: tests/test_solver_checkpoint.py), layout (a) otherwise.
"""
import ctypes
import math

import numpy as np
import torch


def annealing_cos(start, end, pct):
    """cosine anneal from `start` to `end` as pct goes 0 -> 1 (learning_schedules_fastai.py:67-71)"""
    return end + (start - end) / 2 * (math.cos(math.pi * pct) + 1)


class OneCycle:
    """lr: low -> lr_max over the first pct_start of the steps, then lr_max -> low/1e4; momentum: moms[0] -> moms[1] -> moms[0]
    (learning_schedules_fastai.py:77-95).  `step(i)` sets `optimizer.lr` / `optimizer.mom` for global step i."""

    def __init__(self, optimizer, total_step, lr_max, moms, div_factor, pct_start):
        self.optimizer, self.total_step = optimizer, int(total_step)
        self.lr_max, self.moms, self.div_factor, self.pct_start = lr_max, tuple(moms), div_factor, pct_start
        low = lr_max / div_factor
        split = int(pct_start * total_step)
        # (start, end, f(pct)) exactly as LRSchedulerStep lays the phases out (int() of the fractional starts)
        self.lr_phases = [(0, split, lambda p: annealing_cos(low, lr_max, p)),
                          (split, self.total_step, lambda p: annealing_cos(lr_max, low / 1e4, p))]
        self.mom_phases = [(0, split, lambda p: annealing_cos(self.moms[0], self.moms[1], p)),
                           (split, self.total_step, lambda p: annealing_cos(self.moms[1], self.moms[0], p))]
        if optimizer is not None:
            optimizer.lr, optimizer.mom = low, self.moms[0]

    def values(self, step):
        lr = mom = None
        for start, end, f in self.lr_phases:
            if step >= start:
                lr = f((step - start) / (end - start))
        for start, end, f in self.mom_phases:
            if step >= start:
                mom = f((step - start) / (end - start))
        return lr, mom

    def step(self, step):
        lr, mom = self.values(step)
        if lr is not None:
            self.optimizer.lr = lr
        if mom is not None:
            self.optimizer.mom = mom


def reference_param_order(model):
    """The parameter order of the reference's optimizer (= the indices of its optimizer state_dict): leaf modules in
    depth-first order (`flatten_model`, apis/train.py:159-160), non-batch-norm leaves first, then the batch-norm leaves
    (`split_bn_bias`, fastai_optim.py:17-28), trainable parameters only, shared parameters once.  Parameters owned by a module
    that also has children are not reached by the reference's grouping and are therefore not part of the order."""
    from torch import nn

    def leaves(m):
        ch = list(m.children())
        return sum((leaves(c) for c in ch), []) if ch else [m]

    ls = leaves(model)
    groups = ([m for m in ls if not isinstance(m, nn.modules.batchnorm._BatchNorm)], [m for m in ls if isinstance(m, nn.modules.batchnorm._BatchNorm)])
    out, seen = [], set()
    for g in groups:
        part = []
        for m in g:
            for q in m.parameters(recurse=False):
                if q.requires_grad and id(q) not in seen:
                    seen.add(id(q))
                    part.append(q)
        out.append(part)
    return out


class OneCycleAdam:
    """Adam with true weight decay under an externally scheduled lr / momentum (see module docstring)."""

    def __init__(self, params, lr=3e-3, mom=0.9, beta=0.99, eps=1e-8, wd=0.01, max_grad_norm=35.0, model=None):
        self.model = model      # only for load_state_dict of a reference-written optimizer state
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.mom, self.beta, self.eps, self.wd = lr, mom, beta, eps, wd
        self.max_grad_norm = max_grad_norm
        self.state = {}
        self.step_count = 0
        self._clip = None       # device [norm, clip_coef] of the last clip_and_step

    # -- torch.optim-style surface used by the reference's hooks ---------------------------------------------------
    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def _state(self, p):
        st = self.state.get(p)
        if st is None:
            st = dict(step=0, exp_avg=torch.zeros_like(p, memory_format=torch.preserve_format),
                      exp_avg_sq=torch.zeros_like(p, memory_format=torch.preserve_format))
            self.state[p] = st
        return st

    def state_dict(self):
        """With the model attached: the reference's layout - torch.optim.Adam state over the two split_bn_bias groups in
        `reference_param_order` (standard-layout moments, torch-Adam group keys), which `torch.optim.Adam(groups).load_state_dict` of a
        reference run resumes from and `load_state_dict` below maps back; the extra top-level keys (`wd`, `step_count`) are ignored
        by torch.  Without the model (or when the model's leaves do not cover this optimizer's parameters): one group in
        `self.params` order, this class's own layout."""
        if self.model is not None:
            groups = reference_param_order(self.model)
            order = [q for g in groups for q in g]
            if len(order) == len(self.params) and {id(q) for q in order} == {id(q) for q in self.params}:
                pos = {id(q): i for i, q in enumerate(order)}
                state = {}
                for p, st in self.state.items():
                    state[pos[id(p)]] = {k: (v.detach().clone(memory_format=torch.contiguous_format) if torch.is_tensor(v) else v) for k, v in st.items()}
                pg, k = [], 0
                for g in groups:
                    pg.append({"lr": self.lr, "betas": (self.mom, self.beta), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                               "params": list(range(k, k + len(g)))})
                    k += len(g)
                return {"state": state, "param_groups": pg, "wd": self.wd, "step_count": self.step_count}
        idx = {p: i for i, p in enumerate(self.params)}
        return {"state": {idx[p]: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()} for p, st in self.state.items()},
                "param_groups": [{"lr": self.lr, "betas": (self.mom, self.beta), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                                  "params": list(range(len(self.params)))}],
                "wd": self.wd, "step_count": self.step_count}

    def load_state_dict(self, sd, model=None):
        groups = sd["param_groups"]
        n_idx = sum(len(g["params"]) for g in groups)
        if len(groups) == 1 and n_idx == len(self.params):
            order = self.params                                    # this class's own layout
        else:
            model = self.model if model is None else model
            if model is None:
                raise ValueError(f"OneCycleAdam.load_state_dict: optimizer state with {len(groups)} param groups / {n_idx} parameters is not "
                                 f"this optimizer's own layout ({len(self.params)} parameters, one group); pass the model to map the reference's "
                                 "split_bn_bias ordering")
            ref_groups = reference_param_order(model)
            if [len(g["params"]) for g in groups] != [len(g) for g in ref_groups]:
                raise ValueError("OneCycleAdam.load_state_dict: param group sizes "
                                 f"{[len(g['params']) for g in groups]} do not match the reference ordering of this model {[len(g) for g in ref_groups]}")
            order = [q for g in ref_groups for q in g]
            ours = {id(q) for q in self.params}
            if any(id(q) not in ours for q in order):
                raise ValueError("OneCycleAdam.load_state_dict: the model has trainable parameters this optimizer does not own")
        flat_ids = [i for g in groups for i in g["params"]]
        pos = {int(i): k for k, i in enumerate(flat_ids)}
        new_state = {}
        for i, st in sd["state"].items():
            if int(i) not in pos:
                raise ValueError(f"OneCycleAdam.load_state_dict: state index {i} is in no param group")
            p = order[pos[int(i)]]
            ent = {}
            for k, v in st.items():
                if k == "step":
                    ent[k] = int(v.item()) if torch.is_tensor(v) else int(v)       # newer torch.optim.Adam keeps a tensor
                elif torch.is_tensor(v):
                    if tuple(v.shape) != tuple(p.shape):
                        raise ValueError(f"OneCycleAdam.load_state_dict: {k} of state {i} has shape {tuple(v.shape)}, its parameter {tuple(p.shape)}")
                    # the kernel pairs elements by storage offset: same strides as the parameter (NCHW file -> channels_last model)
                    ent[k] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(v)
                else:
                    ent[k] = v
            new_state[p] = ent
        self.state = new_state
        self._hip_cache = None   # moments were re-created: the cached pointer tables are stale
        g = groups[0]
        self.lr, (self.mom, self.beta), self.eps = g["lr"], tuple(g["betas"]), g["eps"]
        self.wd = sd.get("wd", self.wd)
        self.step_count = sd.get("step_count", max([st["step"] for st in self.state.values()] or [0]))

    # -- the step ----------------------------------------------------------------------------------------------------
    def clip_and_step(self, max_norm=None):
        """clip_grad_norm_(max_norm) folded into the update; returns the device scalar holding the total gradient norm."""
        max_norm = self.max_grad_norm if max_norm is None else max_norm
        ps = [p for p in self.params if p.grad is not None]
        self.step_count += 1
        if not ps:
            return None
        if ps[0].is_cuda:
            return self._step_hip(ps, max_norm)
        return self._step_torch(ps, max_norm)

    def step(self):
        """plain step on already clipped gradients (OptimWrapper.step, fastai_optim.py:158-171)"""
        return self.clip_and_step(max_norm=float("inf"))

    def _step_torch(self, ps, max_norm):
        norm = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(p.grad.detach().double()) for p in ps]))
        coef = 1.0 if math.isinf(max_norm) else torch.clamp(max_norm / (norm + 1e-6), max=1.0)
        with torch.no_grad():
            for p in ps:
                st = self._state(p)
                st["step"] += 1
                g = p.grad * coef
                p.mul_(1 - self.lr * self.wd)
                st["exp_avg"].mul_(self.mom).add_(g, alpha=1 - self.mom)
                st["exp_avg_sq"].mul_(self.beta).addcmul_(g, g, value=1 - self.beta)
                bc1, bc2 = 1 - self.mom ** st["step"], 1 - self.beta ** st["step"]
                p.addcdiv_(st["exp_avg"], st["exp_avg_sq"].sqrt() / math.sqrt(bc2) + self.eps, value=-self.lr / bc1)
        return norm

    def _step_hip(self, ps, max_norm):
        from . import _lib
        from .dense2d import _stream
        lib = _lib.load()
        # Everything that only depends on the parameter set (layout checks, element counts, the pointer tables of parameters and
        # moments, the launch chunks) is built once and reused while the same parameters, in the same order, receive gradients: the
        # per-step host work in front of the first launch was ~0.9 ms of an otherwise idle device (rocprofv3 gaps, r03).
        # The cache holds RAW device pointers: it is keyed on every parameter's storage pointer, and the moments' pointers are
        # re-read below - a parameter re-seated after the first step (use_channels_last(), load_state_dict(assign=True), p.data = ...)
        # or a moment replaced from outside must never be reached through a stale pointer.
        ids = tuple(map(id, ps)) + tuple(p.data_ptr() for p in ps)
        c = self._hip_cache if getattr(self, "_hip_cache", None) is not None and self._hip_cache["ids"] == ids else None
        if c is not None and any((st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()) != mv for st, mv in zip(c["states"], c["moments"])):
            c = None
        cap = lib.s2d_adam_max_tensors()
        vp = lambda ptrs: (ctypes.c_void_p * len(ptrs))(*ptrs)
        if c is None:
            dense = lambda t: t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))
            assert all(p.dtype == torch.float32 and dense(p) for p in ps), "OneCycleAdam: dense fp32 parameters expected"
            states = [self._state(p) for p in ps]
            numel = [p.numel() for p in ps]
            c = dict(ids=ids, states=states, numel=numel, numel_all=(ctypes.c_int64 * len(ps))(*numel), strides=[p.stride() for p in ps],
                     moments=[(st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()) for st in states], groups=None, group_steps=None,
                     norm_chunks=[(i, min(i + cap, len(ps)), (ctypes.c_int64 * (min(i + cap, len(ps)) - i))(*numel[i:i + cap]))
                                  for i in range(0, len(ps), cap)])
            c["ws_floats"] = lib.s2d_grad_norm_workspace_floats(len(ps), c["numel_all"])
            self._hip_cache = c
        states = c["states"]
        # the kernel pairs elements by storage offset: gradients (and moments, created with preserve_format) must share the
        # parameter's strides - NHWC conv weights (detector.use_channels_last) get their gradient re-laid once here
        # (strides of size-1 dimensions carry no layout: a 1x1 conv weight is the same memory in NCHW and NHWC)
        same = lambda g, p: all(sg == sp for sg, sp, n in zip(g.stride(), p.stride(), p.shape) if n > 1)
        f32, pstrides = torch.float32, c["strides"]
        grads, gptr = [], []
        for i, p in enumerate(ps):   # (one .grad access and one stride tuple per parameter: this loop is ~0.1 ms of idle device per step)
            g = p.grad
            if g.dtype is not f32 or (g.stride() != pstrides[i] and not same(g, p)):
                g = torch.empty_like(p).copy_(g)
            grads.append(g)
            gptr.append(g.data_ptr())
        # bias correction is a scalar per launch: parameters are grouped by their own step counter (torch.optim.Adam keeps one per
        # parameter; a branch that starts receiving gradients later - a toggled PCR head, an unfrozen layer - has a younger one)
        steps = []
        for st in states:
            st["step"] += 1
            steps.append(st["step"])
        if c["group_steps"] is None or any(s != t + 1 for s, t in zip(steps, c["group_steps"])):   # first use, or counters set from outside
            by_step = {}
            for k, s_ in enumerate(steps):
                by_step.setdefault(s_, []).append(k)
            groups = []
            for members in by_step.values():
                for i in range(0, len(members), cap):
                    ks = members[i:i + cap]
                    groups.append(dict(ks=ks, n=len(ks), p=vp([ps[k].data_ptr() for k in ks]), m=vp([c["moments"][k][0] for k in ks]),
                                       v=vp([c["moments"][k][1] for k in ks]), numel=(ctypes.c_int64 * len(ks))(*[c["numel"][k] for k in ks])))
            c["groups"] = groups
        c["group_steps"] = steps
        dev, stream = ps[0].device, _stream()
        clip = None
        if not math.isinf(max_norm):
            ws = torch.empty(c["ws_floats"], dtype=torch.float32, device=dev)
            written = 0
            for lo, hi, numel_chunk in c["norm_chunks"]:
                w = ctypes.c_int(0)
                _lib.check(lib.s2d_grad_sumsq_f32(hi - lo, vp(gptr[lo:hi]), numel_chunk, ws.data_ptr() + 4 * written, ctypes.byref(w), stream),
                           "s2d_grad_sumsq_f32")
                written += w.value
            clip = torch.empty(2, dtype=torch.float32, device=dev)
            _lib.check(lib.s2d_grad_norm_finalize_f32(ws.data_ptr(), written, float(max_norm), clip.data_ptr(), stream),
                       "s2d_grad_norm_finalize_f32")
        self._clip = clip
        for g in c["groups"]:
            ks = g["ks"]
            _lib.check(lib.s2d_adam_step_f32(g["n"], g["p"], vp([gptr[k] for k in ks]), g["m"], g["v"], g["numel"], float(self.lr), float(self.mom),
                                             float(self.beta), float(self.eps), float(self.wd), int(steps[ks[0]]),
                                             None if clip is None else clip.data_ptr() + 4, stream), "s2d_adam_step_f32")
        # the kernel updated the parameters through raw pointers: their autograd version counters did not move, so the packed weight
        # images keyed on them are rebuilt here (one HIP graph of the registered pack launches) or dropped (dense2d.refresh_pack_cache)
        from .dense2d import refresh_pack_cache
        refresh_pack_cache()
        return None if clip is None else clip[0]


def build_one_cycle_optimizer(model, optimizer_config=None):
    """apis/train.py:168-186: Adam betas (0.9, 0.99), true weight decay `wd` (config `optimizer.wd`, 0.01), bn_wd=True."""
    wd = 0.01 if optimizer_config is None else getattr(optimizer_config, "wd", optimizer_config.get("wd", 0.01)
                                                       if isinstance(optimizer_config, dict) else 0.01)
    return OneCycleAdam([p for p in model.parameters() if p.requires_grad], lr=3e-3, mom=0.9, beta=0.99, wd=wd, model=model)


def build_one_cycle_scheduler(optimizer, lr_config, total_steps):
    """configs `lr_config = dict(type="one_cycle", lr_max=0.003, moms=[0.95, 0.85], div_factor=10.0, pct_start=0.4)`"""
    cfg = dict(lr_config)
    assert cfg.pop("type", "one_cycle") == "one_cycle"
    return OneCycle(optimizer, total_steps, cfg["lr_max"], cfg["moms"], cfg["div_factor"], cfg["pct_start"])
