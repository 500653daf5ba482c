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
State layout = torch.optim.Adam's (`state[p] = {step, exp_avg, exp_avg_sq}`), so `state_dict()` round-trips through the
reference's checkpoint format.
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


class OneCycleAdam:
    """Adam with true weight decay under an externally scheduled lr / momentum (see module docstring)."""

    def __init__(self, params, lr=3e-3, mom=0.9, beta=0.99, eps=1e-8, wd=0.01, max_grad_norm=35.0):
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
        idx = {p: i for i, p in enumerate(self.params)}
        return {"state": {idx[p]: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()} for p, st in self.state.items()},
                "param_groups": [{"lr": self.lr, "betas": (self.mom, self.beta), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                                  "params": list(range(len(self.params)))}],
                "wd": self.wd, "step_count": self.step_count}

    def load_state_dict(self, sd):
        for i, st in sd["state"].items():
            p = self.params[int(i)]
            self.state[p] = {k: (v.to(p.device) if torch.is_tensor(v) else v) for k, v in st.items()}
        g = sd["param_groups"][0]
        self.lr, (self.mom, self.beta), self.eps = g["lr"], g["betas"], g["eps"]
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
        dense = lambda t: t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))
        assert all(p.dtype == torch.float32 and dense(p) for p in ps), "OneCycleAdam: dense fp32 parameters expected"
        # the kernel pairs elements by storage offset: gradients (and moments, created with preserve_format) must share the
        # parameter's strides - NHWC conv weights (detector.use_channels_last) get their gradient re-laid once here
        # (strides of size-1 dimensions carry no layout: a 1x1 conv weight is the same memory in NCHW and NHWC)
        same = lambda g, p: all(sg == sp for sg, sp, n in zip(g.stride(), p.stride(), p.shape) if n > 1)
        grads = [p.grad if (p.grad.dtype == torch.float32 and same(p.grad, p)) else torch.empty_like(p).copy_(p.grad) for p in ps]
        states = [self._state(p) for p in ps]
        steps = {st["step"] for st in states}
        assert len(steps) == 1, "OneCycleAdam: parameters must share one step counter"
        step = states[0]["step"] + 1
        for st in states:
            st["step"] = step
        dev, stream = ps[0].device, _stream()
        cap = lib.s2d_adam_max_tensors()
        vp = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        numel_all = (ctypes.c_int64 * len(ps))(*[p.numel() for p in ps])
        clip = None
        if not math.isinf(max_norm):
            ws = torch.empty(lib.s2d_grad_norm_workspace_floats(len(ps), numel_all), dtype=torch.float32, device=dev)
            written = 0
            for i in range(0, len(ps), cap):
                chunk = grads[i:i + cap]
                w = ctypes.c_int(0)
                _lib.check(lib.s2d_grad_sumsq_f32(len(chunk), vp(chunk), (ctypes.c_int64 * len(chunk))(*[g.numel() for g in chunk]),
                                                  ws.data_ptr() + 4 * written, ctypes.byref(w), stream), "s2d_grad_sumsq_f32")
                written += w.value
            clip = torch.empty(2, dtype=torch.float32, device=dev)
            _lib.check(lib.s2d_grad_norm_finalize_f32(ws.data_ptr(), written, float(max_norm), clip.data_ptr(), stream),
                       "s2d_grad_norm_finalize_f32")
        self._clip = clip
        for i in range(0, len(ps), cap):
            sl = slice(i, i + cap)
            n = len(ps[sl])
            _lib.check(lib.s2d_adam_step_f32(n, vp(ps[sl]), vp(grads[sl]), vp([st["exp_avg"] for st in states[sl]]),
                                             vp([st["exp_avg_sq"] for st in states[sl]]),
                                             (ctypes.c_int64 * n)(*[p.numel() for p in ps[sl]]), float(self.lr), float(self.mom),
                                             float(self.beta), float(self.eps), float(self.wd), int(step),
                                             None if clip is None else clip.data_ptr() + 4, stream), "s2d_adam_step_f32")
        # the kernel updated the parameters through raw pointers: their autograd version counters did not move, so drop the
        # packed weight images keyed on them (they are rebuilt at the next forward, as after any optimizer step)
        from .dense2d import clear_pack_cache
        clear_pack_cache()
        return None if clip is None else clip[0]


def build_one_cycle_optimizer(model, optimizer_config=None):
    """apis/train.py:168-186: Adam betas (0.9, 0.99), true weight decay `wd` (config `optimizer.wd`, 0.01), bn_wd=True."""
    wd = 0.01 if optimizer_config is None else getattr(optimizer_config, "wd", optimizer_config.get("wd", 0.01)
                                                       if isinstance(optimizer_config, dict) else 0.01)
    return OneCycleAdam([p for p in model.parameters() if p.requires_grad], lr=3e-3, mom=0.9, beta=0.99, wd=wd)


def build_one_cycle_scheduler(optimizer, lr_config, total_steps):
    """configs `lr_config = dict(type="one_cycle", lr_max=0.003, moms=[0.95, 0.85], div_factor=10.0, pct_start=0.4)`"""
    cfg = dict(lr_config)
    assert cfg.pop("type", "one_cycle") == "one_cycle"
    return OneCycle(optimizer, total_steps, cfg["lr_max"], cfg["moms"], cfg["div_factor"], cfg["pct_start"])
