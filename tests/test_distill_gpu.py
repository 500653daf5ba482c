"""BASELINE configs[2] on the MI355X: the S2D student (`KD_VoxelNet`) training forward and the whole teacher+student
distillation step (`train_step.distill_loss` = TS_Trainer.batch_processor_inline, CenterPoint branch,
/root/reference/det3d/torchie/trainer/trainer.py:775-811; detectors /root/reference/det3d/models/detectors/voxelnet.py:21-105,144-265)
against the ORACLE STACK: the same host code with every HIP launcher replaced by the CPU oracle (tests/cpu_backend.py:
oracle/voxelize.c, oracle/spconv_ref.py rulebooks + gather-mm-scatter, torch-CPU dense layers), in float64.

fp32 parity mode: features / loss terms within 2e-3 (the same bar as the single-stage detector test: MIOpen's fp32 conv
algorithms differ from the CPU's direct sums by ~1e-3 per layer).  The benchmarked mode (bf16 sparse storage + bf16 NHWC
dense kernels) is then held to the stated bf16 tolerance against the same oracle numbers: 5e-2 on every loss term."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import fill_params
from sparse2dense_amd import hip_ops as H
from sparse2dense_amd import waymo_configs
from sparse2dense_amd.registry import build_detector
from sparse2dense_amd.train_step import distill_loss

DEV = "cuda:0"
TERMS = ["sparse2dense_loss", "kd_hm_loss", "kd_reg_loss", "mask_loss", "reconstruction_loss", "hm_loss"]


def _to(ex, device, dtype=None):
    out = {}
    for k, v in ex.items():
        if isinstance(v, list):
            out[k] = [t.to(device=device, dtype=dtype) if (torch.is_tensor(t) and t.is_floating_point() and dtype) else
                      (t.to(device) if torch.is_tensor(t) else t) for t in v]
        elif torch.is_tensor(v):
            out[k] = v.to(device=device, dtype=dtype) if (v.is_floating_point() and dtype) else v.to(device)
        else:
            out[k] = v
    return out


@pytest.fixture(scope="module")
def setup():
    from sparse2dense_amd.data import SyntheticFrames
    frames = SyntheticFrames(1, n_points=12000, seed=31, distill=True, device=DEV)
    ex = frames.example()
    teacher = fill_params(build_detector(waymo_configs.centerpoint_voxelnet()), seed=1)
    student = fill_params(build_detector(waymo_configs.s2d_student()), seed=2)
    for p in teacher.parameters():
        p.requires_grad = False
    return ex, teacher, student


@pytest.fixture(scope="module")
def oracle_run(setup):
    """the oracle stack, float64, on the host"""
    import cpu_backend
    ex, teacher, student = setup
    mp = pytest.MonkeyPatch()
    try:
        cpu_backend.install(mp)
        t64, s64 = copy.deepcopy(teacher).double(), copy.deepcopy(student).double().train()
        ex64 = _to(ex, "cpu", torch.float64)
        feats = {}
        hook = s64.neck.register_forward_hook(lambda m, i, o: feats.update(F_S_a=o[5].detach(), F_S_b=o[6].detach()))
        total, losses = distill_loss(t64, s64, ex64)
        total.backward()
        hook.remove()
        res = dict(total=total.item(), terms={k: float(losses[k][0]) for k in TERMS},
                   F_S_a=feats["F_S_a"], F_S_b=feats["F_S_b"],
                   grads={n: p.grad.clone() for n, p in s64.named_parameters() if p.grad is not None})
    finally:
        mp.undo()
    return res


def _rel(a, b):
    return ((a.double().cpu() - b).norm() / (b.norm() + 1e-30)).item()


def test_kd_voxelnet_and_distill_step_fp32_vs_oracle_stack(setup, oracle_run):
    ex, teacher, student = setup
    t, s = copy.deepcopy(teacher).to(DEV), copy.deepcopy(student).to(DEV).train()
    feats = {}
    hook = s.neck.register_forward_hook(lambda m, i, o: feats.update(F_S_a=o[5].detach(), F_S_b=o[6].detach()))
    total, losses = distill_loss(t, s, ex)
    total.backward()
    hook.remove()
    assert not t.training and all(p.grad is None for p in t.parameters())
    np.testing.assert_allclose(total.item(), oracle_run["total"], rtol=2e-3)
    for k in TERMS:
        np.testing.assert_allclose(float(losses[k][0]), oracle_run["terms"][k], rtol=2e-3, atol=1e-6, err_msg=k)
    assert _rel(feats["F_S_a"], oracle_run["F_S_a"]) <= 2e-3 and _rel(feats["F_S_b"], oracle_run["F_S_b"]) <= 2e-3
    # every student parameter receives a gradient; norm-wise agreement with float64 (train-mode BN through 21 sparse + ~40
    # dense layers: the bar of tests/test_backbone_gpu.py::test_detector_loss_vs_oracle_stack)
    errs = {n: _rel(p.grad, oracle_run["grads"][n]) for n, p in s.named_parameters() if p.grad is not None}
    assert set(errs) == set(oracle_run["grads"])
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    print("fp32 distill step, worst gradient errors:", [(n, f"{e:.1e}") for n, e in worst])
    big = {n: e for n, e in errs.items() if oracle_run["grads"][n].norm() > 1e-8}
    assert max(big.values()) <= 5e-2, worst


def test_distill_step_benchmarked_bf16_mode_within_stated_tolerance(setup, oracle_run):
    ex, teacher, student = setup
    t, s = copy.deepcopy(teacher).to(DEV), copy.deepcopy(student).to(DEV).train()
    for m in (t, s):
        m.dense_dtype = torch.bfloat16
        m.use_channels_last()
    H.set_sparse_compute_dtype("s16")
    try:
        total, losses = distill_loss(t, s, ex)
        total.backward()
    finally:
        H.set_sparse_compute_dtype("f32")
    print("bf16 distill step:", {k: (float(losses[k][0]), oracle_run["terms"][k]) for k in TERMS})
    np.testing.assert_allclose(total.item(), oracle_run["total"], rtol=5e-2)
    for k in TERMS:
        np.testing.assert_allclose(float(losses[k][0]), oracle_run["terms"][k], rtol=5e-2, atol=1e-4, err_msg=k)
    grads = {n: p.grad for n, p in s.named_parameters() if p.grad is not None}
    assert set(grads) == set(oracle_run["grads"]) and all(torch.isfinite(g).all() for g in grads.values())
    # train-mode batch statistics under random weights amplify bf16 storage noise (see test_dense_modules.py); the numeric bar on the
    # gradients of the benchmarked mode is held by the well-conditioned variant below


def _bn_eval(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
            mod.eval()
    return m


@pytest.fixture(scope="module")
def oracle_run_bn_eval(setup):
    """the float64 oracle stack with every batch norm on its running statistics (fill_params: var in [0.5, 1.5])"""
    import cpu_backend
    ex, teacher, student = setup
    mp = pytest.MonkeyPatch()
    try:
        cpu_backend.install(mp)
        t64, s64 = copy.deepcopy(teacher).double(), _bn_eval(copy.deepcopy(student).double().train())
        ex64 = _to(ex, "cpu", torch.float64)
        feats = {}
        hook = s64.neck.register_forward_hook(lambda m, i, o: feats.update(F_S_a=o[5].detach(), F_S_b=o[6].detach()))
        total, losses = distill_loss(t64, s64, ex64)
        total.backward()
        hook.remove()
        res = dict(total=total.item(), terms={k: float(losses[k][0]) for k in TERMS}, F_S_a=feats["F_S_a"], F_S_b=feats["F_S_b"],
                   grads={n: p.grad.clone() for n, p in s64.named_parameters() if p.grad is not None})
    finally:
        mp.undo()
    return res


def test_distill_step_bf16_mode_well_conditioned_gradients_within_stated_tolerance(setup, oracle_run_bn_eval):
    """the BENCHMARKED mode (bf16 sparse storage + bf16 NHWC dense kernels) of the whole distillation step with
    the student's batch norms on their running statistics, against the float64 oracle stack (NO rounding emulation: this is the
    full price of bf16 storage against exact arithmetic).  Measured (r03): features 7-9e-3 (bar 2e-2), every loss term <= 3.3e-3
    (bar 2e-2; stated 5e-2), parameter gradients norm-wise:
      * head, RPN trunk, S2D module convs / batch norms, PCR head, backbone stage 4 + extra conv (192 of 243 tensors): <= 5.0e-2
        -> bar 6e-2;
      * the per-element LayerNorm([256,47,47]) scales (single products, no sum over pixels: the per-element noise of the bf16
        gradient field itself, cf. the input gradient in test_dense_modules.py): 1.9e-1 -> bar 2.5e-1;
      * the sparse backbone, growing with the number of bf16-stored layers behind the loss: stage 3 <= 8.8e-2 (bar 1.2e-1),
        stages 0-2 1e-1 .. 5e-1 (bar 6e-1 plus direction: cosine >= 0.85).  These are OUTSIDE the stated 5e-2: a first-stage
        gradient is the residual of cancelling terms after 26 dense + 20 sparse layers whose activations AND gradients are stored
        in bf16.  The kernels themselves are pinned against the float64 oracle WITH the same storage roundings at 3e-2
        (test_backbone_gpu.py::test_s16_backbone_gradients_vs_bf16_storage_oracle); the fp32 mode of the same step meets 5e-2 on
        every tensor (test_kd_voxelnet_and_distill_step_fp32_vs_oracle_stack).  DESIGN.md section 4 lists this as a parity gap of
        the benchmarked mode, not as met."""
    ex, teacher, student = setup
    ref = oracle_run_bn_eval
    t, s = copy.deepcopy(teacher).to(DEV), _bn_eval(copy.deepcopy(student).to(DEV).train())
    for m in (t, s):
        m.dense_dtype = torch.bfloat16
        m.use_channels_last()
    feats = {}
    hook = s.neck.register_forward_hook(lambda m, i, o: feats.update(F_S_a=o[5].detach(), F_S_b=o[6].detach()))
    H.set_sparse_compute_dtype("s16")
    try:
        total, losses = distill_loss(t, s, ex)
        total.backward()
    finally:
        H.set_sparse_compute_dtype("f32")
        hook.remove()
    ferr = {k: _rel(feats[k].float(), ref[k]) for k in ("F_S_a", "F_S_b")}
    lerr = {k: abs(float(losses[k][0]) - ref["terms"][k]) / (abs(ref["terms"][k]) + 1e-12) for k in TERMS}
    errs = {n: _rel(p.grad, ref["grads"][n]) for n, p in s.named_parameters() if p.grad is not None and ref["grads"][n].norm() > 1e-8}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    print("bf16 distill step, BN on running statistics: features", {k: f"{v:.1e}" for k, v in ferr.items()}, "losses",
          {k: f"{v:.1e}" for k, v in lerr.items()}, "worst gradients", [(n, f"{e:.1e}") for n, e in worst], f"({len(errs)} gradients)")
    print("gradients over 5e-2:", {n: f"{e:.1e}" for n, e in sorted(errs.items()) if e > 5e-2})
    assert set(n for n, p in s.named_parameters() if p.grad is not None) == set(ref["grads"])
    assert max(ferr.values()) <= 2e-2, ferr
    assert max(lerr.values()) <= 2e-2, lerr

    def bar(n):
        if n.startswith(("backbone.conv_input", "backbone.conv1", "backbone.conv2")):
            return 6e-1
        if n.startswith("backbone.conv3"):
            return 1.2e-1
        if n.startswith("neck.convnext_block") and n.split(".")[2] == "1":
            return 2.5e-1
        return 6e-2
    over = {n: (e, bar(n)) for n, e in errs.items() if e > bar(n)}
    assert not over, over
    grads = dict(s.named_parameters())
    for n in errs:
        if bar(n) == 6e-1:
            a, b = grads[n].grad.double().cpu().flatten(), ref["grads"][n].flatten()
            assert float(a @ b / (a.norm() * b.norm())) >= 0.85, n
