"""BASELINE configs[2] on the MI355X: the S2D student (`KD_VoxelNet`) training forward and the whole teacher+student
distillation step (`train_step.distill_loss` = TS_Trainer.batch_processor_inline, CenterPoint branch,
/root/reference/det3d/torchie/trainer/trainer.py:775-811; detectors /root/reference/det3d/models/detectors/voxelnet.py:21-105,144-265)
against the ORACLE STACK: the same host code with every HIP launcher replaced by the CPU oracle (tests/cpu_backend.py:
oracle/voxelize.c, oracle/spconv_ref.py rulebooks + gather-mm-scatter, torch-CPU dense layers), in float64.

fp32 parity mode: features / loss terms within 2e-3 (the same bar as the single-stage detector test: MIOpen's fp32 conv
algorithms differ from the CPU's direct sums by ~1e-3 per layer).  The benchmarked mode (bf16 sparse storage + bf16 NHWC
dense kernels) is then held to the stated bf16 tolerance against the same oracle numbers: 5e-2 on every loss term."""
import copy
import os

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


def _bn_eval(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
            mod.eval()
    return m


ORACLE_DEV = os.environ.get("S2D_ORACLE_DEVICE", DEV)   # "cpu": the host run of r01-r05 (minutes); default: float64 on the device, guarded


def _oracle_step(setup, bn_eval=False, storage_bf16=False):
    """one teacher + student distillation step of the float64 ORACLE STACK (tests/cpu_backend.py) on ORACLE_DEV"""
    import cpu_backend
    from golden_util import add_bf16_storage_hooks
    ex, teacher, student = setup
    with cpu_backend.oracle_stack(ORACLE_DEV, storage_bf16=storage_bf16):
        t64, s64 = copy.deepcopy(teacher).double(), copy.deepcopy(student).double().train()
        if bn_eval:
            _bn_eval(s64)
        if storage_bf16:
            with torch.no_grad():
                for m in (t64, s64):
                    for mod in list(m.neck.modules()) + list(m.bbox_head.modules()):
                        if isinstance(mod, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                            mod.weight.copy_(mod.weight.to(torch.bfloat16).double())
            for m in (t64, s64):
                add_bf16_storage_hooks(m.neck)
                add_bf16_storage_hooks(m.bbox_head)
        t64, s64 = t64.to(ORACLE_DEV), s64.to(ORACLE_DEV)
        ex64 = cpu_backend.to_device(ex, ORACLE_DEV, torch.float64)
        feats = {}
        hook = s64.neck.register_forward_hook(lambda m, i, o: feats.update(F_S_a=o[5].detach().cpu(), F_S_b=o[6].detach().cpu()))
        total, losses = distill_loss(t64, s64, ex64)
        total.backward()
        hook.remove()
        assert total.dtype == torch.float64
        return dict(total=total.item(), terms={k: float(losses[k][0]) for k in TERMS}, F_S_a=feats["F_S_a"], F_S_b=feats["F_S_b"],
                    grads={n: p.grad.detach().cpu() for n, p in s64.named_parameters() if p.grad is not None})


@pytest.fixture(scope="module")
def oracle_run(setup):
    """the oracle stack, float64"""
    return _oracle_step(setup)


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


@pytest.fixture(scope="module")
def oracle_run_bn_eval(setup):
    """the float64 oracle stack with every batch norm on its running statistics (fill_params: var in [0.5, 1.5])"""
    return _oracle_step(setup, bn_eval=True)


@pytest.fixture(scope="module")
def oracle_run_bn_eval_bf16_storage(setup):
    """the float64 oracle stack of `oracle_run_bn_eval` with the product's bf16 STORAGE points restated and nothing else changed:
    sparse rows (cpu_backend.STORAGE_BF16: conv outputs forward and backward, fused BN outputs and their gradients, bf16 weight
    images), 2-D neck / head layer outputs and their gradients + bf16 conv weights (golden_util.add_bf16_storage_hooks); accumulation,
    statistics, losses and the 3-D PCR head exact.  Its distance to the exact run is what bf16 storage costs ANY arithmetic."""
    return _oracle_step(setup, bn_eval=True, storage_bf16=True)


def test_distill_step_bf16_mode_against_the_bf16_storage_oracle(setup, oracle_run_bn_eval, oracle_run_bn_eval_bf16_storage):
    """what the benchmarked mode's early-backbone gradient gap (1e-1 .. 5e-1 against the exact float64 oracle) IS.
    r04 decomposition on the GPU (scratch run, same fixture): fp32 sparse + bf16 dense neck -> conv1 4.4e-1; bf16 sparse + fp32
    dense -> 1.2e-1; fp32 gradient ROWS change nothing (oracle experiment: backward-only rounding 7e-3, forward-only 5e-2).  The
    deviation is produced by bf16 rounding of FORWARD activations: ~0.5 % of the ReLU / GELU' decisions flip, each flip switches a
    gradient path on or off, and the white noise this adds to dBEV (8e-2 norm-wise) is damped less than the coherent signal on its
    way back through 20 sparse layers.  It is a property of the storage type, not of a kernel - shown here by restating ONLY the
    storage roundings in the float64 CPU stack:
      (a) `spread` = bf16-storage oracle vs exact oracle, i.e. how far the REFERENCE arithmetic itself moves under bf16 storage.
          Measured: conv_input 2.1e-1, conv1 4.8e-1, conv2 2.2e-1, conv3 9.0e-2, conv4 3.6e-2, LayerNorm scales 1.9e-1, everything
          else <= 4.6e-2 - the same figures the GPU shows against the exact oracle (2.1e-1, 4.1e-1, 2.1e-1, 7.7e-2, 4.2e-2, 1.9e-1).
          The amplification is in the (exact) backward pass: with an fp32 sparse stack and only the neck in bf16 the error still
          doubles per sparse stage on the way back (2.7e-2 at extra_conv -> 4.4e-1 at conv1);
      (b) where the problem is well conditioned (spread <= 3e-2: RPN trunk, up-samplers, fusion convs, PCR head, CenterHead - 150+
          tensors) the GPU matches the bf16-storage oracle to <= 5e-2 (measured <= 1.7e-2): the kernels compute what the storage
          type prescribes.  Where it is not, two bf16-storage runs that differ only in ACCUMULATION precision (fp32 on the GPU,
          exact here) already differ by as much as either differs from the exact run (conv1 4.2e-1): the early-stage gradient of this
          randomly initialised student is 2:1 signal to flip-noise under bf16 storage, whoever computes it;
      (c) GPU vs the exact oracle stays inside 2 x spread + 5e-2 for every tensor.
    The fp32 mode of the same step holds 5e-2 on every tensor (first test of this file) and is benchmarked beside the bf16 number
    (`other_workloads.s2d_student_fp32`)."""
    ex, teacher, student = setup
    exact, emul = oracle_run_bn_eval, oracle_run_bn_eval_bf16_storage
    t, s = copy.deepcopy(teacher).to(DEV), _bn_eval(copy.deepcopy(student).to(DEV).train())
    for m in (t, s):
        m.dense_dtype = torch.bfloat16
        m.use_channels_last()
    H.set_sparse_compute_dtype("s16")
    try:
        total, losses = distill_loss(t, s, ex)
        total.backward()
    finally:
        H.set_sparse_compute_dtype("f32")
    names = [n for n, p in s.named_parameters() if p.grad is not None and exact["grads"][n].norm() > 1e-8]
    grads = dict(s.named_parameters())
    spread = {n: _rel(emul["grads"][n], exact["grads"][n]) for n in names}
    vs_emul = {n: _rel(grads[n].grad, emul["grads"][n]) for n in names}
    vs_exact = {n: _rel(grads[n].grad, exact["grads"][n]) for n in names}

    def group(d):
        g = {}
        for n, e in d.items():
            k = ".".join(n.split(".")[:2])
            g[k] = max(g.get(k, 0.0), e)
        return {k: f"{v:.1e}" for k, v in g.items()}
    print("spread of the float64 stack under bf16 storage:", group(spread))
    print("GPU vs bf16-storage oracle:", group(vs_emul))
    print("GPU vs exact oracle:", group(vs_exact))
    lerr = {k: abs(float(losses[k][0]) - emul["terms"][k]) / (abs(emul["terms"][k]) + 1e-12) for k in TERMS}
    assert max(lerr.values()) <= 2e-2, lerr

    well = [n for n in names if spread[n] <= 3e-2]
    assert len(well) >= 120, len(well)
    over = {n: (vs_emul[n], spread[n]) for n in well if vs_emul[n] > 5e-2}
    assert not over, over
    loose = {n: (vs_exact[n], spread[n]) for n in names if vs_exact[n] > 2 * spread[n] + 5e-2}
    assert not loose, loose
    # the storage type, not the kernels, sets the early-stage figures: the reference arithmetic moves at least half as far
    for stage in ("backbone.conv_input", "backbone.conv1", "backbone.conv2"):
        sp = max(spread[n] for n in names if n.startswith(stage + "."))
        ge = max(vs_exact[n] for n in names if n.startswith(stage + "."))
        assert sp >= 0.5 * ge, (stage, sp, ge)


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
