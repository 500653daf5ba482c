"""Whole-backbone / whole-detector parity on the MI355X: HIP path (voxelizer -> rulebooks -> MFMA
sparse convs -> fused BN -> densify [-> neck -> head -> loss]) against the CPU oracle stack with
the same name-seeded weights.  fp32 tolerance end-to-end through 21 sparse convs + 21 batch-stat
BNs: rtol 2e-3 / atol 2e-4 on features, rtol 1e-3 on scalar losses."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import fill_params
from oracle import spconv_ref as R
from oracle import voxelize as OV
from sparse2dense_amd import backbones, scene, waymo_configs
from sparse2dense_amd.registry import build_backbone, build_detector

DEV = "cuda:0"
# the float64 / fp32 oracle modules (oracle/spconv_ref.py: torch index_select / mm / index_add, numpy rulebooks) run on ODEV: the device by
# default (seconds instead of minutes; tests/test_oracle_device.py holds device == host), "cpu" with S2D_ORACLE_DEVICE=cpu
ODEV = os.environ.get("S2D_ORACLE_DEVICE", DEV)


def _scene_voxels(n_points, seed, batch=1):
    feats, coors = [], []
    for b in range(batch):
        s = scene.make_scene(n_points, seed=seed + b)
        v, c, n = OV.points_to_voxel(s["points"], scene.WAYMO_VOXEL, scene.WAYMO_RANGE, 5, 150000)
        feats.append(OV.voxel_mean(v, n))
        coors.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
    return np.concatenate(feats), np.concatenate(coors)


def _rel_errors(grads, exact):
    out = {}
    for name, b in exact.items():
        a = grads[name]
        if a is None or b is None:
            assert a is None and b is None, name
            continue
        b = b.cpu()
        out[name] = ((a.double().cpu() - b).norm() / (b.norm() + 1e-30)).item()
    return out


def _grad_dict(module):
    return {n: p.grad for n, p in module.named_parameters()}


def _is_bn_fed_conv_bias(name):
    # conv bias feeding a batch-statistics BN: the exact gradient is zero (BN removes the mean)
    return name.endswith("conv1.bias") or name.endswith("conv2.bias")


def _compare_grads(net, ref64, tol, ref32=None, slack=3.0):
    """Norm-wise relative error of every parameter gradient against the float64 oracle.  Training-mode
    BN makes the fp32 backward of this 21-layer stack ill-conditioned (the fp32 CPU oracle itself is
    ~1e-2 off the fp64 one), so when `ref32` is given the bar is: the HIP path must be at most
    `slack` x as far from exact arithmetic as the fp32 CPU restatement is, or within `tol`."""
    exact = _grad_dict(ref64)
    e_gpu = _rel_errors(_grad_dict(net), exact)
    e_cpu = _rel_errors(_grad_dict(ref32), exact) if ref32 is not None else {}
    for name, err in sorted(e_gpu.items()):
        if _is_bn_fed_conv_bias(name) and net.training:
            wn = exact[name[:-4] + "weight"].norm().item()
            assert _grad_dict(net)[name].norm().item() <= 1e-3 * wn + 1e-6, name
            continue
        bar = max(tol, slack * e_cpu.get(name, 0.0))
        assert err <= bar, (name, err, bar)


@pytest.mark.parametrize("kind,ref_cls,channels", [("SpMiddleResNetFHD", R.RefSpMiddleResNetFHD, 256),
                                                  ("SpMiddleFHD", R.RefSpMiddleFHD, 128)])
def test_backbone_forward_backward_vs_oracle(kind, ref_cls, channels):
    feats, coors = _scene_voxels(8000, seed=7, batch=2)   # BASELINE config 1 scene ("second8k"), B=2
    net = fill_params(build_backbone(dict(type=kind, num_input_features=5, ds_factor=8))).train()
    ref = fill_params(ref_cls(5)).double().train().to(ODEV)   # float64 oracle = exact arithmetic for our purposes
    ref32 = fill_params(ref_cls(5)).train()                    # fp32 oracle ON THE HOST: calibrates the conditioning of the backward (the
    # bar below was set against the host's fp32 summation order in r01; torch's device kernels happen to land 3x closer to float64)
    assert sorted(net.state_dict()) == sorted(ref.state_dict())
    grid = np.array([1504, 1504, 40])
    bev_ref, ms_ref = ref(torch.from_numpy(feats).double().to(ODEV), coors, 2, grid)
    bev32, _ = ref32(torch.from_numpy(feats), coors, 2, grid)
    net = net.to(DEV)
    bev, ms = net(torch.from_numpy(feats).to(DEV), torch.from_numpy(coors).to(DEV), 2, grid)
    assert bev.shape == (2, channels, 188, 188) == bev_ref.shape
    torch.testing.assert_close(bev.cpu().double(), bev_ref.detach().cpu(), rtol=1e-3, atol=1e-4)
    if kind == "SpMiddleResNetFHD":
        for k in ["conv1", "conv2", "conv3", "conv4"]:
            assert np.array_equal(ms[k].indices.cpu().numpy(), ms_ref[k].indices), k
            torch.testing.assert_close(ms[k].features.cpu().double(), ms_ref[k].features.detach().cpu(), rtol=1e-3, atol=1e-4)
    # running statistics updated identically (momentum 0.01, unbiased variance)
    sd, sr = net.state_dict(), ref.state_dict()
    for k in sd:
        if "running" in k:
            torch.testing.assert_close(sd[k].cpu().double(), sr[k].cpu(), rtol=1e-4, atol=1e-6)
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(sr[k]) == 1
    g = torch.randn(bev_ref.shape, generator=torch.Generator().manual_seed(5))
    (bev_ref * g.double().to(ODEV)).sum().backward()
    (bev32 * g).sum().backward()
    (bev * g.to(DEV)).sum().backward()
    _compare_grads(net, ref, tol=5e-3, ref32=ref32)


def test_backbone_eval_mode_forward_backward_matches_oracle():
    """Running-stat BN (the teacher's mode) is well conditioned: tight tolerance on every gradient."""
    feats, coors = _scene_voxels(8000, seed=11)
    net = fill_params(build_backbone(dict(type="SpMiddleResNetFHD", num_input_features=5))).eval().to(DEV)
    ref = fill_params(R.RefSpMiddleResNetFHD(5)).double().eval().to(ODEV)
    grid = np.array([1504, 1504, 40])
    a, _ = net(torch.from_numpy(feats).to(DEV), torch.from_numpy(coors).to(DEV), 1, grid)
    b, _ = ref(torch.from_numpy(feats).double().to(ODEV), coors, 1, grid)
    torch.testing.assert_close(a.cpu().double(), b.detach().cpu(), rtol=1e-3, atol=1e-4)
    g = torch.randn(b.shape, generator=torch.Generator().manual_seed(6))
    (b * g.double().to(ODEV)).sum().backward()
    (a * g.to(DEV)).sum().backward()
    _compare_grads(net, ref, tol=5e-4)


def test_detector_loss_vs_oracle_stack():
    """CenterPoint-voxelnet single stage: loss of the HIP detector == loss of (oracle backbone +
    the same torch neck/head on CPU) for identical weights and targets."""
    from sparse2dense_amd import necks, heads  # noqa: F401
    from sparse2dense_amd.data import SyntheticFrames
    frames = SyntheticFrames(1, n_points=20000, seed=31)
    ex = frames.example()
    model = fill_params(build_detector(waymo_configs.centerpoint_voxelnet())).train()
    ref_bb = R.RefSpMiddleResNetFHD(5)
    ref_bb.load_state_dict(model.backbone.state_dict())
    ref_bb.double().train().to(ODEV)
    import copy
    import cpu_backend
    cpu_neck, cpu_head = copy.deepcopy(model.neck).double().train().to(ODEV), copy.deepcopy(model.bbox_head).double().train().to(ODEV)
    # oracle side: CPU voxelizer + oracle backbone + torch-CPU neck/head
    pts = frames.points[0].cpu().numpy()
    v, c, n = OV.points_to_voxel(pts, scene.WAYMO_VOXEL, scene.WAYMO_RANGE, 5, 150000)
    assert np.array_equal(ex["coordinates"][:, 1:].cpu().numpy(), c)
    coors = np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1)
    with cpu_backend.oracle_stack(ODEV):   # (the neck / head are the product's torch modules: no s2d kernel may serve the float64 run)
        bev, _ = ref_bb(torch.from_numpy(OV.voxel_mean(v, n)).double().to(ODEV), coors, 1, np.array([1504, 1504, 40]))
        preds = cpu_head(cpu_neck(bev))
        ex_cpu = {k: [t.to(ODEV).double() if t.is_floating_point() else t.to(ODEV) for t in ex[k]]
                  for k in ["hm", "anno_box", "ind", "mask", "cat"]}
        loss_ref = sum(cpu_head.loss(ex_cpu, preds)["loss"])
        loss_ref.backward()
    model = model.to(DEV)
    losses = model(ex, return_loss=True)
    loss = sum(losses["loss"])
    loss.backward()
    np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=1e-3)
    _compare_grads(model.backbone, ref_bb, tol=5e-2)   # train-mode BN conditioning + MIOpen's fp32 conv algorithms in neck/head


def test_bf16_mode_within_stated_tolerance():
    """BASELINE configs[1] names bf16: MFMA inputs rounded to bf16 (fp32 accumulate, fp32 storage /
    statistics / master weights).  Stated tolerance vs the fp32/fp64 oracle:
    2e-2 on features, 5e-2 on the loss."""
    from sparse2dense_amd import hip_ops as H
    from sparse2dense_amd.data import SyntheticFrames
    feats, coors = _scene_voxels(8000, seed=7, batch=2)
    net = fill_params(build_backbone(dict(type="SpMiddleResNetFHD", num_input_features=5))).train().to(DEV)
    ref = fill_params(R.RefSpMiddleResNetFHD(5)).double().train().to(ODEV)
    grid = np.array([1504, 1504, 40])
    with torch.no_grad():
        b = ref(torch.from_numpy(feats).double().to(ODEV), coors, 2, grid)[0].cpu()
    H.set_sparse_compute_dtype("bf16")
    try:
        a, _ = net(torch.from_numpy(feats).to(DEV), torch.from_numpy(coors).to(DEV), 2, grid)
        err = ((a.cpu().double() - b).norm() / b.norm()).item()
        assert err <= 2e-2, err
        # whole detector, bf16 sparse + bf16 NHWC dense, against its own fp32 run
        frames = SyntheticFrames(1, n_points=20000, seed=31)
        ex = frames.example()
        model = fill_params(build_detector(waymo_configs.centerpoint_voxelnet())).train().to(DEV)
        H.set_sparse_compute_dtype("f32")
        l32 = sum(model(ex, return_loss=True)["loss"]).item()
        H.set_sparse_compute_dtype("bf16")
        model.dense_dtype = torch.bfloat16
        model.use_channels_last()
        l16 = sum(model(ex, return_loss=True)["loss"])
        l16.backward()
        assert abs(l16.item() - l32) <= 5e-2 * abs(l32), (l16.item(), l32)
        assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    finally:
        H.set_sparse_compute_dtype("f32")


def test_s16_storage_mode_within_stated_tolerance():
    """bf16 feature STORAGE in the sparse stack (fp32 accumulate / statistics / master weights): SURVEY.md §8(c)
    "bf16 storage/fp32 accumulate vs fp32 oracle: rtol 2e-2 on features, 5e-2 on losses"."""
    from sparse2dense_amd import hip_ops as H
    from sparse2dense_amd.data import SyntheticFrames
    feats, coors = _scene_voxels(8000, seed=7, batch=2)
    net = fill_params(build_backbone(dict(type="SpMiddleResNetFHD", num_input_features=5))).train().to(DEV)
    ref = fill_params(R.RefSpMiddleResNetFHD(5)).double().train().to(ODEV)
    grid = np.array([1504, 1504, 40])
    with torch.no_grad():
        b = ref(torch.from_numpy(feats).double().to(ODEV), coors, 2, grid)[0].cpu()
    H.set_sparse_compute_dtype("s16")
    try:
        a, aux = net(torch.from_numpy(feats).to(DEV), torch.from_numpy(coors).to(DEV), 2, grid)
        assert a.dtype == torch.float32 and aux["conv4"].features.dtype == torch.bfloat16
        err = ((a.cpu().double() - b).norm() / b.norm()).item()
        assert err <= 2e-2, err
        a2, _ = net(torch.from_numpy(feats).to(DEV), torch.from_numpy(coors).to(DEV), 2, grid, bev_nhwc_bf16=True)
        assert a2.dtype == torch.bfloat16 and (a2.float() - a).abs().max() <= 1e-2 * a.abs().max()
        frames = SyntheticFrames(1, n_points=20000, seed=31)
        ex = frames.example()
        model = fill_params(build_detector(waymo_configs.centerpoint_voxelnet())).train().to(DEV)
        H.set_sparse_compute_dtype("f32")
        l32 = sum(model(ex, return_loss=True)["loss"])
        l32.backward()
        g32 = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        model.zero_grad()
        H.set_sparse_compute_dtype("s16")
        model.dense_dtype = torch.bfloat16
        model.use_channels_last()
        l16 = sum(model(ex, return_loss=True)["loss"])
        l16.backward()
        assert abs(l16.item() - l32.item()) <= 5e-2 * abs(l32.item()), (l16.item(), l32.item())
        assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    finally:
        H.set_sparse_compute_dtype("f32")


def _round_conv_weights_to_bf16(module):
    """the s16 kernels read bf16 weight images: start both sides from weights that are exactly representable"""
    with torch.no_grad():
        for p in module.parameters():
            if p.dim() == 5:
                p.copy_(p.to(torch.bfloat16).to(p.dtype))
    return module


@pytest.mark.parametrize("train", [False, True])
def test_s16_backbone_gradients_vs_bf16_storage_oracle(train):
    """The BENCHMARKED mode (bf16 feature storage) against the float64 oracle restated with the same storage roundings
    (oracle/spconv_ref.py `bf16_storage`: input, every conv output, every fused BN(+residual)(+ReLU) output and the
    gradients flowing back through those points are rounded to bf16; accumulation and statistics exact).  What is left
    is fp32-vs-exact accumulation, i.e. rare one-bf16-ulp flips.  Eval-mode BN (the teacher's mode) is well conditioned:
    forward 5e-3 (measured 1.6e-4, r02); gradients 3e-2 (measured 0.5-1.8e-2: a forward difference of 1e-4 flips that
    fraction of ReLU masks, which moves a gradient norm-wise by its square root).  Train-mode BN: the same bars, or at most
    3x the error the fp32 CPU restatement itself shows against float64 (the `_compare_grads(ref32=...)` calibration of the
    conditioning; measured forward 1.02e-2 vs 1.03e-2 for the fp32 restatement)."""
    from sparse2dense_amd import hip_ops as H
    feats, coors = _scene_voxels(8000, seed=7, batch=2)
    grid = np.array([1504, 1504, 40])
    mk = lambda m: (m.train() if train else m.eval())
    net = mk(_round_conv_weights_to_bf16(fill_params(build_backbone(dict(type="SpMiddleResNetFHD", num_input_features=5)))))
    ref = mk(_round_conv_weights_to_bf16(fill_params(R.RefSpMiddleResNetFHD(5))).double()).to(ODEV)
    ref32 = mk(_round_conv_weights_to_bf16(fill_params(R.RefSpMiddleResNetFHD(5)))) if train else None   # (host: see above)
    g = torch.randn((2, 256, 188, 188), generator=torch.Generator().manual_seed(5))
    with R.bf16_storage():
        b, ms_ref = ref(torch.from_numpy(feats).double().to(ODEV), coors, 2, grid)
        (b * g.double().to(ODEV)).sum().backward()
        if train:
            b32, _ = ref32(torch.from_numpy(feats), coors, 2, grid)
            (b32 * g).sum().backward()
    b = b.detach().cpu()
    net = net.to(DEV)
    H.set_sparse_compute_dtype("s16")
    try:
        a, ms = net(torch.from_numpy(feats).to(DEV), torch.from_numpy(coors).to(DEV), 2, grid)
        (a * g.to(DEV)).sum().backward()
    finally:
        H.set_sparse_compute_dtype("f32")
    assert ms["conv4"].features.dtype == torch.bfloat16
    err = ((a.cpu().double() - b).norm() / b.norm()).item()
    e32 = ((b32.detach().double().cpu() - b).norm() / b.norm()).item() if train else 0.0
    print(f"s16 vs bf16-storage oracle (train={train}): forward {err:.2e} (fp32 oracle {e32:.2e})")
    assert err <= max(5e-3, 3 * e32), err
    for k in ["conv1", "conv2", "conv3", "conv4"]:
        fr = ms_ref[k].features.detach().cpu()
        fe = ((ms[k].features.double().cpu() - fr).norm() / fr.norm()).item()
        assert fe <= max(5e-3, 3 * e32), (k, fe)
    errs = _rel_errors(_grad_dict(net), _grad_dict(ref))
    print("  gradient errors:", {k: f"{v:.1e}" for k, v in sorted(errs.items()) if k.endswith("weight") and "conv" in k})
    _compare_grads(net, ref, tol=3e-2, ref32=ref32)


def test_second_config1_forward_vs_cpu_reference_path():
    """BASELINE config 1: SECOND voxelnet on the 8k-pt cloud, batch 1 — HIP path vs the CPU reference
    path (C voxelizer + oracle SpMiddleFHD + the same torch RPN / MultiGroupHead forward on CPU)."""
    import copy
    from sparse2dense_amd.data import SyntheticFrames
    frames = SyntheticFrames(1, n_points=8000, seed=7)
    ex = frames.example()
    model = fill_params(build_detector(waymo_configs.second_voxelnet())).eval()
    ref_bb = R.RefSpMiddleFHD(5)
    ref_bb.load_state_dict(model.backbone.state_dict())
    ref_bb.double().eval().to(ODEV)
    import cpu_backend
    neck, head = copy.deepcopy(model.neck).double().eval().to(ODEV), copy.deepcopy(model.bbox_head).double().eval().to(ODEV)
    pts = frames.points[0].cpu().numpy()
    v, c, n = OV.points_to_voxel(pts, scene.WAYMO_VOXEL, scene.WAYMO_RANGE, 5, 150000)
    coors = np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1)
    with torch.no_grad():
        with cpu_backend.oracle_stack(ODEV):
            bev, _ = ref_bb(torch.from_numpy(OV.voxel_mean(v, n)).double().to(ODEV), coors, 1, np.array([1504, 1504, 40]))
            ref = {k: t.cpu() for k, t in head(neck(bev))[0].items()}
        out = model.to(DEV)(ex, return_loss=False, raw_preds=True)[0]
    assert out["box_preds"].shape == (1, 188, 188, 42) and out["cls_preds"].shape == (1, 188, 188, 18)
    assert out["dir_cls_preds"].shape == (1, 188, 188, 12)
    for k in ["box_preds", "cls_preds", "dir_cls_preds"]:
        err = ((out[k].cpu().double() - ref[k]).norm() / ref[k].norm()).item()
        assert err <= 2e-3, (k, err)
