"""The benchmark's cloud size against the ORACLE (VERDICT r05 #5a: tests/test_full_size_gpu.py compares the bf16 mode with the fp32 mode,
i.e. the HIP path with itself): one 150 000-point frame, S2D student training forward + backward,
  HIP fp32 mode  vs  the float64 oracle stack (tests/cpu_backend.py: C voxelizer, numpy rulebooks, gather-mm-scatter, torch dense
                     layers; run on the device through torch's own float64 kernels with the s2d library closed off)
  * voxel coordinates / counts: bit-exact vs oracle/voxelize.c;
  * the eight rulebooks of the backbone pass (4 submanifold + 4 strided: neighbour maps both directions, pair counts, output coordinates)
    and the per-stage `indices`: bit-exact vs oracle/spconv_ref.py;
  * BEV map 5e-3 norm-wise, every loss term 1e-3, gradient norm 2e-2 (train-mode batch norms).
Reference: /root/reference/det3d/models/detectors/voxelnet.py:171-265, /root/reference/det3d/models/backbones/scn.py:88-185."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cpu_backend
from golden_util import fill_params
from oracle import voxelize as OV
from sparse2dense_amd import backbones, scene, waymo_configs
from sparse2dense_amd.registry import build_detector

DEV = "cuda:0"
ODEV = os.environ.get("S2D_ORACLE_DEVICE", DEV)
POINTS = 150000


@pytest.fixture(scope="module")
def setup():
    from sparse2dense_amd.data import SyntheticFrames
    frames = SyntheticFrames(1, n_points=POINTS, seed=0, distill=True, device=DEV)
    ex = frames.example()
    student = fill_params(build_detector(waymo_configs.s2d_student()), seed=2)
    return frames, ex, student


def test_voxels_of_a_150k_point_frame_match_the_c_oracle(setup):
    frames, ex, _ = setup
    v, c, n = OV.points_to_voxel(frames.points[0].cpu().numpy(), scene.WAYMO_VOXEL, scene.WAYMO_RANGE, 5, 150000)
    assert c.shape[0] > 50000   # the benchmark's per-frame voxel count (~65-72 k)
    assert np.array_equal(ex["coordinates"][:, 1:].cpu().numpy(), c) and np.array_equal(ex["num_points"].cpu().numpy(), n)
    assert np.array_equal(ex["voxels"].cpu().numpy(), v)


def test_the_eight_rulebooks_of_a_150k_point_frame_match_the_oracle(setup):
    _, ex, student = setup
    bb = student.backbone
    shape = tuple(int(s) for s in (np.array(ex["shape"][0][::-1]) + [1, 0, 0]))
    strided, subm = bb._specs()
    coors = ex["coordinates"]
    got = backbones.build_geometry(coors, 1, shape, strided, subm)
    with cpu_backend.oracle_stack(ODEV):
        ref = backbones.build_geometry(cpu_backend.to_device(coors, ODEV), 1, shape, strided, subm)
    assert set(got) == set(ref) and len(got) == 8
    for key in ref:
        a, b = got[key], ref[key]
        assert (a.subm, a.kvol, a.n_in, a.n_out, tuple(a.out_shape)) == (b.subm, b.kvol, b.n_in, b.n_out, tuple(b.out_shape)), key
        assert torch.equal(a.pair_count.cpu(), b.pair_count.cpu()), key
        assert torch.equal(a.nbr_out[:, : a.n_out].cpu(), b.nbr_out.cpu()), key
        if not b.subm:
            assert torch.equal(a.nbr_in[:, : a.n_in].cpu(), b.nbr_in.cpu()), key
            assert torch.equal(a.out_coors.reshape(-1, 4)[: a.n_out].cpu(), b.out_coors.reshape(-1, 4).cpu()), key
    print("rulebooks bit-exact:", {str(k): (int(v.n_in), int(v.n_out), int(v.pair_count.sum())) for k, v in ref.items()})


def _step(model, ex):
    grabbed = {}
    hook = model.backbone.register_forward_hook(lambda m, i, o: grabbed.update(bev=o[0].detach(), ms=o[1]))
    losses, F_S_a, F_S_b, preds, mask_loss, offset_loss = model(ex, return_loss=True, return_feature=True)
    hook.remove()
    terms = dict(det=sum(losses["loss"]), hm=sum(losses["hm_loss"]), loc=sum(losses["loc_loss"]), mask=mask_loss, offset=offset_loss)
    total = terms["det"] + mask_loss + offset_loss
    total.backward()
    grads = {n: p.grad.detach().double().cpu() for n, p in model.named_parameters() if p.grad is not None}
    return ({k: float(v) for k, v in terms.items()}, grabbed["bev"].double().cpu(), {k: np.asarray(t.indices.cpu() if torch.is_tensor(t.indices) else t.indices)
                                                                                      for k, t in grabbed["ms"].items()},
            F_S_a.detach().double().cpu(), grads)


def test_s2d_student_step_on_a_150k_point_frame_fp32_mode_vs_the_oracle_stack(setup):
    _, ex, student = setup
    with cpu_backend.oracle_stack(ODEV):
        t_ref, bev_ref, idx_ref, fsa_ref, g_ref = _step(copy.deepcopy(student).double().to(ODEV).train(), cpu_backend.to_device(ex, ODEV, torch.float64))
    t, bev, idx, fsa, g = _step(copy.deepcopy(student).to(DEV).train(), ex)
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-300))
    print("150 k points, fp32 HIP vs float64 oracle stack: losses", t, t_ref, "BEV", rel(bev, bev_ref), "F_S_a", rel(fsa, fsa_ref))
    for k in idx_ref:
        assert np.array_equal(idx[k], idx_ref[k]), k
    assert rel(bev, bev_ref) <= 5e-3
    assert rel(fsa, fsa_ref) <= 5e-3
    for k in t_ref:
        assert abs(t[k] - t_ref[k]) <= 1e-3 * abs(t_ref[k]) + 1e-7, (k, t[k], t_ref[k])
    assert set(g) == set(g_ref)
    n_hip = float(torch.sqrt(sum((v ** 2).sum() for v in g.values())))
    n_ref = float(torch.sqrt(sum((v ** 2).sum() for v in g_ref.values())))
    assert abs(n_hip - n_ref) <= 2e-2 * n_ref, (n_hip, n_ref)
    errs = {n: rel(g[n], g_ref[n]) for n in g_ref if float(g_ref[n].norm()) > 1e-6 * n_ref}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    print("worst gradient errors:", [(n, f"{e:.1e}") for n, e in worst], "median", sorted(errs.values())[len(errs) // 2])
    assert sorted(errs.values())[len(errs) // 2] <= 2e-2, worst
    assert max(errs.values()) <= 1.5e-1, worst   # (train-mode batch norms through 21 sparse + ~40 dense layers, fp32 vs exact)
