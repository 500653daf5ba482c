"""data.PrefetchLoader: the data pipeline of the step (device voxelization, AssignLabel, rulebooks) on a second stream / worker thread.
What it hands out must be exactly what the in-step path builds, and a training step fed by it must give the same numbers."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import fill_params
from sparse2dense_amd import waymo_configs
from sparse2dense_amd.data import PrefetchLoader, SyntheticFrames, attach_geometry
from sparse2dense_amd.registry import build_detector
from sparse2dense_amd.train_step import distill_loss

DEV = "cuda:0"


def _same(a, b, path=""):
    if torch.is_tensor(a):
        assert torch.is_tensor(b) and a.shape == b.shape and a.dtype == b.dtype, path
        assert torch.equal(a, b), path
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, f"{path}[{i}]")
    elif isinstance(a, dict):
        assert a.keys() == b.keys(), path
        for k in a:
            _same(a[k], b[k], f"{path}.{k}")
    else:
        assert (a == b) if not hasattr(a, "shape") else (a == b).all(), path


def test_prefetched_examples_and_rulebooks_equal_the_in_step_ones():
    student = build_detector(waymo_configs.s2d_student()).to(DEV)
    frames = SyntheticFrames(2, n_points=20000, seed=5, distill=True, device=DEV)
    direct = attach_geometry(frames.example(), student.backbone)
    loader = PrefetchLoader(SyntheticFrames(2, n_points=20000, seed=5, distill=True, device=DEV), backbone=student.backbone)
    try:
        for _ in range(3):   # the resident scenes are the same every iteration
            ex = loader.example()
            torch.cuda.synchronize()
            assert ex.keys() == direct.keys()
            for k in direct:
                _same(direct[k], ex[k], k)
            for key in ("coordinates", "dense_coordinates", "reconstruction_coordinates"):
                shape_a, batch_a, plan_a = direct[key]._s2d_geometry
                shape_b, batch_b, plan_b = ex[key]._s2d_geometry
                assert shape_a == shape_b and batch_a == batch_b and plan_a.keys() == plan_b.keys()
                for name in plan_a:
                    ra, rb = plan_a[name], plan_b[name]
                    assert (ra.n_in, ra.n_out, ra.kvol, ra.subm) == (rb.n_in, rb.n_out, rb.kvol, rb.subm)
                    _same(ra.nbr_out, rb.nbr_out, f"{key}.{name}.nbr_out")
                    _same(ra.pair_count, rb.pair_count, f"{key}.{name}.pair_count")
                    if ra.out_coors is not None:
                        _same(ra.out_coors, rb.out_coors, f"{key}.{name}.out_coors")
    finally:
        loader.close()


def test_distillation_step_fed_by_the_loader_matches_the_in_step_pipeline():
    torch.manual_seed(0)
    teacher = fill_params(build_detector(waymo_configs.centerpoint_voxelnet()), seed=1).to(DEV)
    student = fill_params(build_detector(waymo_configs.s2d_student()), seed=2).to(DEV).train()
    for p in teacher.parameters():
        p.requires_grad = False

    def run(example_fn, n):
        out = []
        for _ in range(n):
            student.zero_grad()
            total, _ = distill_loss(teacher, student, example_fn())
            total.backward()
            g = torch.stack([p.grad.float().norm() for p in student.parameters() if p.grad is not None]).norm()
            out.append((float(total), float(g)))
        return out

    frames = SyntheticFrames(1, n_points=12000, seed=31, distill=True, device=DEV)
    ref = run(frames.example, 2)
    loader = PrefetchLoader(SyntheticFrames(1, n_points=12000, seed=31, distill=True, device=DEV), backbone=student.backbone)
    try:
        got = run(loader.example, 3)
    finally:
        loader.close()
    # batch-norm running statistics do not enter the training-mode forward: every iteration sees the same scene and weights
    for a in got:
        assert abs(a[0] - ref[0][0]) <= 1e-5 * abs(ref[0][0]) and abs(a[1] - ref[0][1]) <= 1e-4 * ref[0][1], (a, ref)


def test_loader_surfaces_worker_errors():
    class Broken:
        device = torch.device(DEV)

        def example(self):
            raise ValueError("boom")

    loader = PrefetchLoader(Broken())
    with pytest.raises(ValueError, match="boom"):
        loader.example()
    loader.close()
