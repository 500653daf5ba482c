"""Weight gradients on a second HIP stream (sparse2dense_amd/side.py, S2D_WGRAD_STREAM): the same kernels on the same operands in a
different launch order, so a training run must not depend on the mode - bit-identical losses and parameters - and every gradient
produced on the side stream must be the tensor autograd adopted as `.grad` (a clone would be a main-stream launch before the join).
(tools/side_stress.py repeats such runs; it found the unordered ring initialisation of `spconv_wgrad_s16_coop128` - a latent race of
that kernel that only a second stream's kernels competing for wave slots exposed - and a rare mismatch of the opt-in PCR-branch stream.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(mode, steps=5, n_points=12000, pcr_stream="0"):
    import os
    from sparse2dense_amd import dense2d, hip_ops, side, waymo_configs
    from sparse2dense_amd.data import SyntheticFrames
    from sparse2dense_amd.registry import build_detector
    from sparse2dense_amd.solver import build_one_cycle_optimizer, build_one_cycle_scheduler
    from sparse2dense_amd.train_step import backward_and_step
    old_pcr = os.environ.get("S2D_PCR_STREAM")
    os.environ["S2D_PCR_STREAM"] = pcr_stream   # necks.S2D_RPN: the PCR branch on its own stream (forward and, through autograd, backward)
    side.enable(mode)
    side.CHECK = True
    side._handed.clear()
    side.stats.update(side=0, plain=0, waited=0)
    dense2d.clear_pack_cache()
    hip_ops.set_sparse_compute_dtype("s16")
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    model = build_detector(waymo_configs.s2d_student())
    model.dense_dtype = torch.bfloat16
    model.use_channels_last()
    model = model.to(dev).train()
    frames = SyntheticFrames(1, n_points=n_points, seed=5, distill=True, device=dev)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = build_one_cycle_optimizer(model, dict(wd=0.01))
    sch = build_one_cycle_scheduler(opt, dict(type="one_cycle", lr_max=0.003, moms=[0.95, 0.85], div_factor=10.0, pct_start=0.4), total_steps=100)
    losses, adopted = [], []
    try:
        for it in range(steps):
            out = model(frames.example(), return_loss=True, return_feature=True)
            loss = sum(out[0]["loss"]) + out[4] + out[5]
            if it == steps - 1:   # last step by hand: the gradients themselves are compared
                for p in params:
                    p.grad = None
                loss.backward()
                adopted.append(side.adopted())
                torch.cuda.synchronize()
                grads = [None if p.grad is None else p.grad.detach().clone() for p in params]
            else:
                backward_and_step(loss, params, opt, sch, it, 35.0)
                adopted.append(side.adopted())
            losses.append(float(loss))
        final = torch.cat([p.detach().flatten()[:64].double().cpu() for p in params])
        stats = dict(side.stats)
    finally:
        side.enable(False)
        side.CHECK = False
        if old_pcr is None:
            os.environ.pop("S2D_PCR_STREAM", None)
        else:
            os.environ["S2D_PCR_STREAM"] = old_pcr
        hip_ops.set_sparse_compute_dtype("f32")
        dense2d.clear_pack_cache()
    return losses, final, grads, adopted, stats


def test_training_run_is_independent_of_the_weight_gradient_stream():
    ref_losses, ref_final, ref_grads, _, ref_stats = _run("0")
    assert ref_stats["side"] == 0
    for mode, pcr in (("dense", "0"), ("sparse", "0"), ("1", "0")):
        losses, final, grads, adopted, stats = _run(mode, pcr_stream=pcr)
        assert stats["side"] > 0, (mode, stats)                       # the side stream was used ...
        assert all(ok and n > 0 for ok, n in adopted), (mode, adopted, stats)   # ... and autograd adopted every gradient it produced
        assert losses == ref_losses, (mode, losses, ref_losses)
        assert torch.equal(final, ref_final), mode
        for g, r in zip(grads, ref_grads):
            assert (g is None) == (r is None)
            if g is not None:
                assert torch.equal(g, r), mode
    assert ref_losses[-1] != ref_losses[0]


def test_accumulating_into_an_existing_gradient_takes_the_plain_path():
    """`.grad` already set (gradient accumulation over micro-batches): AccumulateGrad adds on the main stream, so the layer must not go
    to the side stream; the sum equals two plain backward passes"""
    from sparse2dense_amd import dense2d as D, side
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    conv = D.Conv3x3(64, 64, 3, 1, 1).to(dev)
    x = torch.randn(2, 64, 24, 24, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()

    def two_passes():
        conv.weight.grad = conv.bias.grad = None
        for _ in range(2):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                conv(x).float().square().sum().backward()
        torch.cuda.synchronize()
        return conv.weight.grad.clone(), conv.bias.grad.clone()
    ref = two_passes()
    side.enable("1")
    side.stats.update(side=0, plain=0)
    try:
        got = two_passes()
        assert side.stats["side"] == 1 and side.stats["plain"] == 1, side.stats
    finally:
        side.enable(False)
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])


def test_side_stream_under_the_gradient_buckets_on_one_rank(monkeypatch):
    """world_size 1 over RCCL with the N>1 machinery forced on (dp.wrap_ddp -> GradBuckets hooks, self-synchronising batch norms): the
    weight gradients still go to the side stream - the bucket launch joins before it copies them - and the run equals the plain one"""
    import torch.distributed as dist
    from sparse2dense_amd import _lib, collective, dense2d, dp, hip_ops, side, waymo_configs
    from sparse2dense_amd.data import SyntheticFrames
    from sparse2dense_amd.registry import build_detector
    from sparse2dense_amd.solver import build_one_cycle_optimizer, build_one_cycle_scheduler
    from sparse2dense_amd.train_step import backward_and_step
    if dist.is_initialized():
        pytest.skip("a process group is already up")
    monkeypatch.setenv("S2D_FORCE_DDP", "1")
    monkeypatch.setenv("S2D_BUCKET_MB", "4")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29591", rank=0, world_size=1)
    dev = torch.device("cuda:0")

    def run(mode, steps=4):
        side.enable(mode)
        side.stats.update(side=0, plain=0)
        dense2d.clear_pack_cache()
        hip_ops.set_sparse_compute_dtype("s16")
        torch.manual_seed(11)
        model = build_detector(waymo_configs.centerpoint_voxelnet())
        model.dense_dtype = torch.bfloat16
        model.use_channels_last()
        model = dp.wrap_ddp(model.to(dev).train(), 0)
        assert getattr(model, "_s2d_grad_buckets", None) is not None and len(model._s2d_grad_buckets.buckets) > 1
        frames = SyntheticFrames(1, n_points=12000, seed=5, distill=False, device=dev)
        params = [p for p in model.parameters() if p.requires_grad]
        opt = build_one_cycle_optimizer(model, dict(wd=0.01))
        sch = build_one_cycle_scheduler(opt, dict(type="one_cycle", lr_max=0.003, moms=[0.95, 0.85], div_factor=10.0, pct_start=0.4), total_steps=100)
        losses = []
        for it in range(steps):
            loss = sum(model(frames.example(), return_loss=True)["loss"])
            backward_and_step(loss, params, opt, sch, it, 35.0)
            losses.append(float(loss.detach()))
        final = torch.cat([p.detach().flatten()[:64].double().cpu() for p in params])
        launched = list(model._s2d_grad_buckets.launch_log)
        model._s2d_grad_buckets.remove()
        return losses, final, dict(side.stats), launched
    try:
        ref = run("0")
        got = run("1")
        assert ref[2]["side"] == 0 and got[2]["side"] > 0, (ref[2], got[2])
        assert got[3] == ref[3] == sorted(ref[3])          # buckets launched in index order in both modes
        assert got[0] == ref[0], (got[0], ref[0])
        assert torch.equal(got[1], ref[1])
        assert ref[0][-1] != ref[0][0]
    finally:
        side.enable(False)
        hip_ops.set_sparse_compute_dtype("f32")
        dense2d.clear_pack_cache()
        collective._DIRECT = False
        _lib.load().s2d_comm_shutdown()
        dist.destroy_process_group()
