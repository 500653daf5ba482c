"""Packed weight images after the fused optimizer step (dense2d.refresh_pack_cache): in-place re-launch of the registered pack kernels,
replayed as ONE HIP graph once the launch set is stable, against the r02 behaviour (drop every image, rebuild at the next use).
The packs are the same kernels on the same operands, so a training run must not depend on the mode (the pack
kernels inside the graph, with a with / without loss-equality test; ADVICE r01: stale images after an update)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(mode, monkeypatch, steps=7):
    from sparse2dense_amd import dense2d, hip_ops, waymo_configs
    from sparse2dense_amd.data import SyntheticFrames
    from sparse2dense_amd.registry import build_detector
    from sparse2dense_amd.solver import build_one_cycle_optimizer, build_one_cycle_scheduler
    from sparse2dense_amd.train_step import backward_and_step
    monkeypatch.setenv("S2D_PACK_GRAPH", mode)
    dense2d.clear_pack_cache()
    hip_ops.set_sparse_compute_dtype("s16")
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    model = build_detector(waymo_configs.s2d_student())
    model.dense_dtype = torch.bfloat16
    model.use_channels_last()
    model = model.to(dev).train()
    frames = SyntheticFrames(1, n_points=12000, seed=5, distill=True, device=dev)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = build_one_cycle_optimizer(model, dict(wd=0.01))
    sch = build_one_cycle_scheduler(opt, dict(type="one_cycle", lr_max=0.003, moms=[0.95, 0.85], div_factor=10.0, pct_start=0.4), total_steps=100)
    losses = []
    for it in range(steps):
        out = model(frames.example(), return_loss=True, return_feature=True)
        loss = sum(out[0]["loss"]) + out[4] + out[5]
        backward_and_step(loss, params, opt, sch, it, 35.0)
        losses.append(float(loss))
    graph = dense2d._repack_state["graph"] is not None
    final = torch.cat([p.detach().flatten()[:64].double().cpu() for p in params])
    hip_ops.set_sparse_compute_dtype("f32")
    dense2d.clear_pack_cache()
    return losses, final, graph


def test_training_run_is_independent_of_the_pack_refresh_mode(monkeypatch):
    ref_losses, ref_final, ref_graph = _run("off", monkeypatch)
    assert not ref_graph
    for mode, want_graph in (("0", False), ("1", True)):
        losses, final, graph = _run(mode, monkeypatch)
        assert graph == want_graph, (mode, graph)            # the graph was actually captured (and replayed for the last steps)
        assert losses == ref_losses, (mode, losses, ref_losses)   # same kernels, same operands: bit-identical losses
        assert torch.equal(final, ref_final)
    assert ref_losses[-1] != ref_losses[0]                   # the parameters did move


def test_frozen_parameters_keep_their_images_and_stale_ones_are_dropped(monkeypatch):
    from sparse2dense_amd import dense2d as D
    monkeypatch.setenv("S2D_PACK_GRAPH", "0")
    D.clear_pack_cache()
    dev = torch.device("cuda:0")
    a, b = D.Conv3x3(64, 64, 3, 1, 1).to(dev), D.Conv3x3(64, 64, 3, 1, 1).to(dev)
    for p in b.parameters():
        p.requires_grad = False
    x = torch.randn(1, 64, 9, 9, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ya, yb = a(x).clone(), b(x).clone()
    pa, pb = D.pack_weights(a.weight).clone(), D.pack_weights(b.weight).clone()
    with torch.no_grad():   # an update through the raw storage, as the fused Adam does (version counters do not move)
        a.weight.data.mul_(2.0)
        b.weight.data.mul_(2.0)
    D.refresh_pack_cache()
    assert torch.equal(D.pack_weights(a.weight).float(), pa.float() * 2)    # trainable: re-packed in place from the new values
    assert torch.equal(D.pack_weights(b.weight), pb)                        # frozen: image kept (the caller promised not to touch it)
    D.clear_pack_cache()
    assert torch.equal(D.pack_weights(b.weight).float(), pb.float() * 2)
    D.clear_pack_cache()


def test_eval_batchnorm_affine_cache_of_frozen_layers_tracks_its_inputs():
    """dense2d._BNRowFn caches [mean, invstd, scale, shift] of an eval-mode batch norm whose affine parameters are frozen (the
    distillation teacher): the cache must follow load_state_dict / copy_ (version counters) and training-mode statistics updates made
    through raw pointers (_s2d_stats_epoch), and must not be used for trainable layers"""
    from sparse2dense_amd import dense2d as D
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    bn = D.FastBatchNorm2d(64).to(dev)
    x = torch.randn(2, 64, 5, 7, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def ref():
        return torch.nn.functional.batch_norm(x.float(), bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)

    def check():
        y = bn(x)
        assert float((y.float() - ref()).abs().max()) <= 2e-2 * float(ref().abs().max())
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-1, 1); bn.running_mean.uniform_(-1, 1); bn.running_var.uniform_(0.5, 2.0)
    bn.eval()
    check()                                   # trainable: never cached
    assert getattr(bn, "_s2d_eval_fin", None) is None
    for p in bn.parameters():
        p.requires_grad = False
    check(); check()
    assert bn._s2d_eval_fin is not None
    with torch.no_grad():
        bn.running_mean.add_(0.7)             # version bump
    check()
    bn.load_state_dict({k: v * 1.25 if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    check()
    bn.train(); bn(x); bn.eval()              # running statistics moved by the finalize kernel (raw pointers)
    check()
