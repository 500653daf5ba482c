"""The shape-static dense segment replayed as HIP graphs (sparse2dense_amd/graphed.py, `detector.use_hip_graphs()`): the same kernels on the
same operands in the same order as the kernel-by-kernel run, so a training run must not depend on the mode - bit-identical losses,
parameters and gradients - whether the segment is the S2D student's (S2D module + PCR head with padded recon-voxel lists + RPN trunk +
CenterHead + losses), plain CenterPoint's, or the frozen distillation teacher's forward-only graph.
Reference loop: /root/reference/det3d/torchie/trainer/trainer.py:775-811, hooks/optimizer.py:15-21."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(kind, dev):
    from sparse2dense_amd import waymo_configs
    from sparse2dense_amd.registry import build_detector
    m = build_detector(getattr(waymo_configs, kind)())
    m.dense_dtype = torch.bfloat16
    m.use_channels_last()
    return m.to(dev)


def _run(graph, steps=6, n_points=12000, batch=2, kind="s2d_student", vary=False, fork="", defer="dense,aux,pcr", split="0"):
    from sparse2dense_amd import dense2d, graphed, hip_ops, side
    from sparse2dense_amd.data import SyntheticFrames
    from sparse2dense_amd.solver import build_one_cycle_optimizer, build_one_cycle_scheduler
    from sparse2dense_amd.train_step import backward_and_step
    import os
    os.environ["S2D_GRAPH_SPLIT"] = split   # "1": the S2D student's dense part as two graphed segments in sequence (detectors.KD_VoxelNet._dense_call)
    side.enable(False)
    side.graph_fork(fork if fork else False)
    side.graph_defer(defer if defer else False)
    side.stats["deferred"] = 0
    dense2d.clear_pack_cache()
    hip_ops.set_sparse_compute_dtype("s16")
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    model = _model(kind, dev).train()
    if graph:
        model.use_hip_graphs()
    distill = kind == "s2d_student"
    sets = [SyntheticFrames(batch, n_points=n_points + 1500 * k, seed=5 + 10 * k, distill=distill, device=dev) for k in range(3 if vary else 1)]
    params = [p for p in model.parameters() if p.requires_grad]
    opt = build_one_cycle_optimizer(model, dict(wd=0.01))
    sch = build_one_cycle_scheduler(opt, dict(type="one_cycle", lr_max=0.003, moms=[0.95, 0.85], div_factor=10.0, pct_start=0.4), total_steps=100)
    for k in graphed.stats:
        graphed.stats[k] = 0
    losses = []
    try:
        for it in range(steps):
            ex = sets[it % len(sets)].example()
            if distill:
                out = model(ex, return_loss=True, return_feature=True)
                loss = sum(out[0]["loss"]) + out[4] + out[5]
            else:
                loss = sum(model(ex, return_loss=True)["loss"])
            if it == steps - 1:   # last step by hand: the gradients themselves are compared
                for p in params:
                    p.grad = None
                loss.backward()
                torch.cuda.synchronize()
                grads = [None if p.grad is None else p.grad.detach().clone() for p in params]
            else:
                backward_and_step(loss, params, opt, sch, it, 35.0)
            losses.append(float(loss))
        final = torch.cat([p.detach().flatten()[:64].double().cpu() for p in params])
        bn = torch.cat([b.detach().flatten()[:16].double().cpu() for b in model.buffers()])
        st = dict(graphed.stats)
    finally:
        os.environ.pop("S2D_GRAPH_SPLIT", None)
        side.graph_fork(False)
        side.graph_defer("dense,aux,pcr")
        hip_ops.set_sparse_compute_dtype("f32")
        dense2d.clear_pack_cache()
    return losses, final, grads, bn, st


_REF_RUNS = {}


def _eager_ref(kind="s2d_student"):
    """the kernel-by-kernel run of the default settings, computed once per detector kind (every parametrisation compares against the same run)"""
    if kind not in _REF_RUNS:
        _REF_RUNS[kind] = _run(False, kind=kind)
        assert _REF_RUNS[kind][4]["replay"] == 0
    return _REF_RUNS[kind]


def _same(a, b, what):
    la, fa, ga, ba, _ = a
    lb, fb, gb, bb, _ = b
    assert la == lb, (what, la, lb)
    assert torch.equal(fa, fb), what
    assert torch.equal(ba, bb), what + ": batch-norm running statistics"
    for g, r in zip(ga, gb):
        assert (g is None) == (r is None), what
        if g is not None:
            assert torch.equal(g, r), what


@pytest.mark.parametrize("kind,split", [("s2d_student", "1"), ("s2d_student", "0"), ("centerpoint_voxelnet", "1")])
@pytest.mark.parametrize("defer", ["dense,aux,pcr", "dense,aux", ""])
def test_training_run_is_independent_of_the_graph_replay(kind, split, defer):
    """defer = the layer kinds whose weight gradients are captured into a second graph that is replayed on the side stream beside the eager
    sparse backward (side.GRAPH_DEFER, the default); "" = everything in the chain's graph"""
    from sparse2dense_amd import side
    ref = _eager_ref(kind)
    got = _run(True, kind=kind, defer=defer, split=split)
    segs = 2 if (kind == "s2d_student" and split == "1") else 1
    assert got[4]["capture"] == segs and got[4]["replay"] == 4 * segs, got[4]    # 2 eager warm-up calls, then capture + replays
    assert (side.stats["deferred"] > 20) == bool(defer), side.stats
    _same(got, ref, kind)
    assert ref[0][-1] != ref[0][0]


def test_training_step_declares_its_gradients_consumed_and_nothing_is_parked():
    """r06: `train_step.backward_and_step` calls `graphed.grads_consumed()` after the optimizer has read the gradients, so the next forward replay
    does not move the 164 static gradient buffers that are still bound as `.grad` to private copies (0.64 ms of copy kernels + ~1.6 ms of launch-
    thread time per step in the graphed mode before) - and the run stays bit-equal to the kernel-by-kernel one.  Without the declaration (the
    accumulation test below) they are parked."""
    got = _run(True)
    _same(got, _eager_ref(), "graphed training steps with consumed gradients")
    assert got[4]["replay"] > 0 and got[4]["parked"] == 0, got[4]


def test_weight_gradients_as_branches_of_the_backward_graph():
    """side.graph_fork("dense,aux"): inside the capture the weight-gradient launch groups fork onto a second capturing stream and rejoin
    at the end - parallel branches of the replayed backward graph; same kernels, same operands: bit-equal to the kernel-by-kernel run"""
    from sparse2dense_amd import side
    ref = _eager_ref()
    side.stats["forked"] = 0
    got = _run(True, fork="dense,aux", defer="", split="0")
    assert side.stats["forked"] > 20, side.stats
    assert got[4]["capture"] == 1 and got[4]["replay"] == 4, got[4]
    _same(got, ref, "forked weight gradients")


def test_graph_replay_with_recon_lists_of_changing_length():
    """three point clouds of different sizes in rotation: the sparse stack is eager and re-planned per cloud, the PCR head's recon-voxel lists
    shrink and grow inside their padded buffers, the dense graphs are captured once"""
    ref = _run(False, steps=8, vary=True)
    got = _run(True, steps=8, vary=True)
    assert got[4]["capture"] == 1 and got[4]["replay"] == 6, got[4]
    _same(got, ref, "varying clouds")


def test_parameter_surgery_drops_the_capture():
    """load_state_dict (version counters move) behind a captured segment: the next call must not replay a graph that reads stale packed
    weight images - it re-runs eagerly and captures again"""
    from sparse2dense_amd import dense2d, graphed, hip_ops
    from sparse2dense_amd.data import SyntheticFrames
    hip_ops.set_sparse_compute_dtype("s16")
    dense2d.clear_pack_cache()
    dev = torch.device("cuda:0")
    try:
        torch.manual_seed(3)
        model = _model("centerpoint_voxelnet", dev).train().use_hip_graphs()
        twin = _model("centerpoint_voxelnet", dev).train()
        frames = SyntheticFrames(1, n_points=9000, seed=2, device=dev)
        for k in graphed.stats:
            graphed.stats[k] = 0

        def loss_of(m):
            return float(sum(m(frames.example(), return_loss=True)["loss"]))
        for _ in range(4):
            loss_of(model)
        assert graphed.stats["capture"] == 1 and graphed.stats["replay"] == 2
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        torch.manual_seed(99)
        for p in model.parameters():
            sd_key = [k for k, v in model.named_parameters() if v is p][0]
            sd[sd_key] = sd[sd_key] + 0.01 * torch.randn_like(p)
        model.load_state_dict(sd)
        twin.load_state_dict(sd)
        a = loss_of(model)
        assert graphed.stats["dropped"] == 1
        twin.load_state_dict(model.state_dict())   # (running statistics moved by the calls above)
        b = loss_of(twin)
        assert a == b, (a, b)
    finally:
        hip_ops.set_sparse_compute_dtype("f32")
        dense2d.clear_pack_cache()


def _run_distill(graph, steps=6, n_points=12000, batch=2):
    """the full distillation step (trainer.py:775-811): frozen teacher forward (a forward-only graph under no_grad) + student step whose
    feature maps and predictions ALSO receive gradients (sparse2dense_loss, kd_hm, kd_reg on top of the detection and PCR losses)"""
    from sparse2dense_amd import dense2d, graphed, hip_ops, side
    from sparse2dense_amd.data import SyntheticFrames
    from sparse2dense_amd.solver import build_one_cycle_optimizer, build_one_cycle_scheduler
    from sparse2dense_amd.train_step import backward_and_step, distill_loss
    side.enable(False)
    side.graph_defer("dense,aux,pcr")
    dense2d.clear_pack_cache()
    hip_ops.set_sparse_compute_dtype("s16")
    dev = torch.device("cuda:0")
    torch.manual_seed(21)
    student = _model("s2d_student", dev).train()
    teacher = _model("centerpoint_voxelnet", dev).eval()
    for p in teacher.parameters():
        p.requires_grad = False
    if graph:
        student.use_hip_graphs()
        teacher.use_hip_graphs()
    frames = SyntheticFrames(batch, n_points=n_points, seed=8, distill=True, device=dev)
    params = [p for p in student.parameters() if p.requires_grad]
    opt = build_one_cycle_optimizer(student, dict(wd=0.01))
    sch = build_one_cycle_scheduler(opt, dict(type="one_cycle", lr_max=0.003, moms=[0.95, 0.85], div_factor=10.0, pct_start=0.4), total_steps=100)
    for k in graphed.stats:
        graphed.stats[k] = 0
    out = []
    try:
        for it in range(steps):
            loss, terms = distill_loss(teacher, student, frames.example())
            out.append((float(loss.detach()), float(terms["sparse2dense_loss"][0]), float(terms["kd_hm_loss"][0]), float(terms["kd_reg_loss"][0])))
            if it == steps - 1:
                for p in params:
                    p.grad = None
                loss.backward()
                torch.cuda.synchronize()
                grads = [None if p.grad is None else p.grad.detach().clone() for p in params]
            else:
                backward_and_step(loss, params, opt, sch, it, 35.0)
        final = torch.cat([p.detach().flatten()[:64].double().cpu() for p in params])
        st = dict(graphed.stats)
    finally:
        hip_ops.set_sparse_compute_dtype("f32")
        dense2d.clear_pack_cache()
    return out, final, grads, st


def test_distillation_step_is_independent_of_the_graph_replay():
    ref_out, ref_final, ref_grads, st0 = _run_distill(False)
    got_out, got_final, got_grads, st = _run_distill(True)
    assert st0["replay"] == 0 and st["capture"] == 2 and st["replay"] == 8, st   # teacher (forward only) + student segments, 4 replays each
    assert got_out == ref_out, (got_out, ref_out)
    assert torch.equal(got_final, ref_final)
    for g, r in zip(got_grads, ref_grads):
        assert (g is None) == (r is None)
        if g is not None:
            assert torch.equal(g, r)
    assert ref_out[-1][0] != ref_out[0][0] and ref_out[0][1] > 0 and ref_out[0][2] > 0


def test_graph_replay_with_the_batch_norm_collectives_captured_inside_on_one_rank(monkeypatch):
    """the N>1 route at world_size 1 over RCCL (dp.wrap_ddp -> gradient buckets, self-synchronising batch norms on the library's own RCCL
    communicator, collective.init_direct) with S2D_DENSE_GRAPH_SYNCBN=1: every batch-norm all-reduce of the dense segment is a node of the
    forward / backward HIP graph (capture mode thread_local: c10d's watchdog thread polls its events while the capture runs), the gradient
    buckets fire from the hooks `_Replay.backward` runs - and the run equals the kernel-by-kernel one bit for bit.
    Reference: /root/reference/det3d/torchie/apis/train.py:360-391 (SyncBN + DistributedDataParallel)."""
    import torch.distributed as dist
    from sparse2dense_amd import _lib, collective, dense2d, dp, graphed, hip_ops, side, waymo_configs
    from sparse2dense_amd.data import SyntheticFrames
    from sparse2dense_amd.registry import build_detector
    from sparse2dense_amd.solver import build_one_cycle_optimizer, build_one_cycle_scheduler
    from sparse2dense_amd.train_step import backward_and_step
    if dist.is_initialized():
        pytest.skip("a process group is already up")
    monkeypatch.setenv("S2D_FORCE_DDP", "1")
    monkeypatch.setenv("S2D_BUCKET_MB", "4")
    monkeypatch.setenv("S2D_DENSE_GRAPH_SYNCBN", "1")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29593", rank=0, world_size=1)
    dev = torch.device("cuda:0")

    def run(graph, steps=6):
        side.enable(False)
        side.graph_defer("dense,aux,pcr")
        dense2d.clear_pack_cache()
        hip_ops.set_sparse_compute_dtype("s16")
        torch.manual_seed(11)
        model = build_detector(waymo_configs.s2d_student())
        model.dense_dtype = torch.bfloat16
        model.use_channels_last()
        model = dp.wrap_ddp(model.to(dev).train(), 0)
        assert collective.sync_on() and collective.direct_enabled()
        if graph:
            getattr(model, "module", model).use_hip_graphs()
        frames = SyntheticFrames(2, n_points=12000, seed=5, distill=True, device=dev)
        params = [p for p in model.parameters() if p.requires_grad]
        opt = build_one_cycle_optimizer(model, dict(wd=0.01))
        sch = build_one_cycle_scheduler(opt, dict(type="one_cycle", lr_max=0.003, moms=[0.95, 0.85], div_factor=10.0, pct_start=0.4), total_steps=100)
        for k in graphed.stats:
            graphed.stats[k] = 0
        losses = []
        for it in range(steps):
            out = model(frames.example(), return_loss=True, return_feature=True)
            loss = sum(out[0]["loss"]) + out[4] + out[5]
            backward_and_step(loss, params, opt, sch, it, 35.0)
            losses.append(float(loss.detach()))
        final = torch.cat([p.detach().flatten()[:64].double().cpu() for p in params])
        model._s2d_grad_buckets.remove()
        return losses, final, dict(graphed.stats)
    try:
        ref = run(False)
        got = run(True)
        assert ref[2]["replay"] == 0 and got[2]["capture"] == 1 and got[2]["replay"] == 4, (ref[2], got[2])
        assert got[0] == ref[0], (got[0], ref[0])
        assert torch.equal(got[1], ref[1])
        assert ref[0][-1] != ref[0][0]
    finally:
        hip_ops.set_sparse_compute_dtype("f32")
        dense2d.clear_pack_cache()
        collective._DIRECT = False
        _lib.load().s2d_comm_shutdown()
        dist.destroy_process_group()


def _eval_between_training_steps(graph):
    """train 2 steps -> eval x 4 (graph: 2 eager warm-up calls, capture, replay) -> train 1 step -> eval x 2 (replays): the evaluation outputs"""
    from sparse2dense_amd import dense2d, graphed, hip_ops, side
    from sparse2dense_amd.data import SyntheticFrames
    from sparse2dense_amd.solver import build_one_cycle_optimizer, build_one_cycle_scheduler
    from sparse2dense_amd.train_step import backward_and_step
    side.enable(False)
    dense2d.clear_pack_cache()
    hip_ops.set_sparse_compute_dtype("s16")
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    model = _model("s2d_student", dev).train()
    if graph:
        model.use_hip_graphs()
    frames = SyntheticFrames(2, n_points=12000, seed=5, distill=True, device=dev)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = build_one_cycle_optimizer(model, dict(wd=0.01))
    sch = build_one_cycle_scheduler(opt, dict(type="one_cycle", lr_max=0.003, moms=[0.95, 0.85], div_factor=10.0, pct_start=0.4), total_steps=100)
    for k in graphed.stats:
        graphed.stats[k] = 0
    outs, it = [], 0

    def train(n):
        nonlocal it
        model.train()
        for _ in range(n):
            out = model(frames.example(), return_loss=True, return_feature=True)
            backward_and_step(sum(out[0]["loss"]) + out[4] + out[5], params, opt, sch, it, 35.0)
            it += 1

    def evaluate(n):
        model.eval()
        with torch.no_grad():
            for _ in range(n):
                out = model(frames.example(), return_loss=True, return_feature=True)   # eval mode: no PCR branch, forward-only segment
                outs.append((out[1].float().clone(), out[2].float().clone()))
    try:
        train(2)
        evaluate(4)
        train(1)
        evaluate(2)
        torch.cuda.synchronize()
        st = dict(graphed.stats)
    finally:
        hip_ops.set_sparse_compute_dtype("f32")
        dense2d.clear_pack_cache()
    return outs, st


def test_forward_only_capture_with_a_warm_pack_cache_survives_optimizer_steps():
    """an evaluation capture taken with a WARM packed-weight cache baked in the addresses of images that have no in-place refresh
    (layer-norm / 1x1 matrices); the next optimizer step freed them (the fused Adam writes through raw pointers: the capture's staleness check
    sees nothing) and the replay read freed memory.  The capture now builds such images inside itself (dense2d.CAPTURE_PACKS)."""
    ref, st0 = _eval_between_training_steps(False)
    got, st = _eval_between_training_steps(True)
    assert st0["replay"] == 0 and st["replay"] >= 4 and st["capture"] >= 2, st
    for k, ((a1, b1), (a0, b0)) in enumerate(zip(got, ref)):
        assert torch.equal(a1, a0) and torch.equal(b1, b0), f"evaluation call {k} differs from the kernel-by-kernel run"
    assert not torch.equal(ref[3][0], ref[5][0])   # the training step in between did move the weights


def _two_backwards(graph):
    from sparse2dense_amd import dense2d, graphed, hip_ops, side
    from sparse2dense_amd.data import SyntheticFrames
    side.enable(False)
    dense2d.clear_pack_cache()
    hip_ops.set_sparse_compute_dtype("s16")
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    model = _model("s2d_student", dev).train()
    if graph:
        model.use_hip_graphs()
    sets = [SyntheticFrames(2, n_points=12000, seed=5 + 10 * k, distill=True, device=dev) for k in range(2)]
    params = [p for p in model.parameters() if p.requires_grad]

    def loss_of(k):
        out = model(sets[k].example(), return_loss=True, return_feature=True)
        return sum(out[0]["loss"]) + out[4] + out[5]
    singles = []
    try:
        for _ in range(3):   # warm-up calls + capture, one backward each
            for p in params:
                p.grad = None
            loss_of(0).backward()
        for k in (0, 1):     # each micro-batch on its own (diagnostics: which sum did a wrong accumulation produce?)
            for p in params:
                p.grad = None
            loss_of(k).backward()
            side.join()
            torch.cuda.synchronize()
            singles.append([None if p.grad is None else p.grad.detach().clone() for p in params])
        for p in params:
            p.grad = None
        loss_of(0).backward()
        loss_of(1).backward()          # second micro-batch: accumulates onto the first
        side.join()
        torch.cuda.synchronize()
        grads = [None if p.grad is None else p.grad.detach().clone() for p in params]
        for p in params:               # ... and the zero_grad(set_to_none=False) pattern: .grad stays bound and is zeroed in place
            if p.grad is not None:
                p.grad.zero_()
        loss_of(1).backward()
        side.join()
        torch.cuda.synchronize()
        grads2 = [None if p.grad is None else p.grad.detach().clone() for p in params]
    finally:
        hip_ops.set_sparse_compute_dtype("f32")
        dense2d.clear_pack_cache()
    return grads, grads2, singles, [n for n, p in model.named_parameters() if p.requires_grad]


def test_gradient_accumulation_over_two_backward_passes_matches_the_eager_run():
    """the first backward binds p.grad to the capture's static buffer; a second backward replayed into that very buffer and then
    added it to itself (2 x the second gradient, the first lost).  Now: the bound values are copied out before the replay."""
    ref, ref2, ref_single, names = _two_backwards(False)
    got, got2, got_single, _ = _two_backwards(True)
    for k in (0, 1):   # the single passes agree (what the other graph tests hold) ...
        for n, g, r in zip(names, got_single[k], ref_single[k]):
            assert (g is None) == (r is None) and (g is None or torch.equal(g, r)), ("single backward", k, n)
    for what, pair in (("two backwards", (got, ref)), ("zeroed in place + one backward", (got2, ref2))):
        for i, (n, g, r) in enumerate(zip(names, *pair)):
            assert (g is None) == (r is None), n
            if g is not None and not torch.equal(g, r):   # eager accumulates in place (a += b), the graph path out of place (a + b): the same fp32 sum
                rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
                s0, s1 = ref_single[0][i], ref_single[1][i]
                raise AssertionError((what, n, "vs eager sum", rel(g, r), "vs 2 x second", rel(g, 2 * s1), "vs second", rel(g, s1), "vs first", rel(g, s0)))
