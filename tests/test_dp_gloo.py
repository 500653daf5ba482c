"""N>1 path on CPU: world_size-2 `gloo` job.  Each rank runs the PRODUCT module stack (HIP
launchers swapped for tests/cpu_backend.py) on its own frame under sparse2dense_amd.dp.wrap_ddp:
  * FeatureBatchNorm1d statistics are global (all-reduced sums + counts, different N per rank);
  * BatchNorm2d layers are converted to SyncBatchNorm, FeatureBatchNorm1d is left in place;
  * after the bucketed all-reduce (routes "overlap": launched from the backward hooks, "flat": after
    the backward, "ddp": torch DDP) every rank holds the gradient of the mean-of-ranks loss, equal
    to the single-process gradient on the concatenated 2-frame batch (x 1/2 for the sum loss);
  * a parameter that gets no gradient on one rank, and a rank whose sparse tensor is empty, do not
    unbalance the collectives."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _voxels(seed):
    from oracle import voxelize as OV
    from sparse2dense_amd import scene
    s = scene.make_scene(1200 + 300 * seed, seed=40 + seed, n_cars=15, n_walls=2, n_peds=4)
    v, c, n = OV.points_to_voxel(s["points"], scene.WAYMO_VOXEL, scene.WAYMO_RANGE, 5, 150000)
    return torch.from_numpy(OV.voxel_mean(v, n)).double(), c


def _build(monkeypatch=None):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_backend
    from golden_util import fill_params
    from sparse2dense_amd.registry import build_backbone
    cpu_backend.install(monkeypatch)   # worker processes patch globally; the pytest process via monkeypatch
    return fill_params(build_backbone(dict(type="SpMiddleResNetFHD", num_input_features=5))).double().train()


def _weights(shape):
    return torch.randn(shape, dtype=torch.float64, generator=torch.Generator().manual_seed(9))


def _worker(rank, world, port, out_dir, mode="ddp"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), S2D_DP_MODE=mode, S2D_BUCKET_MB="1")
    sys.path.insert(0, ROOT)
    torch.set_num_threads(2)
    from sparse2dense_amd import dp
    from sparse2dense_amd.spconv import FeatureBatchNorm1d
    r, _, w = dp.init_distributed("gloo")
    assert (r, w) == (rank, world)
    net = _build()
    probe = torch.nn.Sequential(torch.nn.Conv2d(2, 2, 1), torch.nn.BatchNorm2d(2), FeatureBatchNorm1d(4))
    conv = dp.convert_syncbn(probe)
    assert isinstance(conv[1], torch.nn.SyncBatchNorm) and type(conv[2]) is FeatureBatchNorm1d
    ddp = dp.wrap_ddp(net)
    assert isinstance(ddp, torch.nn.parallel.DistributedDataParallel) == (mode == "ddp")
    feats, c = _voxels(rank)
    coors = torch.from_numpy(np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1))
    bev, _ = ddp(feats, coors, 1, np.array([1504, 1504, 40]))
    loss = (bev * _weights((2,) + bev.shape[1:])[rank:rank + 1]).sum()
    if mode == "ddp":
        loss.backward()
    else:
        from sparse2dense_amd.train_step import backward_and_clip
        gb = net._s2d_grad_buckets
        assert gb.overlap == (mode == "overlap") and len(gb.buckets) >= 2   # S2D_BUCKET_MB below: several buckets
        backward_and_clip(loss, list(net.parameters()), max_norm=1e30)
        assert all(b["launched"] for b in gb.buckets)
    torch.save({"grads": {n: p.grad.clone() for n, p in net.named_parameters()},
                "buffers": {k: v.clone() for k, v in net.state_dict().items() if "running" in k}},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["overlap", "ddp", "flat"])
def test_two_rank_dp_equals_single_process_batch(monkeypatch, mode):
    port = 29500 + (os.getpid() % 2000) + {"overlap": 0, "ddp": 3, "flat": 7}[mode]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, port, d, mode), nprocs=2, join=True)
        r0 = torch.load(os.path.join(d, "rank0.pt"))
        r1 = torch.load(os.path.join(d, "rank1.pt"))
    # single process, both frames in one batch
    net = _build(monkeypatch)
    f0, c0 = _voxels(0)
    f1, c1 = _voxels(1)
    coors = np.concatenate([np.concatenate([np.zeros((c0.shape[0], 1), np.int32), c0], 1),
                            np.concatenate([np.ones((c1.shape[0], 1), np.int32), c1], 1)])
    bev, _ = net(torch.cat([f0, f1]), torch.from_numpy(coors), 2, np.array([1504, 1504, 40]))
    (bev * _weights(bev.shape)).sum().backward()
    for n, p in net.named_parameters():
        torch.testing.assert_close(r0["grads"][n], r1["grads"][n], rtol=1e-10, atol=1e-12, msg=n)   # ranks agree
        torch.testing.assert_close(r0["grads"][n] * 2, p.grad, rtol=1e-6, atol=1e-9, msg=n)       # == batch gradient
    for k, v in net.state_dict().items():
        if "running" in k:   # SyncBN: statistics over the voxels of BOTH ranks
            torch.testing.assert_close(r0["buffers"][k], v, rtol=1e-8, atol=1e-11, msg=k)
            torch.testing.assert_close(r1["buffers"][k], v, rtol=1e-8, atol=1e-11, msg=k)


def _worker_uneven(rank, world, port, out_dir):
    """rank 1 has an EMPTY sparse tensor in front of a SyncBN'd FeatureBatchNorm1d and produces no gradient for one
    parameter: the forward/backward statistics exchanges and the bucket all-reduces must still pair up."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), S2D_DP_MODE="overlap")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(2)
    import cpu_backend
    cpu_backend.install(None)
    from sparse2dense_amd import dp
    from sparse2dense_amd.spconv import FeatureBatchNorm1d
    from sparse2dense_amd.train_step import backward_and_clip
    dp.init_distributed("gloo")

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(4, 8)
            self.bn = FeatureBatchNorm1d(8, eps=1e-3, momentum=0.01)
            self.only_rank0 = torch.nn.Parameter(torch.ones(8))

        def forward(self, x, use_extra):
            y = self.bn(self.lin(x))
            return (y * self.only_rank0).sum() if use_extra else y.sum() * 1.0

    torch.manual_seed(3)
    net = dp.wrap_ddp(Net().double())
    x = torch.randn(6 if rank == 0 else 0, 4, dtype=torch.float64, generator=torch.Generator().manual_seed(5))
    loss = net(x, use_extra=(rank == 0))
    backward_and_clip(loss, list(net.parameters()), max_norm=1e30)
    torch.save({n: p.grad.clone() for n, p in net.named_parameters()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_empty_rank_and_unused_parameter_do_not_unbalance_collectives():
    port = 31700 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_uneven, args=(2, port, d), nprocs=2, join=True)
        r0 = torch.load(os.path.join(d, "rank0.pt"))
        r1 = torch.load(os.path.join(d, "rank1.pt"))
    for n in r0:
        torch.testing.assert_close(r0[n], r1[n], rtol=1e-12, atol=1e-14, msg=n)
    # reference: the 6 rows in one process (rank 1 contributed nothing); averaged over 2 ranks -> x 1/2
    torch.manual_seed(3)
    lin = torch.nn.Linear(4, 8).double()
    bn = torch.nn.BatchNorm1d(8, eps=1e-3, momentum=0.01).double()
    extra = torch.ones(8, dtype=torch.float64, requires_grad=True)
    x = torch.randn(6, 4, dtype=torch.float64, generator=torch.Generator().manual_seed(5))
    (bn(lin(x)) * extra).sum().backward()
    torch.testing.assert_close(r0["only_rank0"] * 2, extra.grad, rtol=1e-9, atol=1e-12)
    torch.testing.assert_close(r0["lin.weight"] * 2, lin.weight.grad, rtol=1e-7, atol=1e-10)
    torch.testing.assert_close(r0["bn.weight"] * 2, bn.weight.grad, rtol=1e-7, atol=1e-10)


def _worker_order(rank, world, port, out_dir):
    """Many small buckets of DIFFERENT sizes; the parameter of the FIRST bucket (last registered = first gradient of the backward)
    gets no gradient on rank 1.  Before r03 rank 1 launched buckets 1, 2, ... from their hooks and bucket 0 in finish(), rank 0
    launched 0, 1, 2, ...: all-reduces of different buckets were paired on the communicator (ADVICE r02 / VERDICT r02 weak #7)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), S2D_DP_MODE="overlap", S2D_BUCKET_MB="0.00001")
    sys.path.insert(0, ROOT)
    torch.set_num_threads(2)
    from sparse2dense_amd import dp
    from sparse2dense_amd.train_step import backward_and_clip
    dp.init_distributed("gloo")

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(4, 8)
            self.b = torch.nn.Linear(8, 16)
            self.c = torch.nn.Linear(16, 4)
            self.gate = torch.nn.ParameterList([torch.nn.Parameter(torch.ones(4))])   # a child registered last -> bucket 0

        def forward(self, x, use_gate):
            y = self.c(torch.tanh(self.b(torch.tanh(self.a(x)))))
            return (y * self.gate[0]).sum() if use_gate else y.sum()

    torch.manual_seed(11)
    net = dp.wrap_ddp(Net().double())
    gb = net._s2d_grad_buckets
    assert len(gb.buckets) == 7 and gb.buckets[0]["params"][0] is net.gate[0]
    assert len({b["flat"].numel() for b in gb.buckets}) > 3          # sizes differ: a mis-paired all-reduce cannot pass silently
    for step in range(2):                                              # second step: prepare() resets the order cursor
        x = torch.randn(5, 4, dtype=torch.float64, generator=torch.Generator().manual_seed(20 + rank + 2 * step))
        loss = net(x, use_gate=(rank == 0))
        backward_and_clip(loss, list(net.parameters()), max_norm=1e30)
        assert gb.launch_log == list(range(7)), gb.launch_log          # index order on EVERY rank
    torch.save({n: p.grad.clone() for n, p in net.named_parameters()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_buckets_launch_in_index_order_when_an_early_bucket_has_no_gradient_on_one_rank():
    port = 35300 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_order, args=(2, port, d), nprocs=2, join=True)
        r0 = torch.load(os.path.join(d, "rank0.pt"))
        r1 = torch.load(os.path.join(d, "rank1.pt"))
    for n in r0:
        torch.testing.assert_close(r0[n], r1[n], rtol=1e-12, atol=1e-14, msg=n)
    # reference: mean over the two ranks' losses of the second step, in one process
    torch.manual_seed(11)

    class Ref(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(4, 8); self.b = torch.nn.Linear(8, 16); self.c = torch.nn.Linear(16, 4)
            self.gate = torch.nn.ParameterList([torch.nn.Parameter(torch.ones(4))])
    ref = Ref().double()
    tot = 0
    for rank in range(2):
        x = torch.randn(5, 4, dtype=torch.float64, generator=torch.Generator().manual_seed(20 + rank + 2))
        y = ref.c(torch.tanh(ref.b(torch.tanh(ref.a(x)))))
        tot = tot + ((y * ref.gate[0]).sum() if rank == 0 else y.sum())
    (tot / 2).backward()
    for n, p in ref.named_parameters():
        torch.testing.assert_close(r0[n], p.grad, rtol=1e-9, atol=1e-12, msg=n)


def _worker_order_syncbn(rank, world, port, out_dir):
    """self-synchronising batch norms on BOTH sides of a parameter that gets no gradient on rank 1.  Rank 0
    launches bucket 0 from its hook, between the SyncBN-backward all-reduces of bn2 and bn1; rank 1 holds it (and every later bucket)
    back until finish(), i.e. behind ALL SyncBN collectives.  On one communicator the two ranks' sequences differ; with the buckets
    on their own process group (r04) both sequences are the same on every rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), S2D_DP_MODE="overlap", S2D_BUCKET_MB="0.00001")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(2)
    import cpu_backend
    from sparse2dense_amd import collective, dp
    from sparse2dense_amd.spconv import FeatureBatchNorm1d
    from sparse2dense_amd.train_step import backward_and_clip
    cpu_backend.install(None)
    dp.init_distributed("gloo")
    calls = []
    orig = collective.allreduce_sum_
    collective.allreduce_sum_ = lambda t: (calls.append(int(t.numel())), orig(t))[1]

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(4, 8)
            self.bn1 = FeatureBatchNorm1d(8)
            self.b = torch.nn.Linear(8, 16)
            self.bn2 = FeatureBatchNorm1d(16)
            self.c = torch.nn.Linear(16, 4)
            self.gate = torch.nn.ParameterList([torch.nn.Parameter(torch.ones(16))])   # registered last -> bucket 0, consumed between bn2 and c

        def forward(self, x, use_gate):
            h = self.bn2(self.b(self.bn1(self.a(x), relu=True)), relu=True)
            if use_gate:
                h = h * self.gate[0]
            return self.c(h).sum()

    torch.manual_seed(13)
    net = dp.wrap_ddp(Net().double())
    gb = net._s2d_grad_buckets
    assert gb.group is not None and gb.group is not dist.group.WORLD
    assert gb.buckets[0]["params"][0] is net.gate[0]
    for step in range(2):
        x = torch.randn(6, 4, dtype=torch.float64, generator=torch.Generator().manual_seed(40 + rank + 2 * step))
        calls.clear()
        loss = net(x, use_gate=(rank == 0))
        backward_and_clip(loss, list(net.parameters()), max_norm=1e30)
        assert gb.launch_log == list(range(len(gb.buckets))), gb.launch_log
        assert len(calls) == 4, calls          # two SyncBN layers: one forward and one backward vector each, on the DEFAULT group
    torch.save({n: p.grad.clone() for n, p in net.named_parameters()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_held_back_bucket_cannot_cross_pair_with_syncbn_collectives():
    port = 36400 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_order_syncbn, args=(2, port, d), nprocs=2, join=True)
        r0 = torch.load(os.path.join(d, "rank0.pt"))
        r1 = torch.load(os.path.join(d, "rank1.pt"))
    assert set(r0) == set(r1)
    for n in r0:
        torch.testing.assert_close(r0[n], r1[n], rtol=1e-12, atol=1e-14, msg=n)
        assert torch.isfinite(r0[n]).all()
    assert float(r0["gate.0"].abs().sum()) > 0      # rank 0's gradient of the gated parameter, averaged with rank 1's zeros


def test_direct_rccl_route_falls_back_on_every_rank_when_unavailable():
    """collective.init_direct on a non-NCCL process group: returns False and leaves the torch.distributed route on
    (the agreement all-reduce itself needs a GPU; on gloo the early exit is what every rank takes)."""
    port = 33100 + (os.getpid() % 2000)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from sparse2dense_amd import collective
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        assert collective.init_direct(0) is False and not collective.direct_enabled()
        t = torch.ones(4)
        assert collective.allreduce_sum_(t) is t and float(t.sum()) == 4.0
    finally:
        dist.destroy_process_group()
    assert not collective.direct_enabled() and not collective.sync_on()


def _worker_mock_direct(rank, world, port, out_dir):
    """the uneven two-rank scenario (empty rank, parameter without gradient) plus the sparse backbone, with the SyncBN vectors going
    through collective.py's DIRECT route - its communicator replaced by a mock (a second gloo group that logs every call): the call
    sequence csrc/comm.hip would see at world 2"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), S2D_DP_MODE="overlap", S2D_BUCKET_MB="1")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(2)
    from sparse2dense_amd import collective, dp
    from sparse2dense_amd.train_step import backward_and_clip
    dp.init_distributed("gloo")
    net = dp.wrap_ddp(_build())
    comm = dist.new_group()                      # stands in for the library-owned RCCL communicator
    log = []

    def mock_allreduce(t):
        log.append(int(t.numel()))
        dist.all_reduce(t, group=comm)
    feats, c = _voxels(rank)
    coors = torch.from_numpy(np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1))

    def step():
        for p in net.parameters():
            p.grad = None
        out, _ = net(feats, coors, 1, np.array([1504, 1504, 40]))
        loss = (out * _weights(out.shape)).sum()
        backward_and_clip(loss, list(net.parameters()), max_norm=1e30)
        return {n: p.grad.clone() for n, p in net.named_parameters()}
    ref = step()                                 # torch.distributed route
    collective._DIRECT, collective._DIRECT_PG, collective._direct_call = True, dist.group.WORLD, mock_allreduce
    try:
        assert collective.direct_enabled()
        got = step()
    finally:
        collective._DIRECT, collective._DIRECT_PG, collective._direct_call = False, None, None
    torch.save(dict(log=log, same=all(torch.equal(ref[n], got[n]) for n in ref), n=len(ref)), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_direct_route_call_sequence_at_world_2_through_a_mock_communicator():
    """`csrc/comm.hip` has never seen two ranks (no multi-GPU box in any round).  What CAN be checked without one is
    everything above the C entry: at world 2 both ranks issue the same sequence of direct-route all-reduces (sizes and order - a mismatch
    is a deadlock on a real communicator), interleaved with the gradient buckets' collectives on their own group, and the result equals
    the torch.distributed route bit for bit."""
    port = 33900 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_mock_direct, args=(2, port, d), nprocs=2, join=True)
        r0 = torch.load(os.path.join(d, "rank0.pt"))
        r1 = torch.load(os.path.join(d, "rank1.pt"))
    assert r0["log"] == r1["log"] and len(r0["log"]) >= 40, (len(r0["log"]), len(r1["log"]))   # 21 batch norms x (forward + backward)
    assert r0["same"] and r1["same"]
