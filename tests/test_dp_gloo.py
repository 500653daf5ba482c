"""N>1 path on CPU: world_size-2 `gloo` job.  Each rank runs the PRODUCT module stack (HIP
launchers swapped for tests/cpu_backend.py) on its own frame under sparse2dense_amd.dp.wrap_ddp:
  * FeatureBatchNorm1d statistics are global (all-reduced sums + counts, different N per rank);
  * BatchNorm2d layers are converted to SyncBatchNorm, FeatureBatchNorm1d is left in place;
  * after DDP's bucketed all-reduce every rank holds the gradient of the mean-of-ranks loss, equal
    to the single-process gradient on the concatenated 2-frame batch (x 1/2 for the sum loss)."""
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
                      LOCAL_RANK=str(rank), S2D_DP_MODE=mode)
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
    (bev * _weights((2,) + bev.shape[1:])[rank:rank + 1]).sum().backward()
    if mode == "flat":
        assert dp.flat_enabled(ddp)
        dp.allreduce_grads(list(net.parameters()))
    torch.save({"grads": {n: p.grad.clone() for n, p in net.named_parameters()},
                "buffers": {k: v.clone() for k, v in net.state_dict().items() if "running" in k}},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["ddp", "flat"])
def test_two_rank_dp_equals_single_process_batch(monkeypatch, mode):
    port = 29500 + (os.getpid() % 2000) + (7 if mode == "flat" else 0)
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
