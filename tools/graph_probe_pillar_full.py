"""Pillar detector with the whole dense segment graphed (S2D_PILLAR_GRAPH=1): which outputs / gradients go non-finite in replay, and when."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("S2D_PILLAR_GRAPH", "1")
import torch

from sparse2dense_amd import graphed, hip_ops, waymo_configs
from sparse2dense_amd.data import SyntheticPillarFrames
from sparse2dense_amd.registry import build_detector

dev = torch.device("cuda:0")
torch.manual_seed(0)
hip_ops.set_sparse_compute_dtype("s16")
det = build_detector(waymo_configs.pillar_s2d_student())
det.dense_dtype = torch.bfloat16
det.use_channels_last()
det = det.to(dev).train()
if len(sys.argv) > 1 and sys.argv[1] == "graph":
    det.use_hip_graphs()
B, NP = int(os.environ.get("B", "2")), int(os.environ.get("NP", "30000"))
frames = SyntheticPillarFrames(B, n_points=NP, seed=3, device=dev)
named = [(n, p) for n, p in det.named_parameters() if p.requires_grad]
for it in range(6):
    for _, p in named:
        p.grad = None
    out = det(frames.example(), return_loss=True)
    terms = dict(det=sum(out[0]["loss"]), mask=out[4], off=out[5])
    (terms["det"] + 0.5 * (terms["mask"] + terms["off"])).backward()
    torch.cuda.synchronize()
    bad = [n for n, p in named if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    none = [n for n, p in named if p.grad is None]
    fo = {k: bool(torch.isfinite(v.float()).all()) for k, v in zip(("F_S_a", "F_S_b"), out[1:3])}
    print(it, {k: round(float(v), 5) for k, v in terms.items()}, fo, "nonfinite grads:", len(bad), bad[:3], bad[-3:], "no grad:", len(none), none[:4], graphed.stats, flush=True)
