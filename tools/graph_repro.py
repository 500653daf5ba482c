"""Replay determinism of the graphed dense segment at the benchmark's size: one example, no optimizer step, N forward+backward passes;
every pass must produce bit-identical losses and parameter gradients (the eager passes 0-1, the capture pass 2 and the replays).
Prints, per pass, the tensors whose checksum differs from pass 0.
    python tools/graph_repro.py [passes] [batch] [points] [graph 0|1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    points = int(sys.argv[3]) if len(sys.argv) > 3 else 150000
    graph = (sys.argv[4] if len(sys.argv) > 4 else "1") == "1"
    from sparse2dense_amd import graphed, hip_ops, scene, side, waymo_configs
    from sparse2dense_amd.data import SyntheticFrames
    from sparse2dense_amd.registry import build_detector
    side.enable(False)
    hip_ops.set_sparse_compute_dtype("s16")
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    model = build_detector(waymo_configs.s2d_student())
    model.dense_dtype = torch.bfloat16
    model.use_channels_last()
    model = model.to(dev).train()
    if graph:
        model.use_hip_graphs()
    frames = SyntheticFrames(batch, n_points=points, seed=20240928, distill=True, device=dev, beam_jitter=scene.WAYMO_BEAM_JITTER)
    ex = frames.example()
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    params = [p for n, p in model.named_parameters() if p.requires_grad]

    def chk(t):
        t = t.detach().contiguous().reshape(-1)
        if t.element_size() == 2 and t.numel() % 2:
            t = t[:-1]
        return int(t.view(torch.uint8).to(torch.int64).sum().item()) if t.numel() < (1 << 24) else int(t.view(torch.int32).to(torch.int64).sum().item())
    ref = None
    bad_total = 0
    grabbed = {}

    def fwd_hook(mod, args, out):
        bev = out[0]
        if bev.requires_grad:
            bev.register_hook(lambda g: grabbed.__setitem__("dBEV", chk(g)))
        grabbed["BEV"] = chk(bev)
    model.backbone.register_forward_hook(fwd_hook)
    for it in range(passes):
        for p in params:
            p.grad = None
        out = model(ex, return_loss=True, return_feature=True)
        terms = dict(det=sum(out[0]["loss"]), mask=out[4], off=out[5], hm=out[0]["hm_loss"][0], loc=out[0]["loc_loss"][0])
        loss = terms["det"] + terms["mask"] + terms["off"]
        loss.backward()
        torch.cuda.synchronize()
        cur = {"loss:" + k: chk(v) for k, v in terms.items()}
        cur["F_S_a"], cur["F_S_b"] = chk(out[1]), chk(out[2])
        cur.update(grabbed)
        for n, p in zip(names, params):
            cur[n] = None if p.grad is None else chk(p.grad)
        if ref is None:
            ref = cur
            print(f"pass 0: loss {float(loss):.6f}", flush=True)
            continue
        bad = [k for k in cur if cur[k] != ref[k]]
        bad_total += bool(bad)
        good = [k for k in cur if cur[k] == ref[k]]
        print(f"pass {it}: loss {float(loss):.6f} differing {len(bad)}/{len(cur)} " + (f"differing: {bad[:40]}" if len(bad) < len(good) else f"EQUAL: {good}"), flush=True)
    print("graph stats", graphed.stats, "passes with differences:", bad_total, flush=True)


if __name__ == "__main__":
    main()
