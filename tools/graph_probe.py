"""Bisect HIP-graph capture problems of the dense segment: every case runs in a child process (a segfault in hipStreamEndCapture must
not take the others down) and reports whether forward / backward capture and a replay work.
    python tools/graph_probe.py            # all cases
    python tools/graph_probe.py CASE       # one case in this process"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = ["linear", "conv3x3", "conv_bn", "rpn", "head", "head_loss", "s2d_neck_nopcr", "s2d_neck_pcr", "student_b2"]


def capture(fn, inputs, modules, tag):
    params = [p for m in modules for p in m.parameters()]
    import torch
    from sparse2dense_amd.graphed import GraphedSegment
    seg = GraphedSegment(fn, modules, name=tag)
    for it in range(5):
        for p in params:
            p.grad = None
        outs = seg(*inputs)
        loss = sum(o.float().sum() for o in outs if o.requires_grad)
        loss.backward()
        torch.cuda.synchronize()
        print(tag, "iter", it, "loss", float(loss), flush=True)
    from sparse2dense_amd import graphed
    print(tag, "stats", graphed.stats, flush=True)


def run(case):
    import torch
    from sparse2dense_amd import hip_ops, waymo_configs
    from sparse2dense_amd.registry import build_detector
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    bf = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
    if case == "linear":
        m = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8)).to(dev)
        x = torch.randn(32, 64, device=dev, requires_grad=True)
        return capture(lambda x_: (m(x_),), [x], [m], case)
    from sparse2dense_amd import dense2d as D
    if case in ("conv3x3", "conv_bn"):
        layers = [D.Conv3x3(64, 64, 3, 1, 1, bias=False)] + ([D.FastBatchNorm2d(64), torch.nn.ReLU()] if case == "conv_bn" else [])
        m = torch.nn.Sequential(*D.fuse_bn_relu(layers)).to(dev).train()
        x = bf(2, 64, 48, 48)

        def fn(x_):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return (m(x_),)
        return capture(fn, [x], [m], case)
    hip_ops.set_sparse_compute_dtype("s16")
    kind = "centerpoint_voxelnet" if case in ("rpn", "head", "head_loss") else "s2d_student"
    det = build_detector(getattr(waymo_configs, kind)())
    det.dense_dtype = torch.bfloat16
    det.use_channels_last()
    det = det.to(dev).train()
    b = 2
    if case == "rpn":
        x = bf(b, 256, 188, 188)
        return capture(lambda x_: (det._dense(det.neck, x_, keep_first=True),), [x], [det.neck], case)
    if case == "head":
        x = bf(b, 512, 188, 188)

        def fn(x_):
            preds = det._dense(det.bbox_head, x_)
            return tuple(v for p in preds for v in p.values())
        return capture(fn, [x], [det.bbox_head], case)
    from sparse2dense_amd.data import SyntheticFrames
    frames = SyntheticFrames(b, n_points=12000, seed=5, distill=(kind == "s2d_student"), device=dev)
    ex = frames.example()
    tasks = len(det.bbox_head.tasks)
    flat = det._flat_targets(ex, tasks)
    if case == "head_loss":
        x = bf(b, 512, 188, 188)

        def fn(x_, *fl):
            preds = det._dense(det.bbox_head, x_)
            losses = det.bbox_head.loss(det._unflat_targets(fl, tasks), preds)
            return tuple(losses["loss"])
        return capture(fn, [x] + flat, [det.bbox_head], case)
    x = bf(b, 256, 188, 188)
    if case == "s2d_neck_nopcr":
        def fn(x_):
            det.neck.pcr_targets = None
            out = det._dense(det.neck, x_, keep_first=True, keep=(5, 6))
            return (out[0], out[5], out[6])
        det.neck.eval()   # S2D_RPN builds the PCR head only in training mode; eval BN is fine for a capture probe
        return capture(fn, [x], [det.neck], case)
    if case == "s2d_neck_pcr":
        side = []
        for s_ in (4, 2):
            side += list(det._padded_recon(ex, s_))
        det._flush_recon()

        def fn(x_, c4, f4, c2, f2):
            det.neck.pcr_targets = {4: (c4, f4), 2: (c2, f2)}
            out = det._dense(det.neck, x_, keep_first=True, keep=(5, 6))
            return (out[0], out[1], out[2], out[3], out[4])
        return capture(fn, [x] + side, [det.neck], case)
    if case == "student_b2":
        det.use_hip_graphs()
        params = [p for p in det.parameters() if p.requires_grad]
        for it in range(5):
            for p in params:
                p.grad = None
            out = det(frames.example(), return_loss=True, return_feature=True)
            loss = sum(out[0]["loss"]) + out[4] + out[5]
            loss.backward()
            torch.cuda.synchronize()
            print(case, "iter", it, float(loss), flush=True)
        from sparse2dense_amd import graphed
        print(case, "stats", graphed.stats, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
        print("CASE-OK", sys.argv[1], flush=True)
    else:
        for c in CASES:
            r = subprocess.run([sys.executable, "-X", "faulthandler", os.path.abspath(__file__), c], capture_output=True, text=True, timeout=600)
            ok = "CASE-OK" in r.stdout
            print(f"==== {c}: {'ok' if ok else 'FAILED rc=' + str(r.returncode)}")
            print(r.stdout[-1500:])
            if not ok:
                print(r.stderr[-2500:])
