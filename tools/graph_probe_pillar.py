"""Which part of the pillar detector's dense segment breaks under HIP-graph replay?  Each piece in a child process: 2 eager calls, capture,
3 replays; prints the loss per call and whether every gradient is finite and equal to the eager one.
    python tools/graph_probe_pillar.py [case]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = ["encoder_2", "encoder_2_first", "neck", "conv1x1_miopen", "conv2x2s2_miopen", "encoder_1", "upsample117", "decoder_1", "generator", "pcr_loss", "module_2d", "backbone_dense"]


def run(case):
    import torch
    from sparse2dense_amd import hip_ops, waymo_configs
    from sparse2dense_amd.graphed import GraphedSegment
    from sparse2dense_amd.registry import build_detector
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    hip_ops.set_sparse_compute_dtype("s16")
    det = build_detector(waymo_configs.pillar_s2d_student())
    det.dense_dtype = torch.bfloat16
    det.use_channels_last()
    det = det.to(dev).train()
    bb = det.backbone
    bb.dense_dtype = torch.bfloat16
    bf = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
    ac = lambda f: (lambda *a: _ac(f, *a))

    def _ac(f, *a):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = f(*a)
        return out if isinstance(out, tuple) else (out,)
    if case == "conv1x1_miopen":
        mods, fn, x = [bb.encoder_1[1]], ac(bb.encoder_1[1]), bf(2, 64, 234, 234)
    elif case == "conv2x2s2_miopen":
        mods, fn, x = [bb.encoder_1[4]], ac(bb.encoder_1[4]), bf(2, 32, 234, 234)
    elif case == "encoder_2":
        mods, fn, x = [bb.encoder_2], ac(bb.encoder_2), bf(2, 128, 234, 234)
    elif case == "encoder_2_first":
        mods, fn, x = [bb.encoder_2[0]], ac(bb.encoder_2[0]), bf(2, 128, 234, 234)
    elif case == "neck":
        mods, fn, x = [det.neck], ac(lambda x_: det.neck(x_)), bf(2, 64, 468, 468)
    elif case == "encoder_1":
        mods, fn, x = [bb.encoder_1], ac(bb.encoder_1), bf(2, 64, 468, 468)
    elif case == "upsample117":
        mods, fn, x = [bb.decoder_1[-1]], ac(bb.decoder_1[-1]), bf(2, 128, 59, 59)
    elif case == "decoder_1":
        mods, fn, x = [bb.decoder_1], ac(bb.decoder_1), bf(2, 256, 59, 59)
    elif case == "generator":
        x = torch.randn(2, 64, 1, 468, 468, device=dev, requires_grad=True)
        mods = [bb.generator, bb.gen_out, bb.gen_mask]

        def fn(x_):
            g = bb.generator(x_)
            return bb.gen_mask(g), bb.gen_out(g)
    elif case == "pcr_loss":
        from sparse2dense_amd.heads import mask_offset_loss, metric_grid
        x = torch.randn(2, 3, 1, 468, 468, device=dev, requires_grad=True)
        gm = torch.randn(2, 1, 1, 468, 468, device=dev, requires_grad=True)
        gt = torch.zeros(2, 5, 1, 468, 468, device=dev)
        gt[:, :, :, 100:140, 200:260] = torch.randn(2, 5, 1, 40, 60, device=dev)
        mods = []

        def fn(x_, gm_, gt_):
            return mask_offset_loss(x_, gm_, gt_, metric_grid(2, 1, 468, 468, x_))
        seg = GraphedSegment(fn, mods, name=case)
        return drive(seg, [x, gm, gt], [])
    elif case == "module_2d":
        mods, fn, x = [bb], ac(bb._module_2d), bf(2, 64, 468, 468)
    else:
        x = torch.randn(2, 64, 468, 468, device=dev, requires_grad=True)
        mods, fn = [bb], (lambda c_: tuple(bb.dense_forward(c_)))
    seg = GraphedSegment(fn, mods, name=case)
    drive(seg, [x], [p for m in mods for p in m.parameters()])


def drive(seg, inputs, params):
    import torch
    ref = None
    for it in range(5):
        for p in params:
            p.grad = None
        for t in inputs:
            t.grad = None
        outs = seg(*inputs)
        loss = sum(o.float().square().mean() if o.dim() else o.float() for o in outs if o.requires_grad)
        loss.backward()
        torch.cuda.synchronize()
        gs = [t.grad for t in inputs if t.requires_grad] + [p.grad for p in params]
        fin = all(g is None or bool(torch.isfinite(g).all()) for g in gs)
        cur = [None if g is None else g.detach().clone() for g in gs]
        same = True if ref is None else all((a is None) == (b is None) and (a is None or torch.equal(a, b)) for a, b in zip(cur, ref))
        if ref is None:
            ref = cur
        print(f"{seg.name} call {it}: loss {float(loss):.6f} finite {fin} equal-to-first {same}", flush=True)
    from sparse2dense_amd import graphed
    print(seg.name, "stats", graphed.stats, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
        print("CASE-OK", sys.argv[1], flush=True)
    else:
        for c in CASES:
            r = subprocess.run([sys.executable, "-X", "faulthandler", os.path.abspath(__file__), c], capture_output=True, text=True, timeout=600)
            print(f"==== {c}: {'ok' if 'CASE-OK' in r.stdout else 'FAILED rc=' + str(r.returncode)}")
            print("\n".join(l for l in r.stdout.splitlines() if c in l)[-1200:])
            if "CASE-OK" not in r.stdout:
                print(r.stderr[-1500:])
