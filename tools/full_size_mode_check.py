"""End-to-end sanity check at the BENCHMARK's size (4 x 150 k points): one S2D-student training step in the benchmarked bf16-storage mode against the same
step in the fp32 mode (fp32 rows through the exact-fp32 MFMA kernels, fp32 NCHW neck through MIOpen: different kernels end to end, the parity mode of every
oracle test) from identical weights and frames.  Prints the loss terms and, per parameter tensor, the cosine between the two gradients; a kernel that is wrong
only at full size (DESIGN rule 31) shows up as non-finite values or as a group of tensors with a cosine far below its neighbours'.
    python tools/full_size_mode_check.py [--batch 4] [--points 150000]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def one_step(dtype, extra):
    sys.argv = [sys.argv[0], "--no-cpu-baseline", "--no-roofline", "--no-extras", "--no-prefetch", "--dtype", dtype] + extra
    args = bench.parse()
    from sparse2dense_amd import dense2d, side
    from sparse2dense_amd.train_step import backward_and_clip
    side.enable("0")
    dense2d.clear_pack_cache()
    dev = torch.device("cuda:0")
    model, teacher, frames, step = bench.setup_workload(args, "s2d_student", dev, 0)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    losses, _, _, _, mask_loss, offset_loss = model(frames.example(), return_loss=True, return_feature=True)
    loss = sum(losses["loss"]) + (mask_loss + offset_loss)
    backward_and_clip(loss, [p for _, p in named], None)
    torch.cuda.synchronize()
    terms = dict(total=float(loss.detach()), det=float(sum(losses["loss"]).detach()), mask=float(mask_loss.detach()), offset=float(offset_loss.detach()))
    grads = {n: (None if p.grad is None else p.grad.detach().double().flatten().cpu()) for n, p in named}
    del model, frames, step
    torch.cuda.empty_cache()
    return terms, grads


def main():
    extra = sys.argv[1:]
    torch.cuda.set_device(0)
    t32, g32 = one_step("f32", extra)
    t16, g16 = one_step("bf16", extra)
    print("loss terms fp32:", t32)
    print("loss terms bf16:", t16)
    rows = []
    for n in g32:
        a, b = g32[n], g16[n]
        if a is None or b is None:
            continue
        fin = bool(torch.isfinite(a).all() and torch.isfinite(b).all())
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-300)) if fin else float("nan")
        rel = float((a - b).norm() / (a.norm() + 1e-300)) if fin else float("nan")
        rows.append((n, fin, cos, rel))
    bad = [r for r in rows if not r[1]]
    print(f"{len(rows)} gradient tensors, non-finite in either mode: {len(bad)}", [r[0] for r in bad[:6]])
    groups = {}
    for n, fin, cos, rel in rows:
        key = ".".join(n.split(".")[:2])
        groups.setdefault(key, []).append((cos, rel, n))
    for key, v in groups.items():
        worst = min(v)
        print(f"  {key:28s} tensors {len(v):3d}  min cosine {worst[0]:.4f} ({worst[2]})  median rel. diff {sorted(x[1] for x in v)[len(v) // 2]:.3f}")


if __name__ == "__main__":
    main()
