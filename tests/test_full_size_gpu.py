"""End-to-end checks at the BENCHMARK's size (4 x 150 k points) - DESIGN rule 31: a kernel that is right at 8-12 k points can be wrong at the
row counts the benchmark runs it at (r03/r04: `spconv_rg_kernel<128,128,2,8>` above 32 768 rows, NaN backbone gradients in every timed step).

One training step (forward + losses + backward, no update) of the benchmarked workloads from identical weights and frames
  * in the benchmarked bf16-storage mode against the fp32 mode (fp32 rows through the exact-fp32 MFMA kernels, fp32 NCHW neck through
    MIOpen: different kernels end to end - the parity mode of every oracle test): every loss term within 5e-2, every gradient
    finite, gradient norms within a factor 2 and per-tensor cosines with the smooth depth profile of a healthy build (see _compare_modes);
  * in every execution mode of the benchmarked configuration - weight gradients on their own stream, the dense segment as HIP graphs -
    bit-equal to the plain single-stream kernel-by-kernel run.
Reference step: /root/reference/det3d/torchie/trainer/trainer.py:775-811 + hooks/optimizer.py:15-21."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DOCUMENTED_GAP = ("backbone.conv_input", "backbone.conv1", "backbone.conv2")   # DESIGN section 4 (cosine >= 0.85 there at 8 k points)


def _one_step(workload, dtype, graph=False, wgrad="0", passes=1, batch=4, points=150000):
    import bench
    from sparse2dense_amd import dense2d, graphed, side
    from sparse2dense_amd.train_step import backward_and_clip
    argv = sys.argv
    sys.argv = ["bench.py", "--no-cpu-baseline", "--no-roofline", "--no-extras", "--no-prefetch", "--dtype", dtype, "--batch", str(batch),
                "--points", str(points)] + ([] if graph else ["--no-graph"])
    try:
        args = bench.parse()
    finally:
        sys.argv = argv
    side.enable(wgrad if wgrad != "0" else False)
    for k in graphed.stats:
        graphed.stats[k] = 0
    dense2d.clear_pack_cache()
    dev = torch.device("cuda:0")
    model, teacher, frames, step = bench.setup_workload(args, workload, dev, 0)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    try:
        for _ in range(passes):   # (graphs: two eager warm-up calls, then the capture; the LAST pass is the one compared)
            ex = frames.example()
            if workload == "pillar_s2d":
                out = model(ex, return_loss=True)
                det, pcr = sum(out[0]["loss"]), (out[4] + out[5]) * 0.5
                terms = dict(det=det, pcr=pcr)
            else:
                losses, _, _, _, mask_loss, offset_loss = model(ex, return_loss=True, return_feature=True)
                det, pcr = sum(losses["loss"]), mask_loss + offset_loss
                terms = dict(det=det, mask=mask_loss, offset=offset_loss)
            loss = det + pcr
            backward_and_clip(loss, [p for _, p in named], None)
            torch.cuda.synchronize()
        terms = {k: float(v.detach()) for k, v in terms.items()}
        grads = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in named}
        stats = dict(graphed.stats)
    finally:
        side.enable(False)
        from sparse2dense_amd import hip_ops
        hip_ops.set_sparse_compute_dtype("f32")
        dense2d.clear_pack_cache()
    del model, frames, step
    torch.cuda.empty_cache()
    return terms, grads, stats


def _compare_modes(t16, g16, t32, g32, documented=DOCUMENTED_GAP, min_cos=0.4, min_cos_documented=0.25):
    for k in t32:
        assert abs(t16[k] - t32[k]) <= 5e-2 * abs(t32[k]) + 1e-6, ("loss term", k, t16[k], t32[k])
    report = []
    for n, a in g32.items():
        b = g16[n]
        assert (a is None) == (b is None), n
        if a is None:
            continue
        assert bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all()), f"non-finite gradient: {n}"
        a, b = a.double().flatten(), b.double().flatten()
        if float(a.norm()) == 0.0 and float(b.norm()) == 0.0:
            continue
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-300))
        report.append((cos, n, float(a.norm()), float(b.norm()), a.numel()))
    report.sort()
    print("lowest cosines (bf16-storage mode vs fp32 mode):", [(round(r[0], 4), r[1]) for r in report[:12]])
    if os.environ.get("S2D_TEST_REPORT"):
        with open(os.environ["S2D_TEST_REPORT"], "a") as f:
            f.write(f"# loss terms fp32 {t32} bf16 {t16}\n")
            for r in report:
                f.write(f"{r[0]:+.4f} |g32| {r[2]:.3e} |g16| {r[3]:.3e} n {r[4]:8d} {r[1]}\n")
    # What the numbers look like on a healthy build (r05, random initial weights, training-mode batch norms): 1.000 at the head's last layers,
    # 0.95 at its shared conv, 0.92 at the neck's up-sampling branches, 0.62-0.70 through the S2D module and the sparse stack - a smooth decay
    # with depth (ReLU / GELU' / sign(L1) decisions flipped by bf16 rounding of the activations, amplified by 40 training-mode batch norms
    # at random initialisation; the well-conditioned variants of tests/test_distill_gpu.py hold 5e-2 on the same tensors).  A kernel that is
    # wrong at this size does not look like that: non-finite values, a group of tensors near 0, or norms off by a factor.  Hence: finite,
    # norms within a factor 2, cosine >= 0.4 (>= 0.25 in the three documented early sparse stages), for every tensor that HAS a gradient
    # (r06: these floors are now CALIBRATED - tests/test_oracle_full_size_gpu.py runs the float64 oracle with only the bf16 storage roundings at
    # 150 k points: its own cosines to the exact gradient are 0.27-0.8 through the S2D module and the sparse stack, median 0.773 against the
    # product's 0.776; the bound that pins the kernels is there, this one catches a kernel that breaks at 4 x 150 k points)
    # - 0.4, not the measured floor: a 128-element batch-norm scale of the pillar S2D module sat at 0.50 and moved to 0.49 when the order of a
    # statistics fold changed (r05, `bn_reduce_finalize_*`); the small tensors scatter by a few hundredths with any reordering -
    # (conv biases in front of a training-mode batch norm have a mathematically zero one: fp32 norm <= 1e-2, skipped; so are vectors of <= 4
    # elements and the LayerNorm affines).
    for cos, n, n32, n16, numel in report:
        if n32 <= 1e-2 or numel <= 4:
            continue
        if "norm.weight" in n or "norm.bias" in n or (".1.weight" in n and "convnext" in n) or (".1.bias" in n and "convnext" in n):
            continue
        assert 0.5 <= n16 / n32 <= 2.0, ("gradient norm", n, n16, n32)
        floor = min_cos_documented if n.startswith(documented) else min_cos
        assert cos >= floor, (n, cos, floor)


def test_s2d_student_step_at_benchmark_size_bf16_vs_fp32_and_across_execution_modes():
    t32, g32, _ = _one_step("s2d_student", "f32")
    t16, g16, _ = _one_step("s2d_student", "bf16")
    print("loss terms fp32:", t32, "bf16-storage:", t16)
    _compare_modes(t16, g16, t32, g32)
    del g32
    # weight gradients of every wired layer kind on the second stream: bit-equal at this size (r04: `tools/side_stress.py`)
    ts, gs, _ = _one_step("s2d_student", "bf16", wgrad="1")
    assert ts == t16
    for n, g in g16.items():
        assert (g is None) == (gs[n] is None) and (g is None or torch.equal(g, gs[n])), ("weight-gradient stream", n)
    del gs
    # the dense segment replayed as HIP graphs: third pass = first replay, fifth = third replay (a memset node inside the graph went
    # out of order from the FOURTH replay on, r05); no optimizer step in between, so every pass must reproduce the eager gradients
    tg, gg, st = _one_step("s2d_student", "bf16", graph=True, passes=7)
    assert st["capture"] >= 1 and st["replay"] >= 5 * st["capture"], st
    assert tg == t16, (tg, t16)
    for n, g in g16.items():
        assert (g is None) == (gg[n] is None) and (g is None or torch.equal(g, gg[n])), ("HIP graphs", n)


def test_pillar_s2d_step_at_benchmark_size_bf16_vs_fp32():
    t32, g32, _ = _one_step("pillar_s2d", "f32")
    t16, g16, _ = _one_step("pillar_s2d", "bf16")
    print("pillar loss terms fp32:", t32, "bf16:", t16)
    _compare_modes(t16, g16, t32, g32, documented=("reader.",))
