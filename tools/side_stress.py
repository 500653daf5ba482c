"""Stress run of the multi-stream modes: 4-step training runs of the S2D student (12 k points) in every stream mode, per-step checksums of
every parameter gradient compared BIT FOR BIT with the single-stream run; prints the first step and the tensors that differ.
    python tools/side_stress.py 12 [mode:pcr,...] [points] [frames]   # 12 repetitions of (sparse | all weight gradients on the side stream | + PCR-branch stream)
r04 findings: `spconv_wgrad_s16_coop128` (shared pair ring initialised without a barrier: one run in ~8 had one conv4 weight gradient off
in the last digits; fixed, 0 of 36 afterwards); S2D_PCR_STREAM=1: one run in ~25 with differing backbone gradients (left opt-in); and at the
benchmark's size (`... 3 0:0,1:0 150000 4`) the single-stream run "differed" from itself: NaN gradients - the wrong `spconv_rg_kernel<128,128,2,8>` (DESIGN rule 31)."""
import os, sys, torch
sys.path.insert(0, ".")
from sparse2dense_amd import dense2d, hip_ops, side, waymo_configs
from sparse2dense_amd.data import SyntheticFrames
from sparse2dense_amd.registry import build_detector
from sparse2dense_amd.solver import build_one_cycle_optimizer, build_one_cycle_scheduler
from sparse2dense_amd.train_step import backward_and_clip

def run(mode, pcr, steps=4):
    os.environ["S2D_PCR_STREAM"] = pcr
    side.enable(mode)
    dense2d.clear_pack_cache()
    hip_ops.set_sparse_compute_dtype("s16")
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    model = build_detector(waymo_configs.s2d_student())
    model.dense_dtype = torch.bfloat16
    model.use_channels_last()
    model = model.to(dev).train()
    frames = SyntheticFrames(BATCH, n_points=POINTS, seed=5, distill=True, device=dev)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    params = [p for _, p in named]
    opt = build_one_cycle_optimizer(model, dict(wd=0.01))
    sch = build_one_cycle_scheduler(opt, dict(type="one_cycle", lr_max=0.003, moms=[0.95, 0.85], div_factor=10.0, pct_start=0.4), total_steps=100)
    sums = []
    for it in range(steps):
        out = model(frames.example(), return_loss=True, return_feature=True)
        loss = sum(out[0]["loss"]) + out[4] + out[5]
        sch.step(it)
        backward_and_clip(loss, params, None)
        sums.append([0.0 if p.grad is None else float(p.grad.double().abs().sum()) for p in params] + [float(loss.detach())])
        opt.clip_and_step(35.0)
    side.enable(False)
    return sums, [n for n, _ in named] + ["loss"]

POINTS = int(sys.argv[3]) if len(sys.argv) > 3 else 12000
BATCH = int(sys.argv[4]) if len(sys.argv) > 4 else 1
COMBOS = [tuple(c.split(":")) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [("sparse", "0"), ("1", "0"), ("sparse", "1")]
ref, names = run("0", "0")
fails = 0
PREV = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    for mode, pcr in COMBOS:
        got, _ = run(mode, pcr)
        if os.environ.get("S2D_STRESS_VS_PREV") == "1":   # compare with the previous run instead of the first one (first-run effects)
            ref, got_prev = (PREV[0] if PREV else ref), got
            PREV[:] = [got_prev]
        for s, (a, b) in enumerate(zip(got, ref)):
            bad = [names[i] for i, (x, y) in enumerate(zip(a, b)) if x != y]
            if bad and os.environ.get("S2D_STRESS_DETAIL") == "1":
                rel = {names[i]: abs(x - y) / max(abs(y), 1e-30) for i, (x, y) in enumerate(zip(a, b)) if x != y}
                same = [names[i] for i, (x, y) in enumerate(zip(a, b)) if x == y and names[i].startswith("backbone")]
                print("   relative checksum differences (last 12 in registration order):", [(n, f"{r:.1e}") for n, r in list(rel.items())[-12:]], flush=True)
                print("   max rel", max(rel.values()), "; backbone tensors that agree:", same[-8:], flush=True)
            if bad:
                fails += 1
                print(f"FAIL rep {rep} mode {mode} pcr {pcr} first bad step {s}: {len(bad)} tensors, e.g. {bad[:12]}", flush=True)
                break
print("done, fails", fails, flush=True)
