"""Stress run of the multi-stream modes: 4-step training runs of the S2D student (12 k points) in every stream mode, per-step checksums of
every parameter gradient compared BIT FOR BIT with the single-stream run; prints the first step and the tensors that differ.
    python tools/side_stress.py 12 [mode:pcr,...] [points] [frames]   # 12 repetitions of (sparse | all weight gradients on the side stream | + PCR-branch stream)
r04 findings: `spconv_wgrad_s16_coop128` (shared pair ring initialised without a barrier: one run in ~8 had one conv4 weight gradient off
in the last digits; fixed, 0 of 36 afterwards); S2D_PCR_STREAM=1: one run in ~25 with differing backbone gradients (left opt-in); and at the
benchmark's size (`... 3 0:0,1:0 150000 4`) the single-stream run "differed" from itself: NaN gradients - the wrong `spconv_rg_kernel<128,128,2,8>` (DESIGN rule 31).
r05 findings (HISTORY.md section 7 (f), rule 36): S2D_PCR_STREAM=1 (`sparse:1`): 0 mismatches in 140 runs since the dense kernels' accumulators are cleared by a
kernel instead of hipMemsetAsync (rule 32).  Mode `pcr` (the PCR head's weight gradients on the eager side stream): 68 of 70 runs differ; root cause found with the
switches below - not a data hazard but packed-FP32 code of the NEXT level's statistics kernel (`pcr_level_bwd_dense_kernel<32,16,2>`, 544 v_pk_fma_f32) whose high lanes
become timing-dependent while the 16 -> 3 up-sampler's MFMA weight-gradient kernel runs beside it:
  S2D_SIDE_PCR_ONLY=k        only the k-th pcr-kind call of a backward pass leaves the chain (k = 0 alone reproduces it)
  S2D_DEBUG_CT_WGRAD=...     stand-ins for that call: zeros | read | long | lds | ldsfill[:v] (none perturbs) | privws | clone | clone_main (the real kernel on private
                             copies of everything: still perturbs)
  S2D_DEBUG_SUMS2=1          the statistics kernel launched twice back to back: differing sums (odd channels only) are printed per step
  (default build since r06; S2D_BUILD_LOSSES_SLP=1 restores the packed build) losses.hip without the SLP vectoriser: no v_pk_*, 0 mismatches, single-stream results bit-identical to the packed build (STRESS_DUMP=file)
  S2D_SIDE_DEBUG_SYNC=1, PYTORCH_NO_CUDA_MEMORY_CACHING=1, S2D_STRESS_HOOKS=1 [S2D_STRESS_HOLD=tags | S2D_STRESS_PTRS=1], STRESS_WS_POISON=v: the dead ends (serialisation
  effects, workspace poison, held references / address log)
"""
import os, sys, torch
sys.path.insert(0, ".")
from sparse2dense_amd import dense2d, hip_ops, side, waymo_configs
from sparse2dense_amd.data import SyntheticFrames
from sparse2dense_amd.registry import build_detector
from sparse2dense_amd.solver import build_one_cycle_optimizer, build_one_cycle_scheduler
from sparse2dense_amd.train_step import backward_and_clip

REC = []
if os.environ.get("S2D_STRESS_HOOKS") == "1":   # intermediate tensors of the PCR backward, held BY REFERENCE during the pass (no extra launches) and checksummed after it
    from sparse2dense_amd import dense3d, heads
    _lvl, _ct, _bn, _fin = heads._PcrLevelNormFn.backward, dense3d._ConvT3dFn.backward, dense3d.bncm_backward, hip_ops.bn1d_finalize_bwd

    _HOLD = [v for v in os.environ.get("S2D_STRESS_HOLD", "").split(",") if v]

    _PTRS = os.environ.get("S2D_STRESS_PTRS") == "1"   # log addresses instead of holding references
    PTRLOG = []

    def _note(tag, t):
        if not torch.is_tensor(t):
            return
        if _PTRS:
            PTRLOG.append((tag, t.untyped_storage().data_ptr(), t.untyped_storage().nbytes()))
        elif not _HOLD or any(tag.startswith(h) for h in _HOLD):
            REC.append((tag, t))

    def _lvl_bwd(ctx, go_mask, go_off, dz=None):
        _note("level.in.dz", dz)
        out = _lvl(ctx, go_mask, go_off, dz)
        _note("level.out.dy", out[0]); _note("level.out.dgamma", out[1]); _note("level.out.dbeta", out[2]); _note("level.out.dw_mask", out[3]); _note("level.out.dw_off", out[5])
        return out

    def _ct_bwd(ctx, dout, *a, **k):
        _note("convT.in.dout", dout)
        _note("convT.in_norm", getattr(ctx, "in_norm", None))
        _note("convT.x", ctx.saved_tensors[0])
        out = _ct(ctx, dout, *a, **k)
        _note("convT.out.dx", out[0])
        return out

    def _bn_bwd(dy, x, *a, **k):
        _note("bn3d.in.dy", dy); _note("bn3d.in.x", x)
        out = _bn(dy, x, *a, **k)
        _note("bn3d.out.dx", out[0]); _note("bn3d.out.dgamma", out[1])
        return out

    def _fin_bwd(sums, *a, **k):
        _note("finalize.in.sums", sums)
        out = _fin(sums, *a, **k)
        for i, o in enumerate(out[:5]):
            _note(f"finalize.out{i}", o)
        return out
    heads._PcrLevelNormFn.backward, dense3d._ConvT3dFn.backward = staticmethod(_lvl_bwd), staticmethod(_ct_bwd)
    dense3d.bncm_backward, hip_ops.bn1d_finalize_bwd = _bn_bwd, _fin_bwd


def run(mode, pcr, steps=4):
    dense2d._WS_POISON = None if (mode == "0" and pcr == "0" and not REF_DONE) or not os.environ.get("STRESS_WS_POISON") else float(os.environ["STRESS_WS_POISON"])
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
        if os.environ.get("S2D_STRESS_PTRS") == "1" and it == 0:
            seen = {}
            for tag, ptr, nb in PTRLOG:
                prev = seen.get(ptr)
                print(f"   ptr {ptr:#x} {nb:>10d} B {tag}" + (f"   <-- same block as {prev}" if prev else ""), flush=True)
                seen[ptr] = tag
            PTRLOG.clear()
        if os.environ.get("S2D_DEBUG_SUMS2") == "1":
            from sparse2dense_amd import heads as _heads
            for c_, co_, s1, s2, g1, g2 in _heads.DEBUG_SUMS:
                if not (torch.equal(s1, s2) and torch.equal(g1, g2)):
                    print(f"   step {it}: level C={c_} CO={co_}: two back-to-back launches of pcr_level_bwd_sums differ: sums {int((s1 != s2).sum())} of {s1.numel()}, grads {int((g1 != g2).sum())} of {g1.numel()}", flush=True)
                    idx = (s1 != s2).nonzero().flatten().tolist()
                    print("      indices", idx, "first", [float(s1[i]) for i in idx[:6]], "second", [float(s2[i]) for i in idx[:6]], flush=True)
            _heads.DEBUG_SUMS.clear()
        extra = [float(t.double().abs().sum()) for _, t in REC]
        if it == 0 and REC and len(named) == len(params):
            named += [(f"rec{i}:{tag}", None) for i, (tag, _) in enumerate(REC)]
        REC.clear()
        sums.append([0.0 if p.grad is None else float(p.grad.double().abs().sum()) for p in params] + extra + [float(loss.detach())])
        opt.clip_and_step(35.0)
    side.enable(False)
    return sums, [n for n, _ in named] + ["loss"]

POINTS = int(sys.argv[3]) if len(sys.argv) > 3 else 12000
BATCH = int(sys.argv[4]) if len(sys.argv) > 4 else 1
# (a mode with several kinds is written with "+": aux+dense+pcr+sparse:0)
COMBOS = [tuple(v.replace("+", ",") for v in c.split(":")) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [("sparse", "0"), ("1", "0"), ("sparse", "1")]
REF_DONE = False
ref, names = run("0", "0")
REF_DONE = True
if os.environ.get("STRESS_DUMP"):   # the single-stream run's checksums, for comparisons between builds
    import json
    with open(os.environ["STRESS_DUMP"], "w") as f:
        json.dump({"names": names, "sums": ref}, f)
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
                if os.environ.get("S2D_STRESS_HOOKS") == "1":
                    print("   records:", [(names[i], "same" if x == y else f"{abs(x - y) / max(abs(y), 1e-30):.1e}") for i, (x, y) in enumerate(zip(a, b)) if names[i].startswith("rec")], flush=True)
                if os.environ.get("S2D_STRESS_NECK") == "1":
                    print("   neck tensors that AGREE:", [names[i] for i, (x, y) in enumerate(zip(a, b)) if x == y and names[i].startswith("neck")], flush=True)
                    print("   neck tensors that DIFFER:", [(names[i], f"{abs(x - y) / max(abs(y), 1e-30):.1e}") for i, (x, y) in enumerate(zip(a, b)) if x != y and names[i].startswith("neck")], flush=True)
            if bad:
                fails += 1
                print(f"FAIL rep {rep} mode {mode} pcr {pcr} first bad step {s}: {len(bad)} tensors, e.g. {bad[:12]}", flush=True)
                break
print("done, fails", fails, flush=True)
