#!/usr/bin/env python
"""bench.py — LiDAR frames/s (forward + backward) of the voxel-backbone hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

One step = one pass of the hot path over one batch of synthetic frames that are already resident
in HBM as raw points: device voxelization (+fused reader) -> 8 rulebooks -> SpMiddleResNetFHD
(MFMA sparse convs, fused BN) -> densify -> RPN neck -> CenterHead -> CenterPoint loss ->
backward -> grad-clip(35) -> AdamW step; DDP all-reduce overlapped with backward when N>1.
Default workload = BASELINE.json configs[1] (CenterPoint-voxelnet single stage, 150k-pt 0.1 m
Waymo scene); `--workload s2d_student|s2d_distill` run configs[2].

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     : the dominant hand-written kernel (sparse-conv implicit GEMM instantiation with
                 the largest total time), algorithmic FLOPs / HIP-event-timed launches
  cpu_baseline : the CPU oracle stack (C voxelizer + per-offset gather-mm-scatter backbone + torch
                 CPU neck/head) on a bounded sample of the same workload, host cores stated
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

PEAK_F32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_BF16_MATRIX_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (not the 2:1-sparsity figure)
PEAK_HBM_GBS = 8000.0
RULEBOOK_STATS = None


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--batch", type=int, default=4, help="frames per GPU (weak scaling; reference trains 3-4/GPU)")
    p.add_argument("--points", type=int, default=150000)
    p.add_argument("--workload", default="centerpoint", choices=["centerpoint", "s2d_student", "s2d_distill", "pillar", "pillar_s2d"])
    p.add_argument("--no-optim", action="store_true", help="stop after backward + grad clip")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--cpu-points", type=int, default=150000)
    p.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--torch-profile", action="store_true", help="after the timed region: torch.profiler table of 2 steps "
                   "(ops with input shapes -> stderr); diagnostic only")
    p.add_argument("--dtype", default="bf16", choices=["f32", "bf16"],
                   help="MFMA input dtype of the conv path (fp32 accumulate, fp32 storage/statistics/master weights)")
    p.add_argument("--dense-dtype", default=None, choices=["f32", "bf16"], help="override for the dense neck/head convs")
    p.add_argument("--sparse-dtype", default=None, choices=["f32", "bf16", "s16"],
                   help="override for the sparse stack: bf16 = fp32 storage / bf16 MFMA inputs, s16 = bf16 storage (default with --dtype bf16)")
    p.add_argument("--nchw", action="store_true", help="keep the dense neck/head in NCHW (default: NHWC when bf16)")
    return p.parse_args()


def build_models(args, dev):
    from sparse2dense_amd import hip_ops, waymo_configs
    args.dense_dtype = args.dense_dtype or args.dtype
    args.sparse_dtype = args.sparse_dtype or ("s16" if args.dtype == "bf16" else args.dtype)
    hip_ops.set_sparse_compute_dtype(args.sparse_dtype)
    from sparse2dense_amd.registry import build_detector
    torch.manual_seed(1234)
    teacher = None
    if args.workload == "centerpoint":
        model = build_detector(waymo_configs.centerpoint_voxelnet())
    elif args.workload == "pillar":
        model = build_detector(waymo_configs.centerpoint_pillar())
    elif args.workload == "pillar_s2d":
        model = build_detector(waymo_configs.pillar_s2d_student())
    else:
        model = build_detector(waymo_configs.s2d_student())
        if args.workload == "s2d_distill":
            teacher = build_detector(waymo_configs.centerpoint_voxelnet()).to(dev).eval()
            for p in teacher.parameters():
                p.requires_grad = False
    for m in (model, teacher):
        if m is not None:
            if args.dense_dtype == "bf16":
                m.dense_dtype = torch.bfloat16
            if not args.nchw and args.dense_dtype == "bf16":   # fp32 MIOpen Winograd prefers NCHW (measured)
                m.use_channels_last()
    return model.to(dev).train(), teacher


def make_step(args, model, teacher, frames, optimizer):
    from sparse2dense_amd.train_step import backward_and_clip, distill_loss, single_stage_loss
    params = [p for p in model.parameters() if p.requires_grad]

    def step():
        ex = frames.example()                       # device voxelization of the resident points
        if teacher is not None:
            loss, _ = distill_loss(teacher, model, ex)
        elif args.workload == "pillar_s2d":
            out = model(ex, return_loss=True)          # (losses, F_S_a, F_S_b, preds, mask_loss, offset_loss)
            loss = sum(out[0]["loss"]) + (out[4] + out[5]) * 0.5   # trainer.py:766 weights for the PP branch
        else:
            loss, _ = single_stage_loss(model, ex)
        backward_and_clip(loss, params, 35.0)
        if optimizer is not None:
            optimizer.step()
        return loss

    return step


EVENT_OVERHEAD_US = 0.0


def roofline_pass(step, n_steps=3):
    """Re-runs a few steps with per-launch HIP events around the sparse-conv kernels (on the
    stream they are launched on) and returns the roofline object of the dominant instantiation."""
    from sparse2dense_amd import hip_ops as H
    H.PROFILE = []
    for _ in range(n_steps):
        step()
    torch.cuda.synchronize()
    recs, H.PROFILE = H.PROFILE, None
    # an event pair around NOTHING still measures a few microseconds (the two record operations); calibrate and
    # subtract it, so that the per-launch durations are the kernels' own (rocprofv3 --stats agrees within a few %)
    pairs_ = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
    for a_, b_ in pairs_:
        a_.record(); b_.record()
    torch.cuda.synchronize()
    global EVENT_OVERHEAD_US
    EVENT_OVERHEAD_US = sorted(a_.elapsed_time(b_) for a_, b_ in pairs_)[len(pairs_) // 2] * 1e3
    agg = {}
    rb = dict(ms=0.0, n=0, bytes=0.0)
    for r in recs:
        ms = max(r["start"].elapsed_time(r["end"]) - (0.0 if r["kernel"] == "rulebook_subm" else EVENT_OVERHEAD_US * 1e-3), 1e-4)
        pairs = float(r["pairs"].sum().item()) if r["pairs"] is not None else 0.0
        if r["kernel"] == "rulebook_subm":   # 16 N + 8 R + 4 K bytes
            rb["ms"] += ms; rb["n"] += 1
            rb["bytes"] += 16.0 * r["n_out"] + 8.0 * pairs + 4.0 * r["kvol"]
            continue
        kname = r["kernel"]
        if r.get("dense"):   # mirror of the dispatch in csrc/conv2d_nhwc.hip (which device function serves this launch)
            if r["cout"] % 128 == 0:
                kname = f"conv3x3_k32_nhwc_bf16_kernel<128, {r.get('tile_rows', 128) // 32}>"
            elif r.get("pad") == 1 and r.get("stride") == 1 and r["cin"] >= 128:
                kname = "conv3x3_p1_nhwc_bf16_kernel<64>"
            else:
                kname = "conv3x3_k32_nhwc_bf16_kernel<64, 4>"
        key = (kname, r["cin"], r["cout"], r["n_out"])
        a = agg.setdefault(key, dict(ms=0.0, n=0, flops=0.0, bytes=0.0, tile_rows=r.get("tile_rows", 128)))
        a["ms"] += ms
        a["n"] += 1
        if r.get("dense"):   # dense 3x3 conv, NHWC bf16: 2*M*9*cin*cout flops; every input / output / weight byte once
            a["flops"] += 2.0 * r["n_out"] * 9 * r["cin"] * r["cout"]
            a["bytes"] += 2.0 * (r["in_pixels"] * r["cin"] + r["n_out"] * r["cout"] + 9 * r["cin"] * r["cout"])
            continue
        a["flops"] += 2.0 * pairs * r["cin"] * r["cout"]
        eb = float(r.get("elem_bytes", 4))   # feature storage: fp32, or bf16 on the s16 path (its weight image is bf16 too)
        a["bytes"] += eb * (pairs * r["cin"] + r["n_out"] * r["cout"]) + 8.0 * pairs + eb * r["kvol"] * r["cin"] * r["cout"]
    global RULEBOOK_STATS
    RULEBOOK_STATS = None
    if rb["n"]:
        gbs = rb["bytes"] / (rb["ms"] * 1e-3) / 1e9
        RULEBOOK_STATS = dict(kernel="s2d_rulebook_subm_build (6 launches: set, scan x3, perm, probe)", builds_per_step=rb["n"] // n_steps,
                              avg_us=round(rb["ms"] / rb["n"] * 1e3, 1), algorithmic_gbs=round(gbs, 1),
                              frac_of_hbm_peak=round(gbs / PEAK_HBM_GBS, 4))
    if not agg:
        return None, []
    rows = []
    for (kern, cin, cout, n_out), a in agg.items():
        avg_ms = a["ms"] / a["n"]
        rows.append(dict(kernel=kern, cin=cin, cout=cout, n_out=n_out, tile_rows=a["tile_rows"], launches=a["n"], avg_us=avg_ms * 1e3,
                         total_ms=a["ms"], tflops=a["flops"] / a["n"] / (avg_ms * 1e-3) / 1e12,
                         gbs=a["bytes"] / a["n"] / (avg_ms * 1e-3) / 1e9))
    rows.sort(key=lambda r: -r["total_ms"])

    # the dominant KERNEL is a device function (what rocprofv3 --stats lists); one template instantiation serves several
    # tensor shapes, so group the per-shape rows by instantiation before ranking
    def template_of(r):
        if r["kernel"].startswith("conv3x3_"):
            return r["kernel"]
        if r["kernel"] == "spconv_fwd_s16":
            return f"spconv_fwd_s16_kernel<{r['cin']}, {r['cout']}, {128 if r['cout'] == 128 else 64}>"
        return f"{r['kernel']}<{r['cin']}, {r['cout']}>"
    groups = {}
    for r in rows:
        g = groups.setdefault(template_of(r), dict(rows=[], ms=0.0, n=0, flops=0.0, bytes=0.0))
        g["rows"].append(r)
        g["ms"] += r["total_ms"]; g["n"] += r["launches"]
        g["flops"] += r["tflops"] * 1e12 * r["total_ms"] * 1e-3
        g["bytes"] += r["gbs"] * 1e9 * r["total_ms"] * 1e-3
    name, g = max(groups.items(), key=lambda kv: kv[1]["ms"])
    avg_us = g["ms"] / g["n"] * 1e3
    tflops = g["flops"] / (g["ms"] * 1e-3) / 1e12
    gbs = g["bytes"] / (g["ms"] * 1e-3) / 1e9
    lead = g["rows"][0]["kernel"]
    peak_tf = PEAK_BF16_MATRIX_TFLOPS if (lead.endswith(("bf16", "s16")) or "bf16" in lead) else PEAK_F32_MATRIX_TFLOPS
    dense = lead.startswith("conv3x3")
    intensity = tflops * 1e3 / max(gbs, 1e-9)          # FLOP per algorithmic byte
    hbm_roof_tf = intensity * PEAK_HBM_GBS / 1e3
    shapes = [dict(cin=r["cin"], cout=r["cout"], rows_out=r["n_out"], launches_per_step=r["launches"] // n_steps,
                   avg_us=round(r["avg_us"], 1), tflops=round(r["tflops"], 1)) for r in g["rows"]]
    common = dict(traffic=None, kernel=f"s2d::{name}", avg_launch_us=round(avg_us, 2), launches_per_step=g["n"] // n_steps,
                  algorithmic_tflops=round(tflops, 2), algorithmic_gbs=round(gbs, 1), flop_per_byte=round(intensity, 1),
                  mfma_peak_tflops=peak_tf, shapes=shapes, event_pair_overhead_us_subtracted=round(EVENT_OVERHEAD_US, 2),
                  scope=("dominant hand-written kernel of the step by total time (rocprofv3 --stats agrees, profiles/): "
                         + ("dense 3x3 NHWC bf16 implicit GEMM of the BEV neck/head, forward + data-gradient launches"
                            if dense else "sparse-conv gather implicit GEMM, forward + data-gradient launches")))
    # measured HBM bytes per launch: launch-weighted mean over the shapes, only if every shape has a PMC entry
    per = [(pmc_traffic(r), r["launches"]) for r in g["rows"]]
    if all(tr[0] is not None for tr, _ in per):
        common["traffic"] = round(sum(tr[0] * n for tr, n in per) / g["n"])
        common["traffic_source"] = per[0][0][1]
    else:
        common["traffic_source"] = None
    if hbm_roof_tf < peak_tf:
        roof = dict(bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4), **common)
    else:
        roof = dict(bound="mfma", achieved=round(tflops, 3), peak=peak_tf, unit="TFLOP/s", frac=round(tflops / peak_tf, 4), **common)
    return roof, rows


def effective_cpu_count():
    """usable host cores: affinity mask capped by the cgroup CPU quota (the GPU box shows 256
    logical CPUs but grants 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def pmc_traffic(top):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/r01_final_pmc_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs, keyed by kernel name and launch grid).  The launch grid is a
    function of the tensor shape, so an entry is only found when the profiled shape is the one being benchmarked."""
    try:
        db = json.load(open(os.path.join(ROOT, "profiles", "r01_final_pmc_traffic.json")))["kernels"]
    except Exception:
        return None, None
    xcd = lambda tiles: -(-tiles // 8) * 8
    if top["kernel"].startswith("conv3x3_"):
        bn = 128 if top["cout"] % 128 == 0 else 64
        want, grid = top["kernel"], xcd(-(-top["n_out"] // top.get("tile_rows", 128))) * (top["cout"] // bn) * 256
    elif top["kernel"] == "spconv_fwd_s16":
        bm = 128 if top["cout"] == 128 else 64
        want, grid = f"spconv_fwd_s16_kernel<{top['cin']}, {top['cout']}, {bm}>", xcd(-(-top["n_out"] // bm)) * 256
    else:
        return None, None
    for name, v in db.items():
        if want in name and name.endswith(f"grid={grid}") and v["FETCH_SIZE_KiB"] and v["WRITE_SIZE_KiB"]:
            b = (2.0 * v["FETCH_SIZE_KiB"] + v["WRITE_SIZE_KiB"]) * 1024.0
            return round(b), "profiles/r01_final_pmc_traffic.json (2*FETCH_SIZE + WRITE_SIZE, KiB, per launch of this shape)"
    return None, None


def cpu_baseline_subprocess(args, timeout_s=240):
    """Runs cpu_baseline() in a child process under a hard time limit so that a slow host can never
    stall the bench; returns None (with the reason on stderr) if it does not finish."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-points", str(args.cpu_points)]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
        for line in reversed(out.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        print("cpu_baseline: no result\n" + out.stderr[-2000:], file=sys.stderr)
    except subprocess.TimeoutExpired:
        print(f"cpu_baseline: exceeded {timeout_s}s, omitted", file=sys.stderr)
    return None


def cpu_baseline(args):
    """CPU oracle stack on ONE frame of the same workload (fwd+bwd, single iteration)."""
    from oracle import spconv_ref as R
    from oracle import voxelize as OV
    from sparse2dense_amd import scene, waymo_configs
    from sparse2dense_amd.registry import build_detector
    cores = effective_cpu_count()
    torch.set_num_threads(cores)
    s = scene.make_scene(args.cpu_points, seed=20240928)
    t = scene.assign_targets(s["gt_boxes"], s["gt_classes"])
    ex = {k: [torch.from_numpy(v)[None]] for k, v in t.items()}
    torch.manual_seed(1234)
    det = build_detector(waymo_configs.centerpoint_voxelnet())   # neck/head are torch modules (CPU-capable)
    bb = R.RefSpMiddleResNetFHD(5).train()
    neck, head = det.neck.train(), det.bbox_head.train()
    t0 = time.perf_counter()
    v, c, n = OV.points_to_voxel(s["points"], scene.WAYMO_VOXEL, scene.WAYMO_RANGE, 5, 150000)
    feats = torch.from_numpy(OV.voxel_mean(v, n))
    coors = np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1)
    t1 = time.perf_counter()
    bev, _ = bb(feats, coors, 1, np.array([1504, 1504, 40]))
    t2 = time.perf_counter()
    loss = sum(head.loss(ex, head(neck(bev)))["loss"])
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    total = t4 - t0
    return dict(value=round(1.0 / total, 4), unit="frames/s", cores=cores, kind="port",
                sample=f"1 frame ({args.cpu_points} pts, {c.shape[0]} voxels), fwd+bwd, 1 iteration, no warm-up; "
                       f"voxelize {t1 - t0:.2f}s backbone-fwd {t2 - t1:.2f}s dense-fwd+loss {t3 - t2:.2f}s bwd {t4 - t3:.2f}s")


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args)), flush=True)
        return
    torch.set_num_threads(min(effective_cpu_count(), 16))
    from sparse2dense_amd import dp
    rank, local, world = dp.init_distributed()
    if world != max(args.gpus, 1):
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback for the hot path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    from sparse2dense_amd.data import SyntheticFrames
    torch.backends.cudnn.benchmark = False   # MIOpen exhaustive find costs minutes on a fresh box

    model, teacher = build_models(args, dev)
    model = dp.wrap_ddp(model, local)
    if args.workload.startswith("pillar"):
        from sparse2dense_amd.data import SyntheticPillarFrames
        frames = SyntheticPillarFrames(args.batch, n_points=args.points, seed=20240928 + 1000 * rank, device=dev)
    else:
        frames = SyntheticFrames(args.batch, n_points=args.points, seed=20240928 + 1000 * rank,
                                 distill=(args.workload not in ("centerpoint",)), device=dev)
    optimizer = None
    if not args.no_optim:
        optimizer = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, betas=(0.9, 0.99),
                                      weight_decay=0.01, fused=True)
    step = make_step(args, model, teacher, frames, optimizer)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if args.torch_profile and rank == 0:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            step()
            step()
            torch.cuda.synchronize()
        print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=70,
                                                                 max_name_column_width=48, max_shapes_column_width=70),
              file=sys.stderr)

    ex = frames.example()
    n_vox = int(ex["coordinates"].shape[0])
    roof, rows = (None, [])
    if rank == 0 and world == 1 and not args.no_roofline:
        roof, rows = roofline_pass(step)
    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base = cpu_baseline_subprocess(args)

    if rank == 0:
        frames_total = args.batch * world * args.steps
        out = {
            "metric": "LiDAR frames/sec (fwd+bwd), 150k-pt 0.1m-voxel synthetic Waymo scene",
            "value": round(frames_total / elapsed, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if (args.dense_dtype, args.sparse_dtype) == ("f32", "f32") else
                     ("bf16 (sparse stack: " + {"f32": "fp32", "bf16": "fp32 storage, bf16 MFMA inputs", "s16": "bf16 storage"}[args.sparse_dtype]
                      + "; dense neck/head: " + ("bf16 NHWC activations" if args.dense_dtype == "bf16" else "fp32")
                      + "; fp32 accumulate, statistics, master weights, optimizer)"),
            "data": "synthetic",
            "config": {"workload": {"centerpoint": "CenterPoint-voxelnet single-stage (BASELINE configs[1])",
                                    "s2d_student": "CenterPoint-voxelnet + S2D student (KD_VoxelNet) fwd+bwd",
                                    "s2d_distill": "CenterPoint-voxelnet + S2D distill, teacher+student dual forward "
                                                   "(BASELINE configs[2])",
                                    "pillar": "CenterPoint-Pillar single stage (PFN path)",
                                    "pillar_s2d": "CenterPoint-Pillar + S2D student (BASELINE configs[4], PFN path)"}[args.workload],
                       "points_per_frame": args.points, "frames_per_gpu": args.batch, "global_batch": args.batch * world,
                       "voxels_per_gpu_batch": n_vox, "parallelism": f"dp{world}", "gradient_allreduce": (dp.dp_mode() if (world > 1 or os.environ.get("S2D_FORCE_DDP") == "1") else "none"),
                       "step": "device voxelize + fwd + loss + bwd + clip" + ("" if args.no_optim else " + AdamW"),
                       "loss": round(float(loss.item()), 4)},
            "roofline": roof, "cpu_baseline": base,
        }
        if RULEBOOK_STATS:
            out["config"]["rulebook_stage"] = RULEBOOK_STATS
        if rows:
            out["config"]["spconv_kernels"] = [
                {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()} for r in rows[:6]]
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
