"""Per-shape timing of the bf16-storage sparse-conv gather kernels on the bench scene's REAL rulebooks (150 k points x 4 frames):
every SubM stage (forward = data gradient shape), the strided convs (forward over nbr_out, data gradient over nbr_in) and the
weight gradients.  One process = one kernel variant (S2D_S16_KERNEL / S2D_RG_PLAN are read once by the library), so A/B runs are
separate invocations:

    S2D_S16_KERNEL=lds python tools/spconv_kernel_bench.py
    S2D_RG_PLAN=2,8 python tools/spconv_kernel_bench.py [--check] [--only 128]

--check compares every output with a float64 gather-matmul restatement on the same bf16 operands (8e-3 of max, the
tests/test_s16_gpu.py bar)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sparse2dense_amd import hip_ops as H, waymo_configs
from sparse2dense_amd.data import SyntheticFrames, attach_geometry
from sparse2dense_amd.registry import build_detector

ap = argparse.ArgumentParser()
ap.add_argument("--check", action="store_true")
ap.add_argument("--only", type=int, default=0, help="only shapes with this many input channels")
ap.add_argument("--wgrad", action="store_true")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--trace", action="store_true", help="with S2D_RG_DEBUG bit 32: per-step s_memtime stamps of wave 0 of every workgroup")
args = ap.parse_args()
dev = torch.device("cuda:0")
model = build_detector(waymo_configs.s2d_student()).to(dev)
frames = SyntheticFrames(4, n_points=150000, seed=20240928, distill=True, device=dev, beam_jitter=2.5e-3)
ex = attach_geometry(frames.example(), model.backbone, keys=("coordinates",))
plan = ex["coordinates"]._s2d_geometry[2]


def timeit(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def ref_conv(feat, w, nbr, n_out):
    out = torch.zeros(n_out, w.shape[2], device=feat.device, dtype=torch.float64)
    fr, wr = feat.double(), w.to(torch.bfloat16).double()
    for k in range(w.shape[0]):
        o = torch.nonzero(nbr[k] >= 0).squeeze(1)
        if o.numel():
            out[o] += fr[nbr[k][o].long()] @ wr[k]
    return out


cases = []   # (name, nbr, n_in, n_out, cin, cout, pairs)
for key, c in (("res0", 16), ("res1", 32), ("res2", 64), ("res3", 128)):
    rb = plan[key]
    cases.append((f"subm {key}", rb.nbr_out, rb.n_in, rb.n_out, c, c, int(rb.pair_count.sum())))
for i, (ci, co) in enumerate(((16, 32), (32, 64), (64, 128), (128, 128))):
    rb = plan[("conv", f"down{i}")]
    r = int(rb.pair_count.sum())
    cases.append((f"down{i} fwd", rb.nbr_out, rb.n_in, rb.n_out, ci, co, r))
    cases.append((f"down{i} dgrad", rb.nbr_in, rb.n_out, rb.n_in, co, ci, r))

rows = []
for name, nbr, n_in, n_out, cin, cout, r in cases:
    if args.only and cin != args.only:
        continue
    kvol = nbr.shape[0]
    torch.manual_seed(cin * 131 + cout)
    feat = torch.randn(n_in, cin, device=dev).to(torch.bfloat16)
    w = torch.randn(kvol, cin, cout, device=dev) * 0.05
    packed, kv, ci_, co_ = H.spconv_s16_pack(w, n_out)
    fn = lambda: H.spconv_s16_run(feat, packed, kv, ci_, co_, None, nbr, n_out)
    us = timeit(fn, args.iters)
    rec = dict(case=name, cin=cin, cout=cout, n_out=n_out, kvol=kvol, pairs=r, us=round(us, 1),
               tflops=round(2.0 * r * cin * cout / us * 1e-6, 1), density=round(r / (kvol * n_out), 3))
    if args.check:
        out, partial = H.spconv_s16_run(feat, packed, kv, ci_, co_, None, nbr, n_out, bn_stats=True)
        ref = ref_conv(feat, w, nbr, n_out)
        rec["err"] = float((out.double() - ref).abs().max() / ref.abs().max())
        s = partial.double().sum(0)
        o64 = out.double()
        rec["stats_err"] = float(max((s[0] - o64.sum(0)).abs().max() / o64.sum(0).abs().max(), (s[1] - (o64 * o64).sum(0)).abs().max() / (o64 * o64).sum(0).abs().max()))
    if args.wgrad and kvol == 27:
        dout = torch.randn(n_out, cout, device=dev).to(torch.bfloat16)
        usw = timeit(lambda: H.spconv_s16_wgrad(feat, dout, nbr, kvol), args.iters)
        rec["wgrad_us"] = round(usw, 1)
        rec["wgrad_tflops"] = round(2.0 * r * cin * cout / usw * 1e-6, 1)
    if args.trace:
        import ctypes
        from sparse2dense_amd import _lib
        lib = _lib.load()
        tb = torch.zeros((4096, 64), dtype=torch.int64, device=dev)
        lib.s2d_debug_rg_trace.argtypes = [ctypes.c_void_p]
        lib.s2d_debug_rg_trace.restype = None
        lib.s2d_debug_rg_trace(ctypes.c_void_p(tb.data_ptr()))
        fn(); torch.cuda.synchronize()
        lib.s2d_debug_rg_trace(None)
        t = tb.cpu().numpy().astype("float64")
        used = t[:, 0] > 0
        t = t[used]
        if len(t):
            t0 = t[:, 0].min()
            nst = int((t[0, 2:60] > 0).sum())
            rt0, rt1 = t[:, 60], t[:, 61]
            d = t[:, 1:2 + nst] - t[:, 0:1 + nst]          # [prologue, step 0, step 1, ...] durations per workgroup (100 MHz ticks?)
            rec["trace"] = dict(blocks=int(len(t)), start_spread=float(t[:, 0].max() - t0), prologue_med=float(np.median(d[:, 0])),
                                step_med=[float(x) for x in np.median(d[:, 1:], 0)], step_max=[float(x) for x in d[:, 1:].max(0)],
                                epilogue_med=float(np.median(t[:, 63] - t[:, 1 + nst])), total_med=float(np.median(t[:, 63] - t[:, 0])),
                                realtime_us=dict(wg_med=float(np.median(rt1 - rt0)) / 100, first_start_to_last_end=float(rt1.max() - rt0.min()) / 100,
                                                 start_spread=float(rt0.max() - rt0.min()) / 100, end_spread=float(rt1.max() - rt1.min()) / 100),
                                ghz=float(np.median((t[:, 63] - t[:, 0]) / np.maximum(rt1 - rt0, 1)) / 10),
                                total_pct=[float(x) for x in np.percentile(t[:, 63] - t[:, 0], [0, 10, 50, 90, 100])],
                                loop_pct=[float(x) for x in np.percentile(t[:, 1 + nst] - t[:, 1], [0, 10, 50, 90, 100])],
                                end_spread=float(t[:, 63].max() - t[:, 63].min()), wall=float(t[:, 63].max() - t0))
    rows.append(rec)
    print(json.dumps(rec), flush=True)
big = [x for x in rows if x["cin"] >= 64 and x["cout"] >= 64]
if big:
    # step-weighted: SubM launches run 4 fwd + 4-5 dgrad per step, strided ones once
    wsum = lambda x: (9 if x["case"].startswith("subm") and x["cin"] == 128 else 8 if x["case"].startswith("subm") else 1)
    fl = sum(2.0 * x["pairs"] * x["cin"] * x["cout"] * wsum(x) for x in big)
    tm = sum(x["us"] * wsum(x) for x in big)
    print(json.dumps(dict(summary="C>=64 step-weighted", tflops=round(fl / tm * 1e-6, 1), frac=round(fl / tm * 1e-6 / 2500, 4),
                          kernel=os.environ.get("S2D_S16_KERNEL", "rg"), plan=os.environ.get("S2D_RG_PLAN", "default"))))
