"""Sparse-conv kernel over mask-sorted rows (csrc/rulebook_sort.hip + spconv_rg_kernel<..., SORTED>) against the plain kernel on the bench
scene's REAL rulebooks (4 x 150 k points): 64 -> 64 at stage 2, 128 -> 128 at stage 3, forward shape (= data-gradient shape).
    python tools/mask_sort_kernel_bench.py            # S2D_RG_SORT_CHUNKS=1: global sort instead of XCD-local chunks
Prints per stage: rows, pairs, R/(27 N), active (workgroup, offset) and (tile, offset) fractions of the sorted order, the sort's own time,
both kernels' times and algorithmic fractions of the bf16 MFMA peak (2.5 PFLOP/s)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse2dense_amd import _lib, hip_ops as H, waymo_configs
from sparse2dense_amd.data import SyntheticFrames, attach_geometry
from sparse2dense_amd.registry import build_detector

H.set_sorted_rows(True)   # (opt-in mode: also puts 64 -> 64 on the register-gather weight image, for both kernels timed here)
dev = torch.device("cuda:0")
model = build_detector(waymo_configs.s2d_student()).to(dev)
frames = SyntheticFrames(4, n_points=150000, seed=20240928, distill=True, device=dev, beam_jitter=2.5e-3)
ex = attach_geometry(frames.example(), model.backbone, keys=("coordinates",))
plan = ex["coordinates"]._s2d_geometry[2]


def timeit(fn, n=30):
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


out = {}
for key, c in (("res2", 64), ("res3", 128)):
    rb = plan[key]
    n = int(rb.n_out)
    pairs = int(rb.pair_count.sum())
    feat = torch.randn(n, c, device=dev).to(torch.bfloat16)
    w = torch.randn(27, c, c, device=dev) * 0.05
    packed, kvol, cin, cout = H.spconv_s16_pack(w, n)

    def sort_only():
        if hasattr(rb, "_sorted_rows"):
            del rb._sorted_rows
        H.rulebook_sorted_rows(rb)
    t_sort = timeit(sort_only, 10)
    perm, pmask, _ = H.rulebook_sorted_rows(rb)
    pm = pmask.long()
    tiles_wg = int(_lib.load().s2d_spconv_s16_stats_tiles(n, 27, c, c))
    tpb = -(-(-(-n // 16)) // tiles_wg)

    def active(group):
        ng = -(-n // group)
        pad = torch.zeros(ng * group, dtype=torch.int64, device=dev)
        pad[:n] = pm
        u = pad.view(ng, group)
        acc = torch.zeros(ng, dtype=torch.int64, device=dev)
        for k in range(27):
            acc += ((u >> k) & 1).amax(1)
        return float(acc.sum()) / (27 * ng)
    t_plain = timeit(lambda: H.spconv_s16_run(feat, packed, kvol, cin, cout, None, rb.nbr_out, n, None, "fwd"))
    t_sorted = timeit(lambda: H.spconv_s16_run_sorted(feat, packed, kvol, cin, cout, None, rb, "fwd"))
    a = H.spconv_s16_run(feat, packed, kvol, cin, cout, None, rb.nbr_out, n, None, "fwd").float()
    b = H.spconv_s16_run_sorted(feat, packed, kvol, cin, cout, None, rb, "fwd").float()
    flops = 2.0 * pairs * c * c
    out[key] = dict(rows=n, pairs=pairs, occupancy=pairs / (27.0 * n), tiles_per_workgroup=tpb, active_workgroup_offsets=active(16 * tpb),
                    active_tile_offsets=active(16), sort_us=t_sort, plain_us=t_plain, sorted_us=t_sorted,
                    plain_frac=flops / (t_plain * 1e-6) / 2.5e15, sorted_frac=flops / (t_sorted * 1e-6) / 2.5e15,
                    max_abs_diff=float((a - b).abs().max()), max_abs=float(a.abs().max()))
    print(key, json.dumps(out[key]), flush=True)
print(json.dumps(out))
