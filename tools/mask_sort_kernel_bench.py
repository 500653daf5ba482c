"""(needs the experimental kernel with per-workgroup active-step lists; kept as the record of the measurement)
natural vs mask-sorted output-row order for the SubM s16 kernel on the bench scene's real rulebooks (timing only: with a permuted
map the output rows come out in permuted order)"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from sparse2dense_amd import hip_ops as H, waymo_configs
from sparse2dense_amd.data import SyntheticFrames, attach_geometry
from sparse2dense_amd.registry import build_detector
dev = torch.device("cuda:0")
model = build_detector(waymo_configs.s2d_student()).to(dev)
frames = SyntheticFrames(4, n_points=150000, seed=20240928, distill=True, device=dev, beam_jitter=2.5e-3)
ex = attach_geometry(frames.example(), model.backbone, keys=("coordinates",))
plan = ex["coordinates"]._s2d_geometry[2]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for key, c in (("res0", 16), ("res1", 32), ("res2", 64), ("res3", 128)):
    rb = plan[key]
    n = rb.n_out
    nbr = rb.nbr_out                       # [27, n]
    mask = torch.zeros(n, dtype=torch.int64, device=dev)
    for k in range(27):
        mask |= (nbr[k] >= 0).long() << k
    feat = torch.randn(n, c, device=dev).to(torch.bfloat16)
    w = torch.randn(27, c, c, device=dev) * 0.05
    packed, kvol, cin, cout = H.spconv_s16_pack(w, n)
    res = {}
    ref = None
    for name, chunk in (("natural", None), ("sorted/xcd-eighth", -(-n // 8)), ("sorted/global", n), ("sorted/8192", 8192)):
        if chunk is None:
            perm = torch.arange(n, device=dev)
        else:
            key_ = (torch.arange(n, device=dev) // chunk) * (1 << 27) + mask
            perm = torch.argsort(key_, stable=True)
        nbr_p = nbr[:, perm].contiguous()
        t = timeit(lambda: H.spconv_s16_run(feat, packed, kvol, cin, cout, None, nbr_p, n))
        out = H.spconv_s16_run(feat, packed, kvol, cin, cout, None, nbr_p, n)
        full = torch.empty_like(out); full[perm] = out
        if ref is None: ref = full
        same = bool(torch.equal(full, ref))
        res[name] = (t, same)
    t_sort = timeit(lambda: torch.argsort((torch.arange(n, device=dev) // (-(-n // 8))) * (1 << 27) + mask, stable=True), 5)
    t_perm = timeit(lambda: nbr[:, perm].contiguous(), 5)
    print(key, "N", n, "C", c, {k: (round(v[0], 1), v[1]) for k, v in res.items()}, "torch argsort us", round(t_sort, 1), "permute map us", round(t_perm, 1))
