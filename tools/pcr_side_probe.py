"""Where does the first difference appear when the PCR head's weight gradients run on the eager side stream (side mode "pcr")?
Records, in call order, clones of the operands / results of the PCR backward nodes in a single-stream run and in side-stream runs and
reports the first record that differs (and how many elements).   python tools/pcr_side_probe.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from sparse2dense_amd import dense2d, dense3d, heads, hip_ops, side, waymo_configs
from sparse2dense_amd.data import SyntheticFrames
from sparse2dense_amd.registry import build_detector

REC = []
LIGHT = os.environ.get("PROBE_LIGHT", "0") == "1"   # checksums instead of clones (less perturbation of the timing)


def note(tag, t):
    if t is None or not torch.is_tensor(t) or os.environ.get("NO_NOTES") == "1":
        return
    REC.append((tag, t.detach().double().abs().sum() if LIGHT else t.detach().clone()))


_lvl_bwd = heads._PcrLevelNormFn.backward
_ct_bwd = dense3d._ConvT3dFn.backward
_bn_bwd = dense3d.bncm_backward
_fin = hip_ops.bn1d_finalize_bwd


def lvl_bwd(ctx, go_mask, go_off, dz=None):
    note("level.in.dz", dz)
    out = _lvl_bwd(ctx, go_mask, go_off, dz)
    note("level.out.dy", out[0])
    note("level.out.dgamma", out[1])
    note("level.out.headgrads", out[3])
    return out


def ct_bwd(ctx, dout, *a, **k):
    note("convT.in.dout", dout)
    out = _ct_bwd(ctx, dout, *a, **k)
    note("convT.out.dx", out[0])
    return out


def bn_bwd(dy, x, *a, **k):
    note("bn3d.in.dy", dy)
    note("bn3d.in.x", x)
    out = _bn_bwd(dy, x, *a, **k)
    note("bn3d.out.dx", out[0])
    note("bn3d.out.dgamma", out[1])
    return out


def fin(sums, *a, **k):
    note("finalize.in.sums", sums)
    return _fin(sums, *a, **k)


heads._PcrLevelNormFn.backward = staticmethod(lvl_bwd)
dense3d._ConvT3dFn.backward = staticmethod(ct_bwd)
dense3d.bncm_backward = bn_bwd
hip_ops.bn1d_finalize_bwd = fin


def poison(dev, value=float("nan")):
    """fill the caching allocator's free lists with `value`: whatever torch.empty hands out next carries it - a kernel that reads memory it (or
    its producer) never wrote shows up as a changed / non-finite result"""
    held = []
    for nbytes in [512 * k for k in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024, 1536, 2000)] * 6:
        held.append(torch.full((nbytes // 4,), value, dtype=torch.float32, device=dev))
    for mb in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024) * 2:
        held.append(torch.full((mb << 18,), value, dtype=torch.float32, device=dev))
    torch.cuda.synchronize()
    del held


def run(mode, poison_value=None):
    REC.clear()
    side.enable(mode)
    dense2d.clear_pack_cache()
    hip_ops.set_sparse_compute_dtype("s16")
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    model = build_detector(waymo_configs.s2d_student())
    model.dense_dtype = torch.bfloat16
    model.use_channels_last()
    model = model.to(dev).train()
    frames = SyntheticFrames(1, n_points=12000, seed=5, distill=True, device=dev)
    ex = frames.example()
    if poison_value is not None and os.environ.get("POISON_FWD", "1") == "1":
        poison(dev, poison_value)
    out = model(ex, return_loss=True, return_feature=True)
    loss = sum(out[0]["loss"]) + out[4] + out[5]
    if poison_value is not None:
        poison(dev, poison_value)
    loss.backward()
    side.join() if hasattr(side, "join") else None
    torch.cuda.synchronize()
    side.enable(False)
    recs = [(t, v.clone()) for t, v in REC]
    recs.append(("loss", loss.detach().clone()))
    for n, p in model.named_parameters():
        if p.grad is not None:
            recs.append(("grad:" + n, p.grad.detach().clone()))
    return recs


ref = run("0")
print("records per run:", len(ref), [t for t, _ in ref][:40])
POISONS = [float(v) for v in os.environ.get("POISON", "").split(",") if v]
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    got = run("0", POISONS[rep % len(POISONS)]) if POISONS else run("pcr")
    assert [t for t, _ in got] == [t for t, _ in ref]
    first = None
    for i, ((t, a), (_, b)) in enumerate(zip(got, ref)):
        if not torch.equal(a, b):
            nd = int((a != b).sum()) if a.dim() else 1
            d = (a.double() - b.double()).abs()
            print(f"rep {rep}: record {i} {t} differs: {nd} of {a.numel()} elements, max abs {float(d.max()):.3e} (max |ref| {float(b.double().abs().max()):.3e})", flush=True)
            if first is None:
                first = i
            if sum(1 for _ in range(1)) and i > (first or 0) + int(os.environ.get("PROBE_MORE", "3")):
                break
    if first is None:
        print(f"rep {rep}: all {len(ref)} records equal", flush=True)
