"""Host study (numpy, no GPU): how many (workgroup, K-step) and (16-row MFMA tile, K-step) units of the SubM gather implicit GEMM are
active when the output rows are processed in natural order vs sorted by their 27-bit neighbour mask (globally or inside chunks that keep
an XCD-local working set), on the 4-frame bench scene.  Numbers quoted in HISTORY.md section 5 ("sparse_gemm") and DESIGN.md section 8."""
import numpy as np, sys, time
sys.path.insert(0, '/root/repo')
from sparse2dense_amd import scene

def voxel_coords(points, vs=(0.1, 0.1, 0.15), rng=(-75.2, -75.2, -2, 75.2, 75.2, 4)):
    lo = np.array(rng[:3]); hi = np.array(rng[3:]); vs = np.array(vs)
    m = np.all((points[:, :3] >= lo) & (points[:, :3] < hi), axis=1)
    c = np.floor((points[m, :3] - lo) / vs).astype(np.int64)   # x,y,z
    return np.unique(c[:, ::-1], axis=0)   # z,y,x

def strided(coords, shape, pad):
    out = []
    oshape = tuple((s + 2 * p - 3) // 2 + 1 for s, p in zip(shape, pad))
    for kz in range(3):
        for ky in range(3):
            for kx in range(3):
                k = np.array([kz, ky, kx]); p = np.array(pad)
                num = coords[:, 1:] + p - k
                ok = np.all(num % 2 == 0, axis=1)
                o = num[ok] // 2
                ok2 = np.all((o >= 0) & (o < np.array(oshape)), axis=1)
                out.append(np.concatenate([coords[ok][ok2][:, :1], o[ok2]], axis=1))
    return np.unique(np.concatenate(out), axis=0), oshape

def subm_masks(coords, shape):
    # coords [N,4] (b,z,y,x) sorted lexicographically
    key = ((coords[:, 0] * (shape[0] + 2) + coords[:, 1] + 1) * (shape[1] + 2) + coords[:, 2] + 1) * (shape[2] + 2) + coords[:, 3] + 1
    order = np.argsort(key); skey = key[order]
    masks = np.zeros(len(coords), dtype=np.int64)
    k = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                nk = key + (dz * (shape[1] + 2) + dy) * (shape[2] + 2) + dx
                pos = np.searchsorted(skey, nk)
                pos[pos >= len(skey)] = len(skey) - 1
                hit = skey[pos] == nk
                masks |= hit.astype(np.int64) << k
                k += 1
    return masks

def evaluate(masks, name, bm=128):
    n = len(masks)
    pad = (-n) % bm
    m = np.concatenate([masks, np.zeros(pad, dtype=np.int64)])
    bits = ((m[:, None] >> np.arange(27)) & 1).astype(np.int32)        # [n,27]
    pairs = bits.sum()
    t16 = bits.reshape(-1, 16, 27).max(1)                               # 16-row tiles active
    wg = bits.reshape(-1, bm, 27).max(1)                                # workgroup steps active
    print(f"  {name:28s} bm={bm}: density {pairs/(n*27):.3f}  16-row tiles active {t16.mean():.3f}  (rows in active tiles per pair {t16.sum()*16/pairs:.2f})  wg steps active {wg.mean():.3f}")
    return t16.mean(), wg.mean()

pts = [scene.make_scene(150000, seed=20240928 + b)["points"] for b in range(4)]
cs = []
for b, p in enumerate(pts):
    c = voxel_coords(p)
    cs.append(np.concatenate([np.full((len(c), 1), b), c], axis=1))
coords = np.concatenate(cs); shape = (41, 1504, 1504)
stages = [("stage1 16ch", None), ("stage2 32ch", (1, 1, 1)), ("stage3 64ch", (1, 1, 1)), ("stage4 128ch", (0, 1, 1))]
for name, pad in stages:
    if pad is not None:
        coords, shape = strided(coords, shape, pad)
    masks = subm_masks(coords, shape)
    print(name, "N =", len(coords), "shape", shape, "pairs", int(sum(bin(m).count('1') for m in masks[:1000])) / 1000 * len(masks))
    for bm in (64, 128):
        evaluate(masks, "natural (b,z,y,x) order", bm)
        o = np.argsort(masks, kind="stable")
        evaluate(masks[o], "sorted by mask", bm)
        # sort by mask with bit-reversed significance / popcount first
        pc = np.array([bin(x).count("1") for x in masks])
        o2 = np.lexsort((masks, pc))
        evaluate(masks[o2], "sorted by popcount,mask", bm)

print("\n==== chunked sort, OPS grouping ====")
def evaluate2(masks, name, bm, ops):
    n = len(masks); pad = (-n) % bm
    m = np.concatenate([masks, np.zeros(pad, dtype=np.int64)])
    ks = ((27 + ops - 1) // ops) * ops
    bits = ((m[:, None] >> np.arange(ks)) & 1).astype(np.int32)
    g = bits.reshape(len(m), ks // ops, ops).max(2)                     # step-level activity per row
    t16 = g.reshape(-1, 16, ks // ops).max(1)
    wg = g.reshape(-1, bm, ks // ops).max(1)
    print(f"  {name:34s} bm={bm} ops={ops}: tile-steps active {t16.mean():.3f}  wg steps active {wg.mean():.3f}")

coords = np.concatenate(cs); shape = (41, 1504, 1504)
cfg = {"stage1 16ch": (64, 4), "stage2 32ch": (64, 2), "stage3 64ch": (64, 1), "stage4 128ch": (128, 1)}
for name, pad in stages:
    if pad is not None:
        coords, shape = strided(coords, shape, pad)
    masks = subm_masks(coords, shape)
    bm, ops = cfg[name]
    print(name, "N =", len(coords))
    evaluate2(masks, "natural", bm, ops)
    for chunk in (1024, 2048, 4096, 8192, 32768, 1 << 30):
        o = np.concatenate([c0 + np.argsort(masks[c0:c0 + chunk], kind="stable") for c0 in range(0, len(masks), chunk)])
        evaluate2(masks[o], f"mask-sorted in chunks of {chunk}", bm, ops)
