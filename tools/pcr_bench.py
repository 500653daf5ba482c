"""PCR-head kernels at the bench shapes (B=4), a few repetitions each; run under rocprofv3 --kernel-trace --stats"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sparse2dense_amd import heads
from sparse2dense_amd.dense3d import ConvTranspose3dK4S2, PointwiseConv3d, FastBatchNorm3d

dev = "cuda"
torch.manual_seed(0)
B = 4
REP = 4


def run(mod, x, bf16=True):
    mod = mod.to(dev)
    mod.bf16_compute = bf16
    x = x.requires_grad_(True)
    for _ in range(REP):
        y = mod(x)
        y.backward(torch.ones_like(y))
        x.grad = None
    torch.cuda.synchronize()


run(PointwiseConv3d(128, 32, 1, 1, 0), torch.randn(B, 128, 5, 188, 188, device=dev))
run(ConvTranspose3dK4S2(32, 32, 4, 2, 1), torch.randn(B, 32, 5, 188, 188, device=dev))
run(ConvTranspose3dK4S2(16, 3, 4, 2, 1), torch.randn(B, 16, 10, 376, 376, device=dev))
run(FastBatchNorm3d(32, fused_relu=True), torch.randn(B, 32, 10, 376, 376, device=dev))
run(FastBatchNorm3d(3, fused_relu=True), torch.randn(B, 3, 20, 752, 752, device=dev))

# fused levels
def level(c, co, d, h, w, m):
    g = torch.randn(B, c, d, h, w, device=dev).relu_().requires_grad_(True)
    mask, off = torch.nn.Conv3d(c, 1, 1).to(dev), torch.nn.Conv3d(c, 3, 1).to(dev)
    nxt = PointwiseConv3d(c, co, 1, 1, 0).to(dev) if co else None
    if nxt is not None:
        nxt.bf16_compute = True
    cells = torch.randperm(B * d * h * w, device=dev)[:m]
    coors = torch.stack([cells // (d * h * w), (cells // (h * w)) % d, (cells // w) % h, cells % w], 1).int()
    feats = torch.randn(m, 5, device=dev)
    for _ in range(REP):
        ml, ol, z = heads.pcr_level(g, mask, off, coors, feats, next_conv=nxt)
        tot = ml + ol
        if z is not None:
            tot = tot + z.sum() * 1e-9
        tot.backward()
        g.grad = None
    torch.cuda.synchronize()


level(32, 16, 10, 376, 376, 120000)
level(3, 0, 20, 752, 752, 300000)
print("done")
