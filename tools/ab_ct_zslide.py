import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse2dense_amd.dense3d import ConvTranspose3dK4S2
dev = "cuda"; torch.manual_seed(0)
big = len(sys.argv) > 1 and sys.argv[1] == "32"
m = (ConvTranspose3dK4S2(32, 32, 4, 2, 1) if big else ConvTranspose3dK4S2(16, 3, 4, 2, 1)).to(dev); m.bf16_compute = True
x = torch.randn(4, 32, 5, 188, 188, device=dev) if big else torch.randn(4, 16, 10, 376, 376, device=dev)
with torch.no_grad():
    for _ in range(3): y = m(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): y = m(x)
    b.record(); torch.cuda.synchronize()
print(f"zslide={os.environ.get('S2D_CT_ZSLIDE','1')} ct fwd {tuple(x.shape)}: {a.elapsed_time(b)/10*1e3:.1f} us  checksum {float(y.double().sum()):.6e} {float(y.double().abs().max()):.5e}")
