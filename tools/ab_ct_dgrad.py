import os, sys, torch
sys.path.insert(0, os.getcwd())
from sparse2dense_amd.dense3d import ConvTranspose3dK4S2
dev = "cuda"; torch.manual_seed(0)
m = ConvTranspose3dK4S2(16, 3, 4, 2, 1).to(dev); m.bf16_compute = True
for p in m.parameters(): p.requires_grad_(False)
x = torch.randn(4, 16, 10, 376, 376, device=dev, requires_grad=True)
y = m(x)
g = torch.randn_like(y)
for _ in range(3): dx, = torch.autograd.grad(y, x, g, retain_graph=True)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): dx, = torch.autograd.grad(y, x, g, retain_graph=True)
b.record(); torch.cuda.synchronize()
print(f"{sys.argv[1] if len(sys.argv) > 1 else ''} ct dgrad 16<-3: {a.elapsed_time(b)/10*1e3:.1f} us  checksum {float(dx.double().sum()):.6e} {float(dx.double().abs().max()):.5e}")
