"""one-off: where do the fp32 (MIOpen) legs of the full-size tests spend their time?  usage: python scratch/miopen_probe.py <workload> <batch> <points>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_full_size_gpu import _one_step
wl, batch, pts = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
for i in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    _one_step(wl, "f32", batch=batch, points=pts)
    torch.cuda.synchronize(); print(f"{wl} f32 B={batch} P={pts} call {i}: {time.time()-t0:.1f} s", flush=True)
t0 = time.time()
_one_step(wl, "bf16", batch=batch, points=pts)
print(f"{wl} bf16 call: {time.time()-t0:.1f} s", flush=True)
