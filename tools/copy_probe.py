"""Which Python lines issue the large aten::copy_ / add kernels of a training step?  torch.profiler with stacks over one eager S2D-student step (B = 4)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile

import bench

GRAPH = len(sys.argv) > 1 and sys.argv[1] == "graph"   # python tools/copy_probe.py graph: the graphed mode's eager remainder (fills / small copies too)
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-roofline", "--no-extras", "--no-prefetch"] + ([] if GRAPH else ["--no-graph"])
args = bench.parse()
dev = torch.device("cuda:0")
model, teacher, frames, step = bench.setup_workload(args, "s2d_student", dev, 0)
if GRAPH:
    bench.set_mode([model, teacher], (True, False, "sparse", "aux,dense,pcr"))
run = step.sync_step
for _ in range(6 if GRAPH else 4):
    run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    run()
    torch.cuda.synchronize()
rows = []
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::add", "aten::add_", "aten::contiguous", "aten::clone", "aten::cat", "aten::gelu", "aten::gelu_backward", "aten::mul",
                   "aten::fill_", "aten::zero_", "aten::zeros", "aten::to", "aten::_to_copy") and ev.device_time_total > (0.5 if GRAPH else 4):
        st = [s for s in (ev.stack or []) if "sparse2dense_amd" in s or "bench.py" in s][:3]
        rows.append((ev.device_time_total, ev.name, str(ev.input_shapes)[:80], " <- ".join(s.split("/")[-1] for s in st)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"{len(rows)} torch elementwise / copy ops with device time, {tot / 1e3:.3f} ms in total")
for r in rows[:45]:
    print(f"{r[0]:8.1f} us  {r[1]:<22s} {r[2]:<82s} {r[3]}")
