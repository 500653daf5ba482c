#!/usr/bin/env python
"""Idle-gap attribution for the last steady-state step of a rocprofv3 kernel trace:
   python tools/gap_analysis.py <dir> [voxelizer launches per step]
Lists the largest gaps between consecutive kernels with the kernels before/after them."""
import csv, glob, os, sys, collections
d = sys.argv[1]; frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4
trace = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "vox_insert" in r["Kernel_Name"]]
starts = idx[::frames]
seg = rows[starts[-2]:starts[-1]]
gaps = []
for a, b in zip(seg[:-1], seg[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    gaps.append((g, a["Kernel_Name"][:70], b["Kernel_Name"][:70]))
tot = sum(g for g, _, _ in gaps if g > 0)
print(f"# total idle {tot/1e6:.3f} ms over {len(gaps)} kernel boundaries")
hist = collections.Counter()
for g, _, _ in gaps:
    hist["<2us" if g < 2000 else "2-5us" if g < 5000 else "5-20us" if g < 20000 else "20-100us" if g < 100000 else ">100us"] += max(g, 0)
for k, v in hist.items():
    print(f"#   gaps {k:9s}: {v/1e6:7.3f} ms")
for g, a, b in sorted(gaps, key=lambda t: -t[0])[:25]:
    print(f"{g/1e3:9.1f} us   after {a}   before {b}")
