#!/usr/bin/env python
"""Summarises a `rocprofv3 --kernel-trace --stats` run of bench.py:
   python tools/prof_summary.py <dir-with-*_kernel_trace.csv> [voxelizer_launches_per_step]
Prints (a) whole-run per-kernel stats and (b) the kernel breakdown of the LAST steady-state step
(steps are delimited by the voxelizer's first kernel; launches per step = frames for the single-stage workloads,
5 for the S2D workloads with the batched voxelizer: points / dense / reconstruction at three scales)."""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    trace = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "vox_insert" in r["Kernel_Name"] or "voxb_insert" in r["Kernel_Name"]]
    starts = idx[::frames]
    a, b = starts[-2], starts[-1]
    seg = rows[a:b]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    agg = collections.defaultdict(lambda: [0, 0])
    for r in seg:
        agg[r["Kernel_Name"]][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg[r["Kernel_Name"]][1] += 1
    tot = sum(v[0] for v in agg.values())
    print(f"# last steady-state step: wall {(t1 - t0) / 1e6:.3f} ms, {len(seg)} kernels, sum of kernel time {tot / 1e6:.3f} ms")
    ours = sum(v[0] for k, v in agg.items() if "s2d::" in k or k.startswith("_ZN3s2d"))
    print(f"# hand-written s2d:: kernels {ours / 1e6:.3f} ms ({100 * ours / tot:.1f} %)")
    print("# share   total_ms  calls  avg_us   kernel")
    for n, (dur, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:220]:
        print(f"{100 * dur / tot:6.2f}% {dur / 1e6:9.3f} {c:6d} {dur / c / 1e3:8.1f}   {n[:140]}")


    # GPU idle time inside the step, attributed to the kernel that ends the gap (what the host was late to launch)
    gaps = collections.defaultdict(lambda: [0, 0])
    idle = 0
    end = int(seg[0]["End_Timestamp"])
    for r in seg[1:]:
        g = int(r["Start_Timestamp"]) - end
        if g > 0:
            idle += g
            gaps[r["Kernel_Name"]][0] += g
            gaps[r["Kernel_Name"]][1] += 1
        end = max(end, int(r["End_Timestamp"]))
    # streams (r04: weight gradients and the data pipeline run on their own HIP streams = hardware queues): kernel time per queue
    if seg and "Queue_Id" in seg[0]:
        perq = collections.defaultdict(lambda: [0, 0])
        for r in seg:
            perq[r["Queue_Id"]][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            perq[r["Queue_Id"]][1] += 1
        print("# kernel time per hardware queue (the sum exceeds the wall time when streams overlap): " +
              ", ".join(f"queue {q}: {v[0] / 1e6:.3f} ms in {v[1]} kernels" for q, v in sorted(perq.items(), key=lambda kv: -kv[1][0])))
        print(f"# device busy (union of the kernel intervals) {(t1 - t0 - idle) / 1e6:.3f} ms of {(t1 - t0) / 1e6:.3f} ms")
    print(f"# GPU idle inside the step: {idle / 1e6:.3f} ms; largest contributors (gap before the kernel):")
    for n, (dur, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f"#   {dur / 1e3:8.1f} us over {c:4d} gaps   {n[:110]}")


if __name__ == "__main__":
    main()
