"""Per-call durations of selected kernels grouped by (name, grid, workgroup) from a rocprofv3 --kernel-trace csv directory:
finds the one slow launch that a --stats average hides (DESIGN.md rule 12).  usage: kernel_calls_by_grid.py <dir> <steps> name-substring..."""
import collections
import csv
import glob
import sys


def main():
    root, steps, keys = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
    f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if any(k in n for k in keys):
            key = (n[:70], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"))
            agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{sum(v) / steps:9.1f} us/step  n/step={len(v) / steps:6.1f} avg={sum(v) / len(v):7.1f} max={max(v):7.1f}  {k}")


if __name__ == "__main__":
    main()
