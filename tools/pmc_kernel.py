#!/usr/bin/env python
"""Per-launch means of rocprofv3 --pmc counters for kernels whose name contains a filter:
   python tools/pmc_kernel.py <filter> <dir> [<dir> ...]
(each dir = one `rocprofv3 --pmc ... --kernel-trace --output-format csv -d <dir>` pass)."""
import collections
import csv
import glob
import os
import sys


def main():
    flt, dirs = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                name = r.get("Kernel_Name") or ""
                if flt not in name:
                    continue
                key = f"{name[:110]} | grid={r.get('Grid_Size', '?')} vgpr={r.get('VGPR_Count', '?')}+{r.get('Accum_VGPR_Count', '?')} lds={r.get('LDS_Block_Size', '?')}"
                a = acc[key][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    for key, cs in acc.items():
        print(key)
        for cn, (tot, n) in sorted(cs.items()):
            print(f"    {cn:42s} {tot / n:16.1f}   ({n} launches)")


if __name__ == "__main__":
    main()
