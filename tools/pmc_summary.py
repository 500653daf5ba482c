#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc passes (one counter per pass, collected with --kernel-trace only):
   python tools/pmc_summary.py <out.json> <dir_FETCH_SIZE> <dir_WRITE_SIZE> [kernel-substring-filter]
Writes {"kernel | grid=N": {FETCH_SIZE_KiB, WRITE_SIZE_KiB, launches}} (per-launch means, units as rocprofv3 reports them: KiB) for
kernels whose name contains the filter (default "s2d::")."""
import collections
import csv
import glob
import json
import os
import sys


def collect(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name") or r.get("Kernel-Name") or ""
            name = f"{name[:150]} | grid={r.get('Grid_Size', '?')}"   # one entry per launch geometry (= per tensor shape)
            cn, cv = r.get("Counter_Name"), r.get("Counter_Value")
            if cn is None or cv is None:
                continue
            a = acc[name][cn]
            a[0] += float(cv)
            a[1] += 1
    return acc


def main():
    out, dirs = sys.argv[1], sys.argv[2:4]
    flt = sys.argv[4] if len(sys.argv) > 4 else "s2d::"
    res = {}
    for d in dirs:
        for name, counters in collect(d).items():
            if flt not in name:
                continue
            e = res.setdefault(name, {"FETCH_SIZE_KiB": None, "WRITE_SIZE_KiB": None, "launches": 0})
            for cn, (tot, n) in counters.items():
                if cn in ("FETCH_SIZE", "WRITE_SIZE"):
                    e[cn + "_KiB"] = round(tot / max(n, 1), 1)
                    e["launches"] = max(e["launches"], n)
    json.dump({"_comment": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes); per-launch means in KiB as "
                           "reported; gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md §HBM), so "
                           "bench.py uses 2*FETCH + WRITE as the upper estimate of HBM bytes",
               "kernels": dict(sorted(res.items()))}, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -(kv[1]["FETCH_SIZE_KiB"] or 0)):
        print(f"{v['FETCH_SIZE_KiB']!s:>12} {v['WRITE_SIZE_KiB']!s:>12} {v['launches']:6d}  {k[:60]} {k[k.rfind('|'):]}")


if __name__ == "__main__":
    main()
