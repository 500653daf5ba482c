"""Lists kernels whose ISA contains chains of serialised global / buffer loads: a load, an `s_waitcnt vmcnt(0)` within a few instructions, and the
next load right behind it (DESIGN rule 40: a packed value widened next to its load under register pressure; scalar constants read from global memory
between the loads of a staging loop).  Compiles every csrc/*.hip to assembly for gfx950 (no GPU needed, ~2 minutes) and prints
(serialised loads, all loads, file, kernel), worst first.

    python tools/isa_serial_loads.py [min_serialised=3]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    least = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    out = tempfile.mkdtemp(prefix="s2d_isa_")
    procs = []
    for src in sorted(glob.glob(os.path.join(ROOT, "sparse2dense_amd", "csrc", "*.hip"))):
        name = os.path.basename(src)[:-4]
        extra = ["-fno-slp-vectorize"] if name == "losses" else []   # (build.py's per-file flag)
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math", "-ffp-contract=off", *extra,
                                       "--cuda-device-only", "-S", "-o", os.path.join(out, name + ".s"), src], stderr=subprocess.DEVNULL))
    for p in procs:
        p.wait()
    rows = []
    for fn in sorted(glob.glob(os.path.join(out, "*.s"))):
        s = open(fn).read()
        for m in re.finditer(r"^(_Z[A-Za-z0-9_]+):", s, re.M):
            end = s.find("s_endpgm", m.end())
            if end < 0:
                continue
            lines = [ln.strip() for ln in s[m.end():end].split("\n") if ln.strip() and not ln.strip().startswith((";", "."))]
            serial = 0
            for k, ln in enumerate(lines):
                if ln.startswith(("global_load", "buffer_load")) and "lds" not in ln:
                    waits = [x for x in range(k + 1, min(k + 6, len(lines))) if lines[x].startswith("s_waitcnt") and "vmcnt(0)" in lines[x]]
                    if waits and any(lines[x].startswith(("global_load", "buffer_load")) for x in range(waits[0] + 1, min(waits[0] + 12, len(lines)))):
                        serial += 1
            loads = sum(1 for ln in lines if ln.startswith(("global_load", "buffer_load")))
            if serial >= least:
                rows.append((serial, loads, os.path.basename(fn), m.group(1)))
    for r in sorted(rows, reverse=True):
        print(r)


if __name__ == "__main__":
    main()
