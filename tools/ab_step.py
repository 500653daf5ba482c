"""In-process A/B of run-time switches on the benchmarked step: ONE model, alternating blocks of timed steps per variant, so that box /
clock / allocator differences between processes (+-3 % between two `bench.py` runs on the same box) cancel.

    python tools/ab_step.py --variants "base" "S2D_PCR_STREAM=1" "side=0" --blocks 6 --steps 20

A variant is a ';'-separated list of NAME=VALUE environment settings read at run time by the package (e.g. S2D_PCR_STREAM) and of
side=<mode> (sparse2dense_amd.side.enable); "base" = nothing set.  Prints ms/step per block and the per-variant median."""
import argparse
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", nargs="+", default=["base", "side=0"])
    ap.add_argument("--blocks", type=int, default=5)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--workload", default="s2d_student")
    ap.add_argument("--batch", type=int, default=4)
    a = ap.parse_args()
    sys.argv = [sys.argv[0], "--workload", a.workload, "--batch", str(a.batch), "--no-cpu-baseline", "--no-roofline", "--no-extras"]
    args = bench.parse()
    from sparse2dense_amd import side
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    torch.backends.cudnn.benchmark = False
    torch.set_num_threads(min(bench.effective_cpu_count(), 16))   # as bench.main does
    model, teacher, frames, step = bench.setup_workload(args, a.workload, dev, 0)
    default_side = ",".join(sorted(side.MODE)) or "0"
    touched = set()

    def apply(variant):
        for k in touched:
            os.environ.pop(k, None)
        side.enable(default_side)
        if variant == "base":
            return
        for item in variant.split(";"):
            k, v = item.split("=", 1)
            if k == "side":
                side.enable(v)
            else:
                os.environ[k] = v
                touched.add(k)

    for v in a.variants:   # every variant's allocator / plan state settles before anything is timed
        apply(v)
        for _ in range(8):
            step()
    torch.cuda.synchronize()
    res = {v: [] for v in a.variants}
    for b in range(a.blocks):
        for v in a.variants:
            apply(v)
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t0) / a.steps * 1e3)
    for v in a.variants:
        print(f"{v:40s} median {statistics.median(res[v]):7.3f} ms   blocks " + " ".join(f"{x:6.2f}" for x in res[v]), flush=True)
    if hasattr(frames, "close"):
        frames.close()


if __name__ == "__main__":
    main()
