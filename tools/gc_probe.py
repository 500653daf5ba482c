"""Which reference cycles of a training step hold device memory until Python's cyclic collector runs?  One step with the collector off,
then a collection with DEBUG_SAVEALL: prints the collected tensors (by size) and, for the largest, the chain of referrers inside the garbage.
    python tools/gc_probe.py [graph|eager]"""
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    import bench
    mode = sys.argv[1] if len(sys.argv) > 1 else "eager"
    sys.argv = ["bench.py", "--no-cpu-baseline", "--no-roofline", "--no-extras"]
    args = bench.parse()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    model, teacher, frames, step = bench.setup_workload(args, "s2d_student", dev, 0)
    bench.set_mode([model, teacher], (mode == "graph", True, "" if mode == "graph" else "aux,dense,pcr,sparse", "aux,dense,pcr" if mode == "graph" else ""))
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    a0 = torch.cuda.memory_allocated()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    print("3 steps without the collector: allocated +%.1f MB" % ((torch.cuda.memory_allocated() - a0) / 1e6), flush=True)
    gc.set_debug(gc.DEBUG_SAVEALL)
    n = gc.collect()
    garbage = list(gc.garbage)
    gc.set_debug(0)
    print("collected", n, "objects;", len(garbage), "saved")
    import collections
    kinds = collections.Counter(type(o).__name__ for o in garbage)
    print(kinds.most_common(25))
    tens = [o for o in garbage if torch.is_tensor(o) and o.is_cuda]
    tens.sort(key=lambda t: -t.numel() * t.element_size())
    print("cuda tensors in the garbage:", len(tens), "total MB", sum(t.numel() * t.element_size() for t in tens) / 1e6)
    ids = {id(o) for o in garbage}
    for t in tens[:6]:
        print("--", tuple(t.shape), t.dtype, "%.1f MB" % (t.numel() * t.element_size() / 1e6))
        seen, cur = set(), t
        for depth in range(6):
            refs = [r for r in gc.get_referrers(cur) if id(r) in ids and id(r) not in seen and r is not garbage and r is not tens]
            if not refs:
                break
            r = refs[0]
            seen.add(id(r))
            desc = type(r).__name__
            if isinstance(r, dict):
                desc += " keys=" + str(list(r.keys())[:8])
            elif isinstance(r, (tuple, list)):
                desc += f" len={len(r)} of " + str([type(x).__name__ for x in r[:6]])
            elif hasattr(r, "__qualname__"):
                desc += " " + r.__qualname__
            elif hasattr(r, "__class__"):
                desc += " " + repr(r)[:100]
            print("    " * (depth + 1) + "<- " + desc)
            cur = r
    gc.garbage.clear()
    if hasattr(frames, "close"):
        frames.close()


if __name__ == "__main__":
    main()
