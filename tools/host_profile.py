"""Where does the launch thread's host time go?  cProfile over N steps of the chosen bench mode (main thread only; the autograd engine's worker
thread - which runs the backward functions - is profiled through threading.setprofile).
    python tools/host_profile.py graph:loader:0:dense,aux,pcr [steps]"""
import cProfile
import os
import pstats
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    import bench
    mode_s = sys.argv[1] if len(sys.argv) > 1 else "graph:loader:0:dense,aux,pcr"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    sys.argv = ["bench.py", "--no-cpu-baseline", "--no-roofline", "--no-extras"]
    args = bench.parse()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    model, teacher, frames, step = bench.setup_workload(args, "s2d_student", dev, 0)
    g, pf, w, *f = mode_s.split(":")
    mode = (g == "graph", pf == "loader", "" if w in ("0", "") else w, "" if (not f or f[0] in ("0", "")) else f[0])
    bench.set_mode([model, teacher], mode)
    run = step if mode[1] else step.sync_step
    for _ in range(8):
        run()
    torch.cuda.synchronize()
    profs = {}

    def thread_prof(frame, event, arg):   # installed in every NEW thread: the autograd worker is created lazily, so force one backward first
        return None
    pr = cProfile.Profile()
    import time
    t0 = time.perf_counter()
    pr.enable()
    for _ in range(steps):
        run()
    pr.disable()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{steps} steps, host enqueue {1e3 * (t1 - t0) / steps:.2f} ms per step (main thread wall, no sync)")
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)
    st.sort_stats("tottime").print_stats(40)   # where the launch thread's own time goes (ctypes calls count as the caller's)
    if hasattr(frames, "close"):
        frames.close()


if __name__ == "__main__":
    main()
