"""A noisy neighbour for robustness runs of bench.py: N processes of large fp32 matmuls / memory copies on the host cores (cache- and
memory-bandwidth-heavy, unlike a spin loop).   python tools/host_noise.py <processes> <threads each> <seconds>"""
import multiprocessing as mp
import sys
import time


def work(threads, seconds):
    import torch
    torch.set_num_threads(threads)
    a = torch.randn(3072, 3072)
    big = torch.randn(64 << 20)
    t0 = time.time()
    while time.time() - t0 < seconds:
        a = (a @ a).clamp_(-1, 1)
        big = big.flip(0) * 1.0001


if __name__ == "__main__":
    n, th, sec = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
    ps = [mp.Process(target=work, args=(th, sec)) for _ in range(n)]
    [p.start() for p in ps]
    [p.join() for p in ps]
