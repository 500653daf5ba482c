import sys, time, torch
sys.path.insert(0, "/root/repo")
from sparse2dense_amd import scene
from sparse2dense_amd.data import SyntheticFrames
dev = torch.device("cuda:0")
frames = SyntheticFrames(4, n_points=150000, seed=20240928, distill=True, device=dev, beam_jitter=scene.WAYMO_BEAM_JITTER)
for _ in range(3):
    frames.example()
torch.cuda.synchronize()
import cProfile, pstats
N = 20
t0 = time.perf_counter()
for _ in range(N):
    ex = frames.example()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"example(): host {1e3 * (t1 - t0) / N:.3f} ms per call (enqueue + its host reads), then {1e3 * (t2 - t1):.3f} ms to drain")
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    ex = frames.example()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(18)
