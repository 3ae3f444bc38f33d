"""Host-side cost of one BruteForce.call at the headline shapes (development tool): wall time per
call when issuing back to back without synchronising vs the GPU time per call, and a cProfile of the
issue path."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.layers import factorized_top_k as ftk
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(42)
corpus = torch.randn((1_000_000, 64), generator=g, device=dev) / 8.0
queries = torch.randn((8192, 64), generator=g, device=dev) / 8.0
index = ftk.BruteForce(k=100).index(corpus)
for _ in range(5):
  index(queries)
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
  index(queries)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"issue {t_issue / n * 1e3:.3f} ms/call, total {t_all / n * 1e3:.3f} ms/call", flush=True)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
  index(queries)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
