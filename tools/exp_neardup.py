"""Near-duplicate clusters (rows within 2 eps of each other, not bit-identical): 1 M x 64 corpus with
one cluster of `size` rows; `aligned` of the 8192 queries point at it (development tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.layers import factorized_top_k as ftk
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(42)
corpus = torch.randn((1_000_000, 64), generator=g, device=dev) / 8.0
queries = torch.randn((8192, 64), generator=g, device=dev) / 8.0
base = torch.randn((1, 64), generator=g, device=dev) / 8.0
for size, aligned in ((0, 0), (300, 64), (300, 2048), (600, 2048), (900, 8192)):
  c = corpus.clone()
  q = queries.clone()
  if size:
    rows = torch.randperm(1_000_000, generator=g, device=dev)[:size]
    c[rows] = base + 1e-7 * torch.randn((size, 64), generator=g, device=dev)      # every row distinct
    q[:aligned] = base * (1.0 + 3.0 * torch.rand((aligned, 1), generator=g, device=dev))
  index = ftk.BruteForce(k=100).index(c)
  for _ in range(2): index(q)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(5): index(q)
  torch.cuda.synchronize()
  print(f"cluster of {size} near-duplicates, {aligned} aligned queries: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms/step, "
        f"redo {index.last_redo_count()} {index.last_redo_reasons()}", flush=True)
  del index
