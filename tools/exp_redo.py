"""Cost of the exact-recompute fallback: 1M x 64 corpus, batch 8192, with bursts of strong
matches planted for a few queries so that their survivor segments overflow (development tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.layers import factorized_top_k as ftk
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(42)
corpus = torch.randn((1_000_000, 64), generator=g, device=dev) / 8.0
queries = torch.randn((8192, 64), generator=g, device=dev) / 8.0
for nflag in (0, 1, 8, 64, 512):
  c = corpus.clone()
  for f in range(nflag):                       # 300 near-copies of query f in unsampled stages
    lo = 128 * (401 + 8 * f)
    qf = queries[f]
    # rows of ordinary norm that score ~0.9 for query f only (ordinary scores for the others)
    c[lo:lo + 300] = qf / (qf * qf).sum() * (0.85 + 0.1 * torch.rand((300, 1), generator=g, device=dev))
  index = ftk.BruteForce(k=100).index(c)
  for _ in range(2): index(queries)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(5): index(queries)
  torch.cuda.synchronize()
  print(f"flagged queries ~{nflag}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms/step, redo "
        f"{index.last_redo_count()} {index.last_redo_reasons()}", flush=True)
  del index
