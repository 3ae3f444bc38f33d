"""BruteForce latency: eager call vs HIP-graph replay (make_graphed_call), 1M x 64 corpus."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.layers import factorized_top_k as ftk
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
N, D, K = 1_000_000, 64, 100
corpus = torch.randn((N, D), generator=g, device=dev) / 8.0
layer = ftk.BruteForce(k=K).index(corpus)
def timeit(fn, iters):
  for _ in range(5): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(iters): fn()
  torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters
for B in (1, 8, 64, 512, 8192):
  q = torch.randn((B, D), generator=g, device=dev) / 8.0
  graphed = layer.make_graphed_call(q)
  te = timeit(lambda: layer(q), 200 if B < 8192 else 30)
  tg = timeit(lambda: graphed(q), 200 if B < 8192 else 30)
  print(json.dumps({"op": "BruteForce top-100, 1M x 64", "batch": B, "eager_ms": te * 1e3,
                    "graphed_ms": tg * 1e3, "eager_qps": B / te, "graphed_qps": B / tg}), flush=True)
