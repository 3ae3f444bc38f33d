import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.layers import factorized_top_k as ftk
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(99)
N, D, B, K = 1_000_000, 64, 8192, 100
queries = torch.randn((B, D), generator=torch.Generator(device=dev).manual_seed(7), device=dev) / 8.0
base = torch.randn((N, D), generator=g, device=dev) / 8.0
distinct = 100_000
w = 1.0 / torch.arange(1, distinct + 1, device=dev, dtype=torch.float64)
pick = torch.multinomial(w, N, replacement=True, generator=g)
c = base[:distinct][pick].contiguous()
index = ftk.BruteForce(k=K, dedup=False).index(c)
torch.cuda.synchronize(); print("index ok", flush=True)
n = int(sys.argv[1])
keep = []
for it in range(n):
  s, i = index(queries)
  if len(sys.argv) > 2: keep.append(index._last_call[0])     # keep every workspace alive
torch.cuda.synchronize(); print("back-to-back", n, "ok", index.last_redo_count(), flush=True)
