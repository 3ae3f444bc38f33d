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
torch.cuda.synchronize(); print("corpus ok", flush=True)
index = ftk.BruteForce(k=K, dedup=False).index(c)
torch.cuda.synchronize(); print("index ok", flush=True)
for mode in sys.argv[1:]:
  os.environ["TFRS_TOPK_FILTER"] = mode
  for it in range(3):
    s, i = index(queries)
    torch.cuda.synchronize(); print(mode, it, "query ok", index.last_redo_count() if mode == "f16" else "", index.last_redo_reasons() if mode == "f16" else "", flush=True)
