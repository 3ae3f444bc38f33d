"""Kernel-trace target: BruteForce top-100 on BASELINE configs[1] (1 M x 64, 8192 queries), N calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.layers import factorized_top_k as ftk
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(42)
corpus = torch.randn((int(os.environ.get("ROWS", 1_000_000)), 64), generator=g, device=dev) / 8.0
queries = torch.randn((int(os.environ.get("BATCH", 8192)), 64), generator=g, device=dev) / 8.0
index = ftk.BruteForce(k=100).index(corpus)
for _ in range(int(os.environ.get("CALLS", 12))):
  out = index(queries)
torch.cuda.synchronize()
