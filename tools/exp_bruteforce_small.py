"""Kernel-trace target: BruteForce top-100 over a resident 12.5M x 128 index (one GPU's shard of BASELINE.json
configs[2]) at ONE small batch size per run (BATCH=1 | 64 | ...): which kernels a small-batch call is made of."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.layers import factorized_top_k as ftk

dev = torch.device("cuda", 0)
n, d, k = int(os.environ.get("ROWS", 12_500_000)), int(os.environ.get("DIM", 128)), 100
nq = int(os.environ.get("BATCH", 1))
g = torch.Generator(device=dev).manual_seed(1)
corpus = torch.randn((n, d), generator=g, device=dev) / (d ** 0.5)
q = torch.randn((nq, d), generator=g, device=dev) / (d ** 0.5)
bf = ftk.BruteForce(k=k).index(corpus)
del corpus
for _ in range(int(os.environ.get("CALLS", 6))):
  bf(q)
torch.cuda.synchronize()
