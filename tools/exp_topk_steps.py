"""rocprofv3 target: 6 BruteForce calls of the BASELINE configs[1] batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.layers import factorized_top_k as ftk
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(42)
corpus = torch.randn((1_000_000, 64), generator=g, device=dev) / 8.0
queries = torch.randn((8192, 64), generator=g, device=dev) / 8.0
index = ftk.BruteForce(k=100).index(corpus)
for _ in range(6):
  index(queries)
torch.cuda.synchronize()
