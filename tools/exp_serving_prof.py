"""rocprofv3 target: B (argv[1]) query BruteForce calls on 1M x 64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.layers import factorized_top_k as ftk
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
corpus = torch.randn((1_000_000, 64), generator=g, device=dev) / 8.0
layer = ftk.BruteForce(k=100).index(corpus)
q = torch.randn((B, 64), generator=g, device=dev) / 8.0
for _ in range(100):
  layer(q)
torch.cuda.synchronize()
