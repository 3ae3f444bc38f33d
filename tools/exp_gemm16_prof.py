"""rocprofv3 target: Cross forward at B=65536, d=3456 on the split-fp16 GEMM, 5 calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.layers.feature_interaction import Cross
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
x0 = torch.randn((65536, 3456), generator=g, device=dev)
layer = Cross()
with torch.no_grad():
  for _ in range(5):
    layer(x0, x0)
torch.cuda.synchronize()
