"""PMC target: in-batch softmax fwd+bwd, B=32768 D=64, 3 iterations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.tasks.retrieval import in_batch_softmax_loss
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
B, D = 32768, 64
q = (torch.randn((B, D), generator=g, device=dev) * 0.05).requires_grad_(True)
c = (torch.randn((B, D), generator=g, device=dev) * 0.05).requires_grad_(True)
for _ in range(3):
  q.grad = None; c.grad = None
  in_batch_softmax_loss(q, c).backward()
torch.cuda.synchronize()
