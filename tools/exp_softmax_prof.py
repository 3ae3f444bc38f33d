"""rocprofv3 target: in-batch softmax fwd+bwd at one batch size (argv: B D iters)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.tasks.retrieval import in_batch_softmax_loss
B, D, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
q = (torch.randn((B, D), generator=g, device=dev) * 0.05).requires_grad_(True)
c = (torch.randn((B, D), generator=g, device=dev) * 0.05).requires_grad_(True)
for _ in range(n):
  q.grad = None; c.grad = None
  in_batch_softmax_loss(q, c).backward()
torch.cuda.synchronize()
