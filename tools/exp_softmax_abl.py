"""Ablation timing: forward only, B=65536 D=64, with an alternative library (argv[1])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd import _lib
if len(sys.argv) > 1:
  _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from recommenders_amd.tasks.retrieval import in_batch_softmax_loss
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
B, D = 65536, 64
q = torch.randn((B, D), generator=g, device=dev) * 0.05
c = torch.randn((B, D), generator=g, device=dev) * 0.05
for _ in range(3): in_batch_softmax_loss(q, c)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): in_batch_softmax_loss(q, c)
b.record(); torch.cuda.synchronize()
print(sys.argv[1:] or "default", "fwd us", a.elapsed_time(b) / 10 * 1e3)
