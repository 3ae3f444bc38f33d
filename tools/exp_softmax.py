"""Times the in-batch softmax forward / forward+backward (C ABI path) at a few batch sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.tasks.retrieval import in_batch_softmax_loss
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
def timeit(fn, warmup=5, iters=50):
  for _ in range(warmup): fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(iters): fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / iters * 1e3
for B, D in [(4096, 64), (16384, 64), (65536, 64), (16384, 128)]:
  q = (torch.randn((B, D), generator=g, device=dev) * 0.05).requires_grad_(True)
  c = (torch.randn((B, D), generator=g, device=dev) * 0.05).requires_grad_(True)
  def fwd():
    with torch.no_grad(): return in_batch_softmax_loss(q, c)
  def both():
    q.grad = None; c.grad = None
    in_batch_softmax_loss(q, c).backward()
  tf, tb = timeit(fwd), timeit(both, iters=20)
  print(f"B={B} D={D} fwd {tf:.1f} us ({2*B*B*D/tf/1e6:.0f} TF)  fwd+bwd {tb:.1f} us ({6*B*B*D/tb/1e6:.0f} TF)", flush=True)
