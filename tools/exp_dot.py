"""DotInteraction forward timing at the C5 shape (development tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.layers.feature_interaction.dot_interaction import _DotInteractionFn
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
for (B, F, D) in ((131072, 101, 32), (65536, 27, 128), (131072, 32, 64)):
  x = torch.randn((B, F, D), generator=g, device=dev)
  for _ in range(3): _DotInteractionFn.apply(x, False, False)
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(10): _DotInteractionFn.apply(x, False, False)
  b.record(); torch.cuda.synchronize()
  t = a.elapsed_time(b) / 10 * 1e-3
  byts = B * F * D * 4 + B * (F * (F - 1) // 2) * 4
  print(f"B={B} F={F} D={D}: {t*1e3:.3f} ms  {byts/t/1e12:.2f} TB/s ({byts/t/8e12*100:.0f}% of 8 TB/s)  {2.0*B*F*F*D/t/1e12:.1f} TFLOP/s full-Gram")
# forward + backward
for (B, F, D) in ((131072, 101, 32), (65536, 27, 128), (131072, 32, 64)):
  x = torch.randn((B, F, D), generator=g, device=dev, requires_grad=True)
  def fb():
    x.grad = None
    _DotInteractionFn.apply(x, False, False).sum().backward()
  for _ in range(2): fb()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(5): fb()
  b.record(); torch.cuda.synchronize()
  print(f"fwd+bwd B={B} F={F} D={D}: {a.elapsed_time(b)/5:.3f} ms")
