"""Times Cross forward / forward+backward at BASELINE configs[3] shapes on both GEMM paths."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.layers.feature_interaction import Cross
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
def timeit(fn, iters=5):
  for _ in range(2): fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(iters): fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / iters
B, d = (65536, 3456) if len(sys.argv) < 2 else (int(sys.argv[1]), int(sys.argv[2]))
x0 = torch.randn((B, d), generator=g, device=dev)
layer = Cross()
for mode in ("f32", "f16"):
  os.environ["TFRS_GEMM_MODE"] = mode
  with torch.no_grad():
    tf = timeit(lambda: layer(x0, x0))
  xr = x0.clone().requires_grad_(True)
  def fb():
    xr.grad = None
    layer.zero_grad(set_to_none=True)
    layer(xr, xr).sum().backward()
  tb = timeit(fb, iters=3)
  fl = 2.0 * B * d * d
  print(f"{mode}: cross fwd {tf:.2f} ms ({fl / tf / 1e9:.0f} TFLOP/s algorithmic)  fwd+bwd {tb:.2f} ms ({4 * fl / tb / 1e9:.0f} TFLOP/s)", flush=True)
