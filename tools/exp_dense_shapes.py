"""dense() forward time per shape on both GEMM paths (routing-rule check)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.layers.feature_interaction import dcn
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
shapes = [(131072, 5082, 1024), (131072, 5080, 1024), (131072, 1024, 512), (131072, 512, 256),
          (65536, 3456, 1024), (65536, 512, 256), (65536, 256, 128), (5082, 131072, 1024), (256, 65536, 128)]
for m, k, n in shapes:
  x = torch.randn((m, k), generator=g, device=dev)
  w = torch.randn((k, n), generator=g, device=dev)
  res = []
  for mode in ("f32", "f16"):
    os.environ["TFRS_GEMM_MODE"] = mode
    res.append(timeit(lambda: dcn.dense(x, w)))
  print(f"m={m} k={k} n={n}: f32 {res[0]:.2f} ms  f16 {res[1]:.2f} ms  ratio {res[0] / res[1]:.2f}", flush=True)
  del x, w
