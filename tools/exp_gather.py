"""Gather tuning sweep (development tool): TFRS_GATHER_VARIANT x TFRS_GATHER_WGS."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.layers import embedding as emb
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
vocab, d, n = 26_000_000, 128, 65536 * 26
table = torch.empty((vocab, d), device=dev).uniform_(-0.05, 0.05)
ids = torch.randint(0, vocab, (n,), generator=g, device=dev)
def timeit(fn, iters=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(iters): fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / iters * 1e-3
byts = n * (2 * d * 4 + 8)
for var in (0, 1, 2, 3, 4):
  for wgs in (1024, 2048, 4096, 8192, 16384):
    os.environ["TFRS_GATHER_VARIANT"] = str(var); os.environ["TFRS_GATHER_WGS"] = str(wgs)
    t = timeit(lambda: emb.gather_rows(table, ids))
    print(f"variant={var} wgs={wgs:6d} ms={t*1e3:.4f} TB/s={byts/t/1e12:.3f}", flush=True)
# reference: a plain device copy of the same number of bytes
src = torch.empty((n, d), device=dev); dst = torch.empty_like(src)
t = timeit(lambda: dst.copy_(src))
print(f"torch copy of the same output size: ms={t*1e3:.4f} TB/s={2*n*d*4/t/1e12:.3f}")
