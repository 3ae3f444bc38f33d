"""Zipf-duplicated corpus (bench.py robustness block) on the default fp16-prefiltered path and on the
all-f32 rounds (development tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.layers import factorized_top_k as ftk
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(99)
N, D, B, K = 1_000_000, 64, 8192, 100
queries = torch.randn((B, D), generator=torch.Generator(device=dev).manual_seed(7), device=dev) / 8.0
base = torch.randn((N, D), generator=g, device=dev) / 8.0
for distinct in (100_000, 10_000):
  w = 1.0 / torch.arange(1, distinct + 1, device=dev, dtype=torch.float64)
  pick = torch.multinomial(w, N, replacement=True, generator=g)
  c = base[:distinct][pick].contiguous()
  index = ftk.BruteForce(k=K, dedup=(os.environ.get("TFRS_EXP_DEDUP", "0") == "1")).index(c)
  for mode in ("f16", "f32"):
    os.environ["TFRS_TOPK_FILTER"] = mode
    for _ in range(2): index(queries)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): s, i = index(queries)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3 * 1e3
    print(f"zipf over {distinct} distinct rows, filter={mode}: {dt:.2f} ms/step, redo {index.last_redo_count() if mode == 'f16' else '-'}", flush=True)
  del index
