"""Large-vocabulary scatter-add / fused sparse Adagrad at BASELINE configs[3] shapes (1.7M looked-up
rows of dim 128 from 26 x 1M-row tables): time incl. the library's own radix sort, vs HBM."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.layers import embedding as emb
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
vocab, d, n = 26_000_000, 128, 65536 * 26
table = torch.empty((vocab, d), device=dev).uniform_(-0.05, 0.05)
acc = torch.full_like(table, 0.1)
ids = torch.randint(0, vocab, (n,), generator=g, device=dev)
go = torch.randn((n, d), generator=g, device=dev)
def timeit(fn, iters=10):
  for _ in range(2): fn()
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
  for a, b in ev:
    a.record(); fn(); b.record()
  torch.cuda.synchronize()
  ts = sorted(a.elapsed_time(b) for a, b in ev)
  return ts[len(ts) // 2] * 1e-3
t = timeit(lambda: emb.adagrad_sparse_update_(table, acc, go, ids, 0.5))
uniq = int(torch.unique(ids).numel())
byts = n * d * 4 + 4 * uniq * d * 4 + n * 8        # grad rows read + table/accum rows read+write + ids
print(json.dumps({"op": "sparse_adagrad (own radix sort + fused segmented update)", "rows": n, "unique": uniq,
                  "dim": d, "ms": t * 1e3, "gbps": byts / t / 1e9, "frac_hbm_peak": byts / t / 8e12,
                  "algorithmic_bytes": byts}))
