"""Streaming over 12.5 M x 128 in 65536-row blocks at several batch sizes (median of 7 calls each, default switches),
equality with BruteForce over the resident corpus -- one line per batch; run once per library for a same-box A/B."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.layers import factorized_top_k as ftk
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
n, d, k, bs = 12_500_000, 128, 100, 65536
corpus = torch.randn((n, d), generator=g, device=dev) / d ** 0.5
class Blocks:
  def __iter__(self):
    for lo in range(0, n, bs):
      yield corpus[lo:lo + bs]
st = ftk.Streaming(k=k).index_from_dataset(Blocks())
bf = ftk.BruteForce(k=k).index(corpus)
def t(fn, it=7):
  for _ in range(2): fn()
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
  for a, b in ev: a.record(); fn(); b.record()
  torch.cuda.synchronize()
  return sorted(a.elapsed_time(b) for a, b in ev)[it // 2]
out = {}
for nq in [int(x) for x in (sys.argv[1:] or ["1", "64", "128", "256", "384", "512", "640"])]:
  q = torch.randn((nq, d), generator=g, device=dev) / d ** 0.5
  ms = t(lambda: st(q))
  a, b = st(q), bf(q)
  out[nq] = [round(ms, 3), bool(torch.equal(a[0], b[0]) and torch.equal(a[1].long(), b[1].long()))]
print(json.dumps(out), flush=True)
