"""Streaming over the 12.5 M x 128 shard of configs[2]: whole-call time per library switch (same box, one process).
  python tools/exp_stream_rho.py TFRS_TOPK_RHO=16 TFRS_TOPK_RHO=32 ..."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd import _lib
from recommenders_amd.layers import factorized_top_k as ftk
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
n, d, k, bs = int(os.environ.get("ROWS", 12_500_000)), int(os.environ.get("DIM", 128)), 100, 65536
corpus = torch.randn((n, d), generator=g, device=dev) / d ** 0.5
class Blocks:                       # a dataset object, iterated afresh by every call (bench.py's streaming leg): no cached image
  def __iter__(self):
    for lo in range(0, n, bs):
      yield corpus[lo:lo + bs]
st = ftk.Streaming(k=k).index_from_dataset(Blocks())
bf = ftk.BruteForce(k=k).index(corpus)
def t(fn, it=9):
  for _ in range(2): fn()
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
  for a, b in ev: a.record(); fn(); b.record()
  torch.cuda.synchronize()
  return sorted(a.elapsed_time(b) for a, b in ev)[it // 2]
batches = [int(x) for x in os.environ.get("BATCHES", "1,64,128,512").split(",")]
qs = {nq: torch.randn((nq, d), generator=g, device=dev) / d ** 0.5 for nq in batches}
want = {nq: bf(q) for nq, q in qs.items()}
for v in [None] + sys.argv[1:] + [None]:
  if v:
    for kv in v.split(","):
      _lib.set_option(kv.split("=")[0], kv.split("=")[1])
  row = {"switch": v}
  for nq, q in qs.items():
    a = st(q)
    same = bool(torch.equal(a[0], want[nq][0]) and torch.equal(a[1].long(), want[nq][1].long()))
    row["b%d" % nq] = round(t(lambda: st(q)), 3)
    row["same%d" % nq] = same
  if v:
    for kv in v.split(","):
      _lib.set_option(kv.split("=")[0], None)
  print(json.dumps(row), flush=True)
