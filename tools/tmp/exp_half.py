import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from recommenders_amd.layers import factorized_top_k as ftk
from recommenders_amd import _lib
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
d = 128
n, k, bs = 12_500_000, 100, 65536
corpus = torch.randn((n, d), generator=g, device=dev) / d ** 0.5
class Blocks:
  def __iter__(self):
    for lo in range(0, n, bs):
      yield corpus[lo:lo + bs]
st = ftk.Streaming(k=k).index_from_dataset(Blocks())
bf = ftk.BruteForce(k=k).index(corpus)
def t(fn, it=9):
  for _ in range(2): fn()
  ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(it)]
  for a,b in ev: a.record(); fn(); b.record()
  torch.cuda.synchronize()
  return sorted(a.elapsed_time(b) for a,b in ev)[it//2]
for nq in [int(x) for x in os.environ.get("NQS", "1,32,33,64,96,128").split(",")]:
  q = torch.randn((nq, d), generator=g, device=dev) / d ** 0.5
  b = bf(q)
  r = {"nq": nq}
  for rep in range(2):
    for hv in ("1", "0"):
      _lib.set_option("TFRS_RAW16_HALF", hv)
      a = st(q)
      r["same"] = r.get("same", True) and bool(torch.equal(a[0], b[0]) and torch.equal(a[1].long(), b[1].long()))
      r.setdefault("half%s_ms" % hv, []).append(round(t(lambda: st(q)), 3))
  _lib.set_option("TFRS_RAW16_HALF", None)
  print(json.dumps(r), flush=True)
