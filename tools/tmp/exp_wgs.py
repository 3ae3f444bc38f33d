import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from recommenders_amd.layers import factorized_top_k as ftk
from recommenders_amd import _lib
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
d = int(os.environ.get("DIM", 128))
n, k, bs = (12_500_000 * 128 // d), 100, 65536
corpus = torch.randn((n, d), generator=g, device=dev) / d ** 0.5
class Blocks:
  def __iter__(self):
    for lo in range(0, n, bs):
      yield corpus[lo:lo + bs]
st = ftk.Streaming(k=k).index_from_dataset(Blocks())
def t(fn, it=9):
  for _ in range(2): fn()
  ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(it)]
  for a,b in ev: a.record(); fn(); b.record()
  torch.cuda.synchronize()
  return sorted(a.elapsed_time(b) for a,b in ev)[it//2]
for nq in [int(x) for x in os.environ.get("NQS", "1,64,128,512").split(",")]:
  q = torch.randn((nq, d), generator=g, device=dev) / d ** 0.5
  r = {"nq": nq}
  for rep in range(2):
    for w in ("512", "256", "384", "768", "1024"):
      _lib.set_option("TFRS_TOPK_WGS", w)
      try:
        st(q)
        r.setdefault(w, []).append(round(t(lambda: st(q)), 3))
      except Exception as e:
        r[w] = str(e)[:60]
  _lib.set_option("TFRS_TOPK_WGS", None)
  print(json.dumps(r), flush=True)
