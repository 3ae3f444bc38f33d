import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from recommenders_amd.layers import factorized_top_k as ftk
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
n, d, k, bs = 12_500_000, int(os.environ.get("DIM", 128)), 100, 65536
if d == 64: n = 25_000_000
corpus = torch.randn((n, d), generator=g, device=dev) / d ** 0.5
class Blocks:
  def __iter__(self):
    for lo in range(0, n, bs):
      yield corpus[lo:lo + bs]
st = ftk.Streaming(k=k).index_from_dataset(Blocks())
def t(fn, it=7):
  for _ in range(2): fn()
  ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(it)]
  for a,b in ev: a.record(); fn(); b.record()
  torch.cuda.synchronize()
  return sorted(a.elapsed_time(b) for a,b in ev)[it//2]
out = {}
for nq in [int(x) for x in os.environ.get("NQS", "1,64,128,256,257,384,512,768,1024").split(",")]:
  q = torch.randn((nq, d), generator=g, device=dev) / d ** 0.5
  out[nq] = round(t(lambda: st(q)), 3)
print(json.dumps(out), flush=True)
