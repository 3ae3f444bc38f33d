import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from recommenders_amd.layers import factorized_top_k as ftk
from recommenders_amd import _lib
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(42)
corpus = torch.randn((1_000_000, 64), generator=g, device=dev) / 8.0
bf = ftk.BruteForce(k=100).index(corpus)
def t(fn, it=15):
  for _ in range(3): fn()
  ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(it)]
  for a,b in ev: a.record(); fn(); b.record()
  torch.cuda.synchronize()
  return sorted(a.elapsed_time(b) for a,b in ev)[it//2]
for nq in (1024, 2048, 4096, 8192):
  q = torch.randn((nq, 64), generator=g, device=dev) / 8.0
  r = {"nq": nq}
  for rep in range(2):
    for w in ("512", "480", "448", "256", "1024"):
      _lib.set_option("TFRS_TOPK_WGS", w)
      bf(q)
      r.setdefault(w, []).append(round(t(lambda: bf(q)), 4))
  _lib.set_option("TFRS_TOPK_WGS", None)
  print(json.dumps(r), flush=True)
