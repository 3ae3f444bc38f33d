import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from recommenders_amd.layers import factorized_top_k as ftk
from recommenders_amd import _lib
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
def t(fn, it=9):
  for _ in range(2): fn()
  ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(it)]
  for a,b in ev: a.record(); fn(); b.record()
  torch.cuda.synchronize()
  return sorted(a.elapsed_time(b) for a,b in ev)[it//2]
for n, d in ((12_500_000, 128), (2_000_000, 128)):
  corpus = torch.randn((n, d), generator=g, device=dev) / d ** 0.5
  bf = ftk.BruteForce(k=100).index(corpus)
  for nq in (1, 64, 512, 1024, 8192):
    q = torch.randn((nq, d), generator=g, device=dev) / d ** 0.5
    r = {"rows": n, "dim": d, "nq": nq}
    ref = None
    for rep in range(2):
      for w in ("512", "256", "1024"):
        _lib.set_option("TFRS_TOPK_WGS", w)
        a = bf(q)
        if ref is None: ref = a
        r["same"] = r.get("same", True) and bool(torch.equal(a[0], ref[0]) and torch.equal(a[1], ref[1]))
        r.setdefault(w, []).append(round(t(lambda: bf(q)), 4))
    _lib.set_option("TFRS_TOPK_WGS", None)
    print(json.dumps(r), flush=True)
  del bf, corpus
