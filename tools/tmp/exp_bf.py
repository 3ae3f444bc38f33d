import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from recommenders_amd.layers import factorized_top_k as ftk
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
out = {}
for n, d in ((12_500_000, 128), (25_000_000, 64), (1_000_000, 64)):
  corpus = torch.randn((n, d), generator=g, device=dev) / d ** 0.5
  bf = ftk.BruteForce(k=100).index(corpus)
  def t(fn, it=9):
    for _ in range(2): fn()
    ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for a,b in ev: a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a,b in ev)[it//2]
  for nq in (1, 64, 256, 512, 1024, 8192):
    q = torch.randn((nq, d), generator=g, device=dev) / d ** 0.5
    out["%dx%d/%d" % (n // 1000000, d, nq)] = round(t(lambda: bf(q)), 3)
  del bf, corpus
print(json.dumps(out), flush=True)
