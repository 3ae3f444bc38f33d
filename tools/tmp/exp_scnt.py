import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from recommenders_amd.layers import embedding as emb
from recommenders_amd import _lib
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
vocab, d, n = 26_000_000, 128, 65536 * 26
table = torch.empty((vocab, d), device=dev).uniform_(-0.05, 0.05)
acc = torch.full_like(table, 0.1)
ids = torch.randint(0, vocab, (n,), generator=g, device=dev)
go = torch.randn((n, d), generator=g, device=dev)
def timeit(fn, iters=15):
  for _ in range(2): fn()
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
  for a, b in ev:
    a.record(); fn(); b.record()
  torch.cuda.synchronize()
  ts = sorted(a.elapsed_time(b) for a, b in ev)
  return ts[len(ts) // 2]
r = {"nt1": [], "nt0": []}
for rep in range(4):
  for v in ("1", "0"):
    _lib.set_option("TFRS_SCATTER_NT", v)
    r["nt" + v].append(round(timeit(lambda: emb.adagrad_sparse_update_(table, acc, go, ids, 0.5)), 4))
print(json.dumps(r))
