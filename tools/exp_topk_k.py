"""Experiment: guaranteed vs statistical threshold plan over K (1 M x 64 corpus, batch 8192); fresh
queries every step, redo counts summed."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.layers import factorized_top_k as ftk
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(42)
corpus = torch.randn((1_000_000, 64), generator=g, device=dev) / 8.0
qs = [torch.randn((8192, 64), generator=g, device=dev) / 8.0 for _ in range(20)]
for k in (10, 100, 200, 256, 400, 512):
  index = ftk.BruteForce(k=k).index(corpus)
  res = {}
  for stat in ("0", "1", "0", "1"):
    os.environ["TFRS_TOPK_STAT"] = stat
    for _ in range(3): out = index(qs[0])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    redo = 0
    reasons = {}
    for q in qs:
      out = index(q)
      redo += index.last_redo_count()
      for kk, vv in index.last_redo_reasons().items(): reasons[kk] = max(reasons.get(kk, 0), vv) if kk == 'longest_list' else reasons.get(kk, 0) + vv
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / len(qs) * 1e3
    ref = torch.topk(qs[-1][:256] @ corpus.T, k)
    ok = bool((ref.indices == out[1][:256]).float().mean() > 0.999)
    res[stat] = {"ms": round(ms, 3), "redo": redo, "reasons": reasons, "idx_match": ok}
  print(json.dumps({"k": k, "guaranteed": res["0"], "statistical": res["1"]}), flush=True)
