"""BruteForce latency / throughput vs query batch size on the 1M x 64 corpus (evidence tool)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.layers import factorized_top_k as ftk
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(42)
corpus = torch.randn((1_000_000, 64), generator=g, device=dev) / 8.0
index = ftk.BruteForce(k=100).index(corpus)
for mode in ("f16", "f32"):
  os.environ["TFRS_TOPK_FILTER"] = mode
  for nq in (1, 16, 64, 256, 1024, 4096, 8192, 32768):
    q = torch.randn((nq, 64), generator=g, device=dev) / 8.0
    for _ in range(3): index(q)
    torch.cuda.synchronize()
    reps = 20 if nq <= 8192 else 5
    t0 = time.perf_counter()
    for _ in range(reps): index(q)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"filter": mode, "batch": nq, "ms": round(dt * 1e3, 4), "queries_per_s": round(nq / dt, 1)}), flush=True)
