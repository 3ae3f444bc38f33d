"""BruteForce top-100 on a CLUSTERED corpus (the fp16-prefiltered path is data dependent: the
headline bench uses i.i.d. Gaussian embeddings).  1M x 64 corpus drawn from 1000 Gaussian clusters
with Zipf(1) popularity (cluster spread 0.35 of the centre norm), queries drawn from the same
mixture; corpus rows shuffled, and grouped by cluster (every cluster one contiguous row range, the
adversarial order for a threshold taken from sampled stages).  Reports queries/s, filter-pass time
and the number of queries that needed the exact-redo path.  JSON lines."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd import _lib
from recommenders_amd.layers import factorized_top_k as ftk

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
N, D, B, K, C = 1_000_000, 64, 8192, 100, 1000
lib = _lib.load()
centers = torch.randn((C, D), generator=g, device=dev) / (D ** 0.5)
pop = 1.0 / torch.arange(1, C + 1, device=dev, dtype=torch.float32)
pop = pop / pop.sum()
def draw(n):
  cl = torch.multinomial(pop, n, replacement=True, generator=g)
  return centers[cl] + 0.35 * torch.randn((n, D), generator=g, device=dev) / (D ** 0.5), cl
corpus, ccl = draw(N)
queries, _ = draw(B)
iid = torch.randn((N, D), generator=g, device=dev) / (D ** 0.5)
for name, cand in (("iid gaussian (bench.py data)", iid), ("clustered, shuffled rows", corpus),
                   ("clustered, rows grouped by cluster", corpus[torch.argsort(ccl)])):
  q = queries if "clustered" in name else torch.randn((B, D), generator=g, device=dev) / (D ** 0.5)
  index = ftk.BruteForce(k=K).index(cand)
  for _ in range(3):
    index(q)
  torch.cuda.synchronize()
  lib.tfrs_profile_enable(1)
  t0 = time.perf_counter()
  steps = 20
  for _ in range(steps):
    index(q)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / steps
  ms, n, fl = ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
  lib.tfrs_profile_read_kind(1, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl))
  lib.tfrs_profile_read(None, None, None)
  lib.tfrs_profile_enable(0)
  print(json.dumps({"corpus": name, "queries_per_s": B / dt, "ms_per_step": dt * 1e3,
                    "filter_pass_ms": ms.value / steps, "redo_queries": index.last_redo_count()}), flush=True)
  del index
