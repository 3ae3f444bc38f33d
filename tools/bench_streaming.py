"""C3-shaped measurement (per-GPU shard of BASELINE.json configs[2]): Streaming top-100 over a
12.5M x 128 shard in blocks of 65536 rows, and BruteForce over the same shard.  Evidence tool."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.layers import factorized_top_k as ftk

dev = torch.device("cuda", 0)
n, d, k, bs = int(os.environ.get("ROWS", 12_500_000)), 128, 100, 65536
g = torch.Generator(device=dev).manual_seed(1)
corpus = torch.randn((n, d), generator=g, device=dev) / (d ** 0.5)


class Blocks:
  def __iter__(self):
    for lo in range(0, n, bs):
      yield corpus[lo:lo + bs]


for nq in [int(x) for x in os.environ.get("BATCHES", "8192,64,1").split(",")]:
  q = torch.randn((nq, d), generator=g, device=dev) / (d ** 0.5)
  bf = ftk.BruteForce(k=k).index(corpus)
  for _ in range(2):
    out_b = bf(q)
  torch.cuda.synchronize()
  t0 = time.perf_counter(); reps = 3
  for _ in range(reps):
    out_b = bf(q)
  torch.cuda.synchronize()
  tb = (time.perf_counter() - t0) / reps
  del bf
  st = ftk.Streaming(k=k).index_from_dataset(Blocks())
  for _ in range(2):
    out_s = st(q)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps):
    out_s = st(q)
  torch.cuda.synchronize()
  ts = (time.perf_counter() - t0) / reps
  # the same blocks as a list (a dataset object is re-iterated on every call -- 191 slices here;
  # a list is recognised by identity and version counters)
  st_list = ftk.Streaming(k=k).index_from_dataset([corpus[lo:lo + bs] for lo in range(0, n, bs)])
  for _ in range(2):
    out_l = st_list(q)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps):
    out_l = st_list(q)
  torch.cuda.synchronize()
  tl = (time.perf_counter() - t0) / reps
  del st_list
  same = bool(torch.equal(out_b[0], out_s[0]) and torch.equal(out_b[1].to(torch.int64), out_s[1].to(torch.int64))
              and torch.equal(out_b[0], out_l[0]))
  print(json.dumps({"rows": n, "dim": d, "batch": nq, "k": k,
                    "bruteforce_ms": tb * 1e3, "bruteforce_qps": nq / tb,
                    "bruteforce_pflops": 2.0 * nq * n * d / tb / 1e15,
                    "streaming_block": bs, "streaming_ms": ts * 1e3, "streaming_qps": nq / ts,
                    "streaming_list_dataset_ms": tl * 1e3, "streaming_vs_bruteforce": ts / tb,
                    "streaming_candidate_GBps": n * d * 4 / ts / 1e9,
                    "streaming_equals_bruteforce": same}), flush=True)
