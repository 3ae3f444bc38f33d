"""Fuzz: Streaming (block by block, cache off; and cached) and query_with_exclusions against the
all-f32 BruteForce scan on random block sizes / shapes; exact equality required."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd.layers import factorized_top_k as ftk

def main(seed: int = 0, cases: int = 30) -> int:
  rng = np.random.default_rng(seed)
  dev = torch.device("cuda", 0)
  bad = 0
  for case in range(cases):
    d = int(rng.choice([4, 16, 32, 64, 100, 128]))
    k = int(rng.choice([1, 5, 10, 100, 300]))
    nq = int(rng.choice([1, 33, 300, 512, 700, 1024, 3000]))
    nblocks = int(rng.integers(1, 12))
    sizes = [int(rng.choice([1, 17, 128, 1000, 4096, 65536, 70001])) for _ in range(nblocks)]
    if sum(sizes) < k:
      sizes.append(k + 64)
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
    blocks = [torch.randn((s, d), generator=g, device=dev) / d ** 0.5 for s in sizes]
    q = torch.randn((nq, d), generator=g, device=dev) / d ** 0.5
    full = torch.cat(blocks)
    n = full.shape[0]
    with_ids = bool(rng.integers(0, 2))
    ids = torch.randperm(n, generator=g, device=dev) + 1000 if with_ids else None
    os.environ["TFRS_TOPK_FILTER"] = "f32"
    ref = ftk.BruteForce(k=k).index(full, ids)
    rs, ri = ref(q)
    os.environ.pop("TFRS_TOPK_FILTER")
    ok = True
    for cache in (False, True):
      st = ftk.Streaming(k=k, cache_packed_blocks=cache)
      if with_ids:
        off = np.cumsum([0] + sizes)
        st.index_from_dataset([(ids[off[j]:off[j + 1]], b) for j, b in enumerate(blocks)])
      else:
        st.index_from_dataset(blocks)
      s, i = st(q)
      ok &= bool(torch.equal(s, rs) and torch.equal(i.to(torch.int64), ri.to(torch.int64)))
    # exclusions: every query excludes its own top few
    e = min(3, k)
    excl = ri[:, :e].contiguous()
    bf = ftk.BruteForce(k=k).index(full, ids)
    if n >= k + e:
      s2, i2 = bf.query_with_exclusions(q, excl, k=min(k, n - e))
      kk = min(k, n - e)
      os.environ["TFRS_TOPK_FILTER"] = "f32"
      rs2, ri2 = ref.query_with_exclusions(q, excl, k=kk)
      os.environ.pop("TFRS_TOPK_FILTER")
      ok &= bool(torch.equal(s2, rs2) and torch.equal(i2.to(torch.int64), ri2.to(torch.int64)))
      ok &= not bool((i2[:, :, None] == excl[:, None, :]).any())
    bad += not ok
    print(json.dumps({"case": case, "d": d, "k": k, "nq": nq, "sizes": sizes, "ids": with_ids, "equal": ok}), flush=True)
  print("MISMATCHES", bad)
  return bad



if __name__ == "__main__":
  sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 30) else 0)
