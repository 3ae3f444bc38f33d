"""Fuzz: the default fp16-prefiltered search (statistical threshold plan on shuffled indexes) against
the all-f32 scan on random shapes and awkward data (duplicates, constant columns, scaled rows,
negative-only scores); exact equality of scores and identifiers is required."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd.layers import factorized_top_k as ftk

def main(seed: int = 0, cases: int = 30) -> int:
  rng = np.random.default_rng(seed)
  dev = torch.device("cuda", 0)
  bad = 0
  for case in range(cases):
    n = int(rng.integers(65536, 900_000))
    d = int(rng.choice([3, 8, 16, 24, 32, 48, 64, 100, 128]))
    k = int(rng.choice([1, 2, 10, 50, 100, 200, 400, 512]))
    nq = int(rng.choice([1, 7, 64, 513, 2048, 5000]))
    kind = str(rng.choice(["gauss", "dups", "const_col", "row_scales", "negative", "clustered"]))
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
    c = torch.randn((n, d), generator=g, device=dev) / d ** 0.5
    q = torch.randn((nq, d), generator=g, device=dev) / d ** 0.5
    if kind == "dups":
      c[n // 2:] = c[: n - n // 2].clone()               # every row twice
    elif kind == "const_col":
      c[:, 0] = 3.0
    elif kind == "row_scales":
      c *= torch.exp(3.0 * torch.randn((n, 1), generator=g, device=dev))
    elif kind == "negative":
      c = -c.abs(); q = q.abs()
    elif kind == "clustered":
      cen = torch.randn((64, d), generator=g, device=dev) / d ** 0.5
      c = cen[torch.arange(n, device=dev) * 64 // n] + 0.3 * c
      q = cen[torch.randint(0, 64, (nq,), generator=g, device=dev)] + 0.3 * q
    layer = ftk.BruteForce(k=k).index(c)
    os.environ["TFRS_TOPK_FILTER"] = "f32"
    s32, i32 = layer(q)
    os.environ.pop("TFRS_TOPK_FILTER")
    s, i = layer(q)
    ok = bool(torch.equal(s, s32) and torch.equal(i, i32))
    bad += not ok
    print(json.dumps({"case": case, "n": n, "d": d, "k": k, "nq": nq, "kind": kind, "equal": ok,
                      "redo": layer.last_redo_count(), "reasons": layer.last_redo_reasons()}), flush=True)
  print("MISMATCHES", bad)
  return bad



if __name__ == "__main__":
  sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 30) else 0)
