"""Fuzz: gather, dense scatter-add backward and the fused sparse Adagrad (row-scan and radix-sort
paths, duplicate and out-of-range ids) on random shapes against float64 references."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd.layers import embedding as emb

def main(seed: int = 0, cases: int = 30, light: bool = False) -> int:
  rng = np.random.default_rng(seed)
  bad = 0
  for case in range(cases):
    vocab = int(rng.choice([1, 7, 2000, 65537] if light else [1, 7, 2000, 65537, 1_000_000, 20_000_000]))
    d = int(rng.choice([1, 3, 4, 32, 64, 100, 128, 256]))
    n = int(rng.choice([1, 5, 4096, 100_000] if light else [1, 5, 4096, 100_000, 1_500_000]))
    if vocab * d > (6.4e7 if light else 1.5e9):
      d = 32
    idt = torch.int64 if rng.integers(0, 2) else torch.int32
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    hot = max(1, min(vocab, int(rng.choice([1, 50, vocab]))))        # few hot rows -> many duplicates
    ids = torch.randint(0, hot, (n,), generator=g, device="cuda").to(idt)
    nbad = int(rng.choice([0, 0, 3]))
    if nbad and n > 10:
      ids[:nbad] = torch.tensor([-1, vocab, vocab + 5][:nbad], device="cuda", dtype=idt)
    table = torch.randn((vocab, d), generator=g, device="cuda")
    grad = torch.randn((n, d), generator=g, device="cuda")
    valid = (ids >= 0) & (ids < vocab)
    vid = ids[valid].long()
    # gather: out-of-range ids read zeros
    out = emb.gather_rows(table, ids)
    ref = torch.zeros((n, d), device="cuda")
    ref[valid] = table[vid]
    ok = bool(torch.equal(out, ref))
    # dense scatter-add backward
    gsum64 = torch.zeros((vocab, d), dtype=torch.float64, device="cuda").index_add_(0, vid, grad[valid].double())
    asum = torch.zeros((vocab, d), dtype=torch.float64, device="cuda").index_add_(0, vid, grad[valid].double().abs())
    dense = emb.scatter_add_rows(grad, ids, vocab)
    ok &= bool(((dense.double() - gsum64).abs() <= 4e-6 * asum + 1e-30).all())
    # fused sparse Adagrad
    acc = torch.full_like(table, 0.1)
    t2 = table.clone()
    emb.adagrad_sparse_update_(t2, acc, grad, ids, 0.05, 1e-7)
    acc64 = 0.1 + gsum64 * gsum64
    touched = torch.zeros((vocab,), dtype=torch.bool, device="cuda")
    touched[vid] = True
    t64 = table.double() - 0.05 * gsum64 / torch.sqrt(acc64 + 1e-7)
    t64[~touched] = table.double()[~touched]
    acc64[~touched] = 0.1
    scale = 1.0 + asum.max()
    ok &= bool(((t2.double() - t64).abs().max() <= 2e-5 * float(scale)))
    ok &= bool(((acc.double() - acc64).abs().max() <= 2e-5 * float(scale * scale)))
    bad += not ok
    print(json.dumps({"case": case, "vocab": vocab, "d": d, "n": n, "hot": hot, "bad_ids": nbad,
                      "int64": idt == torch.int64, "ok": ok}), flush=True)
    del table, grad, dense, gsum64, asum, acc, t2, acc64, t64
  print("MISMATCHES", bad)
  return bad



if __name__ == "__main__":
  sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 30) else 0)
