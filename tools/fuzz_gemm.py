"""Fuzz: Dense forward / backward and Cross forward / backward (auto-routed f32-MFMA, split-K and
split-fp16 GEMMs) on random shapes against float64; errors are measured against the sum of |terms|."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd.layers.feature_interaction import dcn

def main(seed: int = 0, cases: int = 30) -> int:
  rng = np.random.default_rng(seed)
  bad = 0
  TOL = 6e-6
  def rel(a, ref, scale):
    return float((a.double() - ref).abs().max()) / max(float(scale), 1e-30)
  for case in range(cases):
    m = int(rng.choice([1, 3, 100, 257, 4096, 20000, 65536, 70001]))
    k = int(rng.choice([1, 7, 13, 64, 129, 130, 500, 1002, 1024, 3000, 5082]))   # (k % 4 == 2: the 8-byte aligned row images)
    n = int(rng.choice([1, 5, 32, 127, 256, 1000, 2049]))
    if m * k * n > 3e11:
      m = 4096
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn((m, k), generator=g, device="cuda")
    w = torch.randn((k, n), generator=g, device="cuda") / k ** 0.5
    b = torch.randn((n,), generator=g, device="cuda")
    dy = torch.randn((m, n), generator=g, device="cuda")
    y = dcn.dense(x, w, b)
    dx, dw, db = dcn.dense_backward(x, w, dy, True, True, True)
    x64, w64, dy64 = x.double(), w.double(), dy.double()
    errs = {
        "y": rel(y, x64 @ w64 + b.double(), (x64.abs() @ w64.abs()).max() + b.abs().max()),
        "dx": rel(dx, dy64 @ w64.t(), (dy64.abs() @ w64.abs().t()).max()),
        "dw": rel(dw, x64.t() @ dy64, (x64.abs().t() @ dy64.abs()).max()),
        "db": rel(db, dy64.sum(0), dy64.abs().sum(0).max()),
    }
    # Cross on a square shape derived from the same draw
    d = int(rng.choice([8, 100, 257, 1024, 1500]))
    mb = int(rng.choice([5, 300, 8192, 40000]))
    x0 = torch.randn((mb, d), generator=g, device="cuda").requires_grad_(True)
    xi = torch.randn((mb, d), generator=g, device="cuda").requires_grad_(True)
    layer = dcn.Cross(diag_scale=0.3)
    yc = layer(x0, xi)
    gy = torch.randn((mb, d), generator=g, device="cuda")
    yc.backward(gy)
    W, bb = layer.kernel.detach().double(), layer.bias.detach().double()
    x0d, xid, gyd = x0.detach().double(), xi.detach().double(), gy.double()
    u = xid @ W + bb + 0.3 * xid
    su = (xid.abs() @ W.abs()).max() + 1.0
    errs["cross_y"] = rel(yc.detach(), x0d * u + xid, su * x0d.abs().max())
    dz = gyd * x0d
    errs["cross_dx0"] = rel(x0.grad, gyd * u, su * gyd.abs().max())
    errs["cross_dx"] = rel(xi.grad, dz @ W.t() + gyd + 0.3 * dz, (dz.abs() @ W.abs().t()).max() + gyd.abs().max())
    errs["cross_dw"] = rel(layer.kernel.grad, xid.t() @ dz, (xid.abs().t() @ dz.abs()).max())
    ok = all(v <= TOL for v in errs.values())
    bad += not ok
    print(json.dumps({"case": case, "m": m, "k": k, "n": n, "cross": [mb, d], "ok": ok,
                      "worst": max(errs, key=errs.get), "err": max(errs.values())}), flush=True)
  print("MISMATCHES", bad)
  return bad



if __name__ == "__main__":
  sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 30) else 0)
