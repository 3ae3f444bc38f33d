"""Tile order of tfrs::gemm16_big_kernel (TFRS_GEMM16_RASTER = 0 round-robin | 1 XCD-contiguous row-major | 2 XCD-contiguous,
4 row panels x 8 column panels per 32 tiles) on the Cross forward / training pair of BASELINE configs[3] and the DLRM top
MLP product, alternating inside one process (profiles/r05_gemm_raster.txt)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd import _lib
from recommenders_amd.layers.feature_interaction import dcn
lib = _lib.load()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)

def timeit(fn, iters=6, warm=2):
  for _ in range(warm): fn()
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
  for a, b in ev:
    a.record(); fn(); b.record()
  torch.cuda.synchronize()
  ts = sorted(a.elapsed_time(b) for a, b in ev)
  return ts[len(ts) // 2]

b, d = 65536, 3456
x0 = torch.randn((b, d), generator=g, device=dev) * 0.5
x = torch.randn((b, d), generator=g, device=dev) * 0.5
dy = torch.randn((b, d), generator=g, device=dev)
w = torch.randn((d, d), generator=g, device=dev) * 0.05
bias = torch.zeros((d,), device=dev)
y, u = torch.empty_like(x0), torch.empty_like(x0)
dx0, dx, dk, db = torch.empty_like(x0), torch.empty_like(x0), torch.empty_like(w), torch.empty_like(bias)
ws = dcn._gemm_workspace(max(lib.tfrs_gemm_f16_workspace_bytes(b, d, d), lib.tfrs_cross_bwd_workspace_bytes(b, d, 1)), dev)
st = _lib.current_stream()
fwd = lambda: _lib.check(lib.tfrs_cross_fwd_f16(_lib.ptr(x0), _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), 0.0, b, d, _lib.ptr(y), _lib.ptr(ws), ws.numel(), st))
def pair():
  _lib.check(lib.tfrs_cross_fwd_f16_train(_lib.ptr(x0), _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), 0.0, b, d, _lib.ptr(y), _lib.ptr(u), _lib.ptr(ws), ws.numel(), st))
  _lib.check(lib.tfrs_cross_bwd_f16_saved(_lib.ptr(x0), _lib.ptr(x), _lib.ptr(u), _lib.ptr(w), 0.0, _lib.ptr(dy), b, d, _lib.ptr(dx0), _lib.ptr(dx), _lib.ptr(dk), _lib.ptr(db), _lib.ptr(ws), ws.numel(), st))
# the DLRM top MLP's first product: 131072 x 5082 @ 5082 x 1024
xa = torch.randn((131072, 5082), generator=g, device=dev)
wa = torch.randn((5082, 1024), generator=g, device=dev) * 0.02
ws2 = torch.empty((lib.tfrs_gemm_f16_workspace_bytes(131072, 1024, 5082),), dtype=torch.uint8, device=dev)
oa = torch.empty((131072, 1024), device=dev)
mlp = lambda: _lib.check(lib.tfrs_dense_fwd_f16(_lib.ptr(xa), _lib.ptr(wa), None, 131072, 5082, 1024, _lib.ptr(oa), _lib.ptr(ws2), ws2.numel(), st))
ref = {}
for r in ("0", "1", "2", "0", "1", "2"):
  _lib.set_option("TFRS_GEMM16_RASTER", r)
  tf, tp, tm = timeit(fwd), timeit(pair, 4, 1), timeit(mlp)
  fwd(); pair(); mlp()
  outs = [y.clone(), dx.clone(), dk.clone(), oa.clone()]
  same = all(torch.equal(a, b_) for a, b_ in zip(outs, ref.setdefault("o", outs)))
  print(json.dumps({"raster": r, "cross_fwd_ms": round(tf, 3), "cross_pair_ms": round(tp, 3), "dlrm_top_mlp_ms": round(tm, 3), "same": same}), flush=True)
_lib.set_option("TFRS_GEMM16_RASTER", None)
