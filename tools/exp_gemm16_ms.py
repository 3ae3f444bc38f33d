"""Times the split-fp16 products of BASELINE configs[3] / configs[4] through the C ABI: the Cross forward (65536 x 3456 @
3456 x 3456), the training pair, and the DLRM top MLP's first product (131072 x 5082 @ 5082 x 1024, relu), each with the
wide and the scalar epilogue of gemm16_big_kernel (TFRS_GEMM16_EPILOGUE) and a float64 check on sampled rows.  Run once
per library: tools/ab_variants.sh run with EXP=tools/exp_gemm16_ms.py (TFRS_ALLOW_ABLATION=1 for ablation builds)."""
import json, os, sys
os.environ.setdefault("TFRS_ALLOW_ABLATION", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd import _lib
from recommenders_amd.layers.feature_interaction import dcn
lib = _lib.load()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
b, d = 65536, 3456
x0 = torch.randn((b, d), generator=g, device=dev) * 0.5
x = torch.randn((b, d), generator=g, device=dev) * 0.5
dy = torch.randn((b, d), generator=g, device=dev)
w = torch.randn((d, d), generator=g, device=dev) * 0.05
bias = torch.randn((d,), generator=g, device=dev) * 0.1
y, u = torch.empty_like(x0), torch.empty_like(x0)
dx0, dx, dk, db = torch.empty_like(x0), torch.empty_like(x0), torch.empty_like(w), torch.empty_like(bias)
ws = dcn._gemm_workspace(max(lib.tfrs_gemm_f16_workspace_bytes(b, d, d), lib.tfrs_cross_bwd_workspace_bytes(b, d, 1)), dev)
st = _lib.current_stream()
fwd = lambda: _lib.check(lib.tfrs_cross_fwd_f16(_lib.ptr(x0), _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), 0.0, b, d, _lib.ptr(y), _lib.ptr(ws), ws.numel(), st))
def pair():
  _lib.check(lib.tfrs_cross_fwd_f16_train(_lib.ptr(x0), _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), 0.0, b, d, _lib.ptr(y), _lib.ptr(u), _lib.ptr(ws), ws.numel(), st))
  _lib.check(lib.tfrs_cross_bwd_f16_saved(_lib.ptr(x0), _lib.ptr(x), _lib.ptr(u), _lib.ptr(w), 0.0, _lib.ptr(dy), b, d, _lib.ptr(dx0), _lib.ptr(dx), _lib.ptr(dk), _lib.ptr(db), _lib.ptr(ws), ws.numel(), st))
xa = torch.randn((131072, 5082), generator=g, device=dev)
wa = torch.randn((5082, 1024), generator=g, device=dev) * 0.02
ba = torch.randn((1024,), generator=g, device=dev) * 0.1
ws2 = torch.empty((lib.tfrs_gemm_f16_workspace_bytes(131072, 1024, 5082),), dtype=torch.uint8, device=dev)
oa = torch.empty((131072, 1024), device=dev)
mlp = lambda: _lib.check(lib.tfrs_dense_fwd_act(_lib.ptr(xa), _lib.ptr(wa), _lib.ptr(ba), 131072, 5082, 1024, 1, _lib.ptr(oa), None, 1, _lib.ptr(ws2), ws2.numel(), st))

def med(fn, iters=8, warm=2):
  for _ in range(warm): fn()
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
  for a, e in ev:
    a.record(); fn(); e.record()
  torch.cuda.synchronize()
  ts = sorted(a.elapsed_time(e) for a, e in ev)
  return ts[len(ts) // 2]

rows = torch.arange(0, b, 4099, device=dev)
ref = x0[rows].double() * (x[rows].double() @ w.double() + bias.double()) + x[rows].double()
refm = torch.relu(xa[rows].double() @ wa.double() + ba.double())
out = {}
keep = {}
for mode in ("wide", "scalar"):
  _lib.set_option("TFRS_GEMM16_EPILOGUE", "wide" if mode == "wide" else None)
  out[mode] = {"cross_fwd_ms": round(med(fwd), 3), "pair_ms": round(med(pair, 5, 1), 3), "dlrm_mlp0_ms": round(med(mlp), 3)}
  fwd(); pair(); mlp()
  out[mode]["rel_err"] = float((y[rows].double() - ref).abs().max() / ref.abs().max())
  out[mode]["mlp_rel_err"] = float((oa[rows].double() - refm).abs().max() / refm.abs().max())
  keep[mode] = [t.clone() for t in (y, u, dx0, dx, dk, db, oa)]
_lib.set_option("TFRS_GEMM16_EPILOGUE", None)
# shader clock during the forward product's GEMM kernel (cycles of s_memtime over 100 MHz ticks, summed over workgroups)
_lib.set_option("TFRS_GEMM16_CLOCKS", "1")
for _ in range(3): fwd()
torch.cuda.synchronize()
tot = lib.tfrs_gemm_f16_workspace_bytes(b, d, d)
clk = ws[tot - 256: tot - 240].view(torch.int64).cpu().tolist()
_lib.set_option("TFRS_GEMM16_CLOCKS", None)
out["gemm_kernel_mhz"] = round(clk[0] / max(clk[1], 1) * 100.0, 1)
out["gemm_kernel_ms_per_wg_sum"] = round(clk[1] / 1e5, 3)   # sum over workgroups of their lifetimes, ms
out["wide_equals_scalar"] = all(torch.equal(p, q) for p, q in zip(keep["wide"], keep["scalar"]))
out["exec_pflops_fwd"] = round(3 * 2.0 * b * d * d / (out["wide"]["cross_fwd_ms"] * 1e-3) / 1e15, 3)
print(json.dumps(out), flush=True)
