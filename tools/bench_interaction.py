"""Cross (BASELINE configs[3]: B = 65536, d = 3456) and DotInteraction (configs[4]: B = 131072,
F = 101, D = 32) forward / backward timings through the C ABI, with the kernel-variant switches
(TFRS_DOT_FWD, TFRS_DOT_BWD) swept in one process.  JSON lines."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd import _lib
import recommenders_amd as tfrs
from recommenders_amd.layers.feature_interaction import dcn

HBM_PEAK, F16_PEAK = 8.0e12, 2500e12
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
lib = _lib.load()
small = len(sys.argv) > 1 and sys.argv[1] == "small"


def timeit(fn, warmup=2, iters=10):
  for _ in range(warmup):
    fn()
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
  for a, b in ev:
    a.record(); fn(); b.record()
  torch.cuda.synchronize()
  ts = sorted(a.elapsed_time(b) for a, b in ev)
  return ts[len(ts) // 2] * 1e-3


def emit(**kw):
  print(json.dumps(kw), flush=True)

# ---- DotInteraction ----
B, F, D = (131072, 101, 32) if not small else (8192, 27, 16)
x = torch.randn((B, F, D), generator=g, device=dev)
od = F * (F - 1) // 2
out = torch.empty((B, od), device=dev)
dout = torch.randn((B, od), generator=g, device=dev)
dx = torch.empty_like(x)
st = _lib.current_stream()
for kern in ("pc", "direct", "staged", "f32"):   # "pc" = the default (loads and stores on different waves)
  if kern == "pc":
    os.environ.pop("TFRS_DOT_FWD", None)
  else:
    os.environ["TFRS_DOT_FWD"] = kern
  t = timeit(lambda: _lib.check(lib.tfrs_dot_interaction_fwd(_lib.ptr(x), B, F, D, 0, 0, _lib.ptr(out), st)))
  byts = (B * F * D + B * od) * 4
  emit(op="dot_interaction_fwd", kernel=kern, ms=t * 1e3, gbps=byts / t / 1e9, frac_hbm_peak=byts / t / HBM_PEAK,
       algorithmic_bytes=byts)
os.environ.pop("TFRS_DOT_FWD")
for mode in ("h16", "pc", "dense", "gather"):
  os.environ["TFRS_DOT_BWD"] = mode   # "h16" = the default (split-fp16 MFMA on the packed gradient)
  t = timeit(lambda: _lib.check(lib.tfrs_dot_interaction_bwd(_lib.ptr(x), _lib.ptr(dout), B, F, D, 0, 0, _lib.ptr(dx), st)),
             iters=6)
  byts = (2 * B * F * D + B * od) * 4
  emit(op="dot_interaction_bwd", kernel=mode, ms=t * 1e3, gbps=byts / t / 1e9, frac_hbm_peak=byts / t / HBM_PEAK,
       algorithmic_bytes=byts)
os.environ.pop("TFRS_DOT_BWD")
del x, out, dout, dx
if len(sys.argv) > 1 and sys.argv[1] == "dot":
  sys.exit(0)

# ---- Cross ----
Bc, dc = (65536, 3456) if not small else (8192, 1024)
x0 = torch.randn((Bc, dc), generator=g, device=dev)
xi = torch.randn((Bc, dc), generator=g, device=dev)
layer = tfrs.layers.feature_interaction.Cross()
with torch.no_grad():
  t_f = timeit(lambda: layer(x0, xi), iters=5)
fl = 2.0 * Bc * dc * dc
emit(op="cross_fwd", batch=Bc, dim=dc, ms=t_f * 1e3, tflops=fl / t_f / 1e12, frac_f16_mfma_peak=3 * fl / t_f / F16_PEAK)
dy = torch.randn((Bc, dc), generator=g, device=dev)
dx0, dxx = torch.empty_like(x0), torch.empty_like(x0)
dk, db = torch.empty_like(layer.kernel), torch.empty_like(layer.bias)
ws = dcn._gemm_workspace(lib.tfrs_cross_bwd_workspace_bytes(Bc, dc, 1), dev)
t_b = timeit(lambda: _lib.check(lib.tfrs_cross_bwd_f16(
    _lib.ptr(x0), _lib.ptr(xi), _lib.ptr(layer.kernel), _lib.ptr(layer.bias), 0.0, _lib.ptr(dy), Bc, dc,
    _lib.ptr(dx0), _lib.ptr(dxx), _lib.ptr(dk), _lib.ptr(db), _lib.ptr(ws), ws.numel(), st)), iters=5)
emit(op="cross_bwd (tfrs_cross_bwd_f16: 3 fused GEMMs)", batch=Bc, dim=dc, ms=t_b * 1e3, tflops=3 * fl / t_b / 1e12,
     frac_f16_mfma_peak=9 * fl / t_b / F16_PEAK)
emit(op="cross_fwd+bwd (inference forward + recomputing backward)", batch=Bc, dim=dc, ms=(t_f + t_b) * 1e3,
     tflops=4 * fl / (t_f + t_b) / 1e12)
# the training pair: forward also stores u = x W + b + diag x, backward = elementwise dx0 + 2 fused GEMMs
y, u = torch.empty_like(x0), torch.empty_like(x0)
wsf = dcn._gemm_workspace(max(lib.tfrs_gemm_f16_workspace_bytes(Bc, dc, dc),
                              lib.tfrs_cross_bwd_workspace_bytes(Bc, dc, 1)), dev)
t_ft = timeit(lambda: _lib.check(lib.tfrs_cross_fwd_f16_train(
    _lib.ptr(x0), _lib.ptr(xi), _lib.ptr(layer.kernel), _lib.ptr(layer.bias), 0.0, Bc, dc, _lib.ptr(y),
    _lib.ptr(u), _lib.ptr(wsf), wsf.numel(), st)), iters=5)
t_bs = timeit(lambda: _lib.check(lib.tfrs_cross_bwd_f16_saved(
    _lib.ptr(x0), _lib.ptr(xi), _lib.ptr(u), _lib.ptr(layer.kernel), 0.0, _lib.ptr(dy), Bc, dc,
    _lib.ptr(dx0), _lib.ptr(dxx), _lib.ptr(dk), _lib.ptr(db), _lib.ptr(wsf), wsf.numel(), st)), iters=5)
emit(op="cross_fwd_train (stores u)", batch=Bc, dim=dc, ms=t_ft * 1e3, tflops=fl / t_ft / 1e12)
emit(op="cross_bwd_saved (tfrs_cross_bwd_f16_saved: dx0 = dy * u + 2 fused GEMMs)", batch=Bc, dim=dc,
     ms=t_bs * 1e3, tflops=2 * fl / t_bs / 1e12, frac_f16_mfma_peak=6 * fl / t_bs / F16_PEAK)
emit(op="cross_fwd+bwd (training pair)", batch=Bc, dim=dc, ms=(t_ft + t_bs) * 1e3,
     tflops=3 * fl / (t_ft + t_bs) / 1e12)
