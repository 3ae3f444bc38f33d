"""Cross at configs[3] (B = 65536, d = 3456): one inference forward, then one training pair (forward that
stores u + backward from the saved u) -- run under `rocprofv3 --kernel-trace` to list every dispatch in order
(tools/run_cross_trace.sh), so that the operand-preparation passes of each product can be priced."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd import _lib
import recommenders_amd as tfrs
from recommenders_amd.layers.feature_interaction import dcn
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
lib = _lib.load()
st = _lib.current_stream()
Bc, dc = 65536, 3456
x0 = torch.randn((Bc, dc), generator=g, device=dev)
xi = torch.randn((Bc, dc), generator=g, device=dev)
layer = tfrs.layers.feature_interaction.Cross()
with torch.no_grad():
  for _ in range(2):
    layer(x0, xi)
torch.cuda.synchronize()
dy = torch.randn((Bc, dc), generator=g, device=dev)
dx0, dxx = torch.empty_like(x0), torch.empty_like(x0)
dk, db = torch.empty_like(layer.kernel), torch.empty_like(layer.bias)
y, u = torch.empty_like(x0), torch.empty_like(x0)
wsf = dcn._gemm_workspace(max(lib.tfrs_gemm_f16_workspace_bytes(Bc, dc, dc),
                              lib.tfrs_cross_bwd_workspace_bytes(Bc, dc, 1)), dev)
for _ in range(2):
  _lib.check(lib.tfrs_cross_fwd_f16_train(
      _lib.ptr(x0), _lib.ptr(xi), _lib.ptr(layer.kernel), _lib.ptr(layer.bias), 0.0, Bc, dc, _lib.ptr(y),
      _lib.ptr(u), _lib.ptr(wsf), wsf.numel(), st))
  _lib.check(lib.tfrs_cross_bwd_f16_saved(
      _lib.ptr(x0), _lib.ptr(xi), _lib.ptr(u), _lib.ptr(layer.kernel), 0.0, _lib.ptr(dy), Bc, dc,
      _lib.ptr(dx0), _lib.ptr(dxx), _lib.ptr(dk), _lib.ptr(db), _lib.ptr(wsf), wsf.numel(), st))
torch.cuda.synchronize()
