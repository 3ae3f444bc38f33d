"""rocprofv3 target: a few launches of DotInteraction fwd/bwd (configs[4]) and Cross fwd/bwd (configs[3])."""
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
B, F, D = 131072, 101, 32
x = torch.randn((B, F, D), generator=g, device=dev)
od = F * (F - 1) // 2
out = torch.empty((B, od), device=dev)
dout = torch.randn((B, od), generator=g, device=dev)
dx = torch.empty_like(x)
for _ in range(4):
  lib.tfrs_dot_interaction_fwd(_lib.ptr(x), B, F, D, 0, 0, _lib.ptr(out), st)
  lib.tfrs_dot_interaction_bwd(_lib.ptr(x), _lib.ptr(dout), B, F, D, 0, 0, _lib.ptr(dx), st)
torch.cuda.synchronize()
del x, out, dout, dx
Bc, dc = 65536, 3456
x0 = torch.randn((Bc, dc), generator=g, device=dev)
xi = torch.randn((Bc, dc), generator=g, device=dev)
layer = tfrs.layers.feature_interaction.Cross()
dy = torch.randn((Bc, dc), generator=g, device=dev)
x0g, xg = x0.requires_grad_(True), xi.requires_grad_(True)
for _ in range(3):
  y = layer(x0g, xg)
  y.backward(dy)
torch.cuda.synchronize()
