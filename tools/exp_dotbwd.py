"""DotInteraction backward (D = 32): the split-fp16 kernel on the packed gradient (default) against the round-2
f32 producer / consumer kernel (TFRS_DOT_BWD=pc), alternating in one process so that clocks and the box are
the same.  (A 12-wave variant -- eight product waves splitting the k steps, partial sums through LDS -- was
measured here at 1.11-1.13 against 1.15-1.22 ms: within the run-to-run spread, not kept.)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd import _lib
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
lib = _lib.load()
def timeit(fn, warmup=3, iters=16):
  for _ in range(warmup): fn()
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
  for a, b in ev:
    a.record(); fn(); b.record()
  torch.cuda.synchronize()
  ts = sorted(a.elapsed_time(b) for a, b in ev)
  return ts[len(ts) // 2]
B, D = 131072, 32
st = _lib.current_stream()
for F, self_i in ((101, 0), (101, 1), (64, 0)):
  x = torch.randn((B, F, D), generator=g, device=dev)
  od = F * (F + 1) // 2 if self_i else F * (F - 1) // 2
  dout = torch.randn((B, od), generator=g, device=dev)
  dx = torch.empty_like(x)
  for rnd in range(2):
    for kern in ("h16", "pc"):
      _lib.set_option("TFRS_DOT_BWD", "pc" if kern == "pc" else None)
      t = timeit(lambda: _lib.check(lib.tfrs_dot_interaction_bwd(_lib.ptr(x), _lib.ptr(dout), B, F, D, self_i, 0, _lib.ptr(dx), st)))
      byts = (2 * B * F * D + B * od) * 4
      print(json.dumps({"f": F, "self": self_i, "kernel": kern, "round": rnd, "ms": round(t, 4), "tbps": round(byts / t / 1e9, 3)}), flush=True)
  del x, dout, dx
