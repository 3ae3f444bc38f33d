"""DotInteraction forward (configs[4]): the producer / consumer kernel (default) against the direct kernel
(TFRS_DOT_FWD=direct), alternating in one process so that clocks and the box are the same."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd import _lib
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
lib = _lib.load()
def timeit(fn, warmup=3, iters=20):
  for _ in range(warmup): fn()
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
  for a, b in ev:
    a.record(); fn(); b.record()
  torch.cuda.synchronize()
  ts = sorted(a.elapsed_time(b) for a, b in ev)
  return ts[len(ts) // 2]
B, D = 131072, 32
st = _lib.current_stream()
for F, self_i in ((101, 0), (101, 1), (64, 0), (27, 0)):
  x = torch.randn((B, F, D), generator=g, device=dev)
  od = F * (F + 1) // 2 if self_i else F * (F - 1) // 2
  out = torch.empty((B, od), device=dev)
  for rnd in range(2):
    for kern in ("pc", "direct"):
      _lib.set_option("TFRS_DOT_FWD", None if kern == "pc" else "direct")
      t = timeit(lambda: _lib.check(lib.tfrs_dot_interaction_fwd(_lib.ptr(x), B, F, D, self_i, 0, _lib.ptr(out), st)))
      byts = (B * F * D + B * od) * 4
      print(json.dumps({"f": F, "self": self_i, "kernel": kern, "round": rnd, "ms": round(t, 4), "tbps": round(byts / t / 1e9, 3)}), flush=True)
  del x, out
