"""DotInteraction backward (D = 32): time and bytes/s by feature count -- packed rows whose byte length
is a multiple of 16 (F = 97: aligned 16-byte loads) against neighbours that are not -- and the debug
decomposition of the kernel when the library is built with the TFRS_DOT_BWD_DBG switch."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd import _lib
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
lib = _lib.load()
def timeit(fn, warmup=2, iters=6):
  for _ in range(warmup): fn()
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
  for a, b in ev:
    a.record(); fn(); b.record()
  torch.cuda.synchronize()
  ts = sorted(a.elapsed_time(b) for a, b in ev)
  return ts[len(ts) // 2]
B, D = 131072, 32
st = _lib.current_stream()
dbgs = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
for F, self_i in ((96, 0), (101, 0)):
  x = torch.randn((B, F, D), generator=g, device=dev)
  od = F * (F + 1) // 2 if self_i else F * (F - 1) // 2
  dout = torch.randn((B, od), generator=g, device=dev)
  dx = torch.empty_like(x)
  for dbg in dbgs:
    _lib.set_option("TFRS_DOT_BWD_DBG", str(dbg))
    t = timeit(lambda: _lib.check(lib.tfrs_dot_interaction_bwd(_lib.ptr(x), _lib.ptr(dout), B, F, D, self_i, 0, _lib.ptr(dx), st)))
    byts = (2 * B * F * D + B * od) * 4
    print(json.dumps({"f": F, "self": self_i, "row_bytes_mod16": (od * 4) % 16, "dbg": dbg, "ms": round(t, 4), "gbps": round(byts / t / 1e6, 1)}), flush=True)
  del x, dout, dx
