import json, os, sys
sys.path.insert(0, os.getcwd())
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
B, F, D = 131072, 101, 32
x = torch.randn((B, F, D), generator=g, device=dev)
od = F * (F - 1) // 2
dout = torch.randn((B, od), generator=g, device=dev)
dx = torch.empty_like(x)
st = _lib.current_stream()
for dbg in (0, 1, 2, 4, 6, 8, 9, 7, 15, 14):
  _lib.set_option("TFRS_DOT_BWD_DBG", str(dbg))
  t = timeit(lambda: _lib.check(lib.tfrs_dot_interaction_bwd(_lib.ptr(x), _lib.ptr(dout), B, F, D, 0, 0, _lib.ptr(dx), st)))
  print(json.dumps({"dbg": dbg, "ms": t}), flush=True)
