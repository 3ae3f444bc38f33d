"""DotInteraction forward (configs[4]) against a plain copy of the same bytes in the same process (the
box's streaming rate for this read / write mix), row alignment of the packed output (out_stride: 8 / 16 /
128-byte aligned rows) and grid size.  (Round 4 also measured nontemporal loads / stores in the kernel:
0.870-0.876 against 0.855-0.872 ms at the default stride -- nothing; the switch was not kept.)  Evidence tool."""
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
B, D, F = 131072, 32, 101
st = _lib.current_stream()
x = torch.randn((B, F, D), generator=g, device=dev)
od = F * (F - 1) // 2
byts = (B * F * D + B * od) * 4
# calibration: a plain copy moving the same number of bytes (half read, half written)
a = torch.empty(byts // 8, device=dev); b = torch.empty_like(a)
t = timeit(lambda: b.copy_(a))
print(json.dumps({"what": "torch copy, same bytes", "ms": round(t, 4), "tbps": round(byts / t / 1e9, 3)}), flush=True)
del a, b
for rnd in range(2):
  for stride in (od, od + 2, od + 6):
    out = torch.empty((B, stride), device=dev)
    for nt in (None, "1"):
      for grid in (None, "256"):
        _lib.set_option("TFRS_DOT_FWD_GRID", grid)
        t = timeit(lambda: _lib.check(lib.tfrs_dot_interaction_fwd_strided(_lib.ptr(x), B, F, D, 0, _lib.ptr(out), stride, st)))
        print(json.dumps({"stride": stride, "grid": grid or "512", "round": rnd, "ms": round(t, 4),
                          "tbps": round(byts / t / 1e9, 3)}), flush=True)
    del out
_lib.set_option("TFRS_DOT_FWD_GRID", None)
