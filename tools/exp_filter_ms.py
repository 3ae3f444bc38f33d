"""Timing of the fp16 filter pass on the BASELINE configs[1] batch (per-launch HIP-event times from
tfrs_profile_*), repeated; `python tools/exp_filter_ms.py NAME=VALUE ...` runs one more round per library
switch given (TFRS_* options).  Used by tools/ab_scan16.sh for same-box A/B runs of two builds."""
import ctypes, os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd import _lib
from recommenders_amd.layers import factorized_top_k as ftk

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(42)
corpus = torch.randn((1_000_000, 64), generator=g, device=dev) / 8.0
queries = torch.randn((8192, 64), generator=g, device=dev) / 8.0
index = ftk.BruteForce(k=100).index(corpus)
lib = _lib.load()
ref = None
vals = [None] + sys.argv[1:] + [None]
for v in vals:
  if v:
    for kv in v.split(","):          # several switches at once: A=1,B=2
      _lib.set_option(kv.split("=")[0], kv.split("=")[1])
  for _ in range(3):
    out = index(queries)
  torch.cuda.synchronize()
  if ref is None:
    ref = (out[0].clone(), out[1].clone())
  same = bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]))
  lib.tfrs_profile_enable(1)
  t0 = time.perf_counter()
  steps = 20
  for _ in range(steps):
    index(queries)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / steps
  res = {}
  for kind in (1, 2):
    ms, n, fl = ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
    lib.tfrs_profile_read_kind(kind, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl))
    res[kind] = (ms.value / steps, fl.value / max(ms.value, 1e-9) / 1e9)
  lib.tfrs_profile_read(None, None, None)
  lib.tfrs_profile_enable(0)
  if v:
    for kv in v.split(","):
      _lib.set_option(kv.split("=")[0], None)
  print(json.dumps({"switch": v, "step_ms": round(dt * 1e3, 4), "filter_ms": round(res[1][0], 4),
                    "filter_tflops": round(res[1][1], 1), "thr_ms": round(res[2][0], 4), "same": same}), flush=True)
