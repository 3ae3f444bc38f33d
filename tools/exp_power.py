"""Is the fp16 filter pass clock/power limited?  Same kernel, same shapes, same survivor
statistics, but operand data of different switching activity: dense Gaussian (the bench data),
and the same corpus with 7 of every 8 features zeroed (1/8 of the multiplier activity; scores
stay Gaussian, top-K stays well defined).  A large speed-up on the sparse corpus means the
dense run is bounded by the power its operands draw, not by issue slots."""
import ctypes, os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd import _lib
from recommenders_amd.layers import factorized_top_k as ftk

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(42)
lib = _lib.load()
queries = torch.randn((8192, 64), generator=g, device=dev) / 8.0
dense = torch.randn((1_000_000, 64), generator=g, device=dev) / 8.0
mask = (torch.arange(64, device=dev) % 8 == 0).float()
for name, corpus in (("dense gaussian", dense), ("1 of 8 features non-zero", dense * mask * (8 ** 0.5)),
                     ("zeros except feature 0", dense * (torch.arange(64, device=dev) == 0).float() * 8.0)):
  index = ftk.BruteForce(k=100).index(corpus)
  for _ in range(3):
    index(queries)
  torch.cuda.synchronize()
  lib.tfrs_profile_enable(1)
  steps = 20
  for _ in range(steps):
    index(queries)
  torch.cuda.synchronize()
  res = {}
  for kind in (1, 2):
    ms, n, fl = ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
    lib.tfrs_profile_read_kind(kind, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl))
    res[kind] = (ms.value / steps, fl.value / max(ms.value, 1e-9) / 1e9)
  lib.tfrs_profile_read(None, None, None)
  lib.tfrs_profile_enable(0)
  print(json.dumps({"corpus": name, "filter_ms": round(res[1][0], 4), "filter_tflops": round(res[1][1], 1),
                    "binmax_ms": round(res[2][0], 4), "redo": index.last_redo_count()}), flush=True)
  del index
