"""Timing experiment: fp16 prefilter scan with and without survivors (TFRS_DEBUG_NO_SURVIVORS)."""
import ctypes, os, sys, time
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
for env in [{}, {"TFRS_TOPK_WGS": "512"},
            {"TFRS_TOPK_RHO": "4"}, {"TFRS_TOPK_RHO": "2"}, {"TFRS_TOPK_RHO": "16"}, {"TFRS_TOPK_PREFIX": "16384"}]:
  for k in ("TFRS_TOPK_WGS", "TFRS_TOPK_RHO", "TFRS_TOPK_PREFIX"):
    os.environ.pop(k, None)
  os.environ.update(env)
  for _ in range(2):
    index(queries)
  torch.cuda.synchronize()
  lib.tfrs_profile_enable(1)
  t0 = time.perf_counter()
  steps = 10
  for _ in range(steps):
    index(queries)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / steps
  ms, n, fl = ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
  lib.tfrs_profile_read_kind(1, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl))
  lib.tfrs_profile_read(None, None, None)
  lib.tfrs_profile_enable(0)
  print(env, f"step={dt*1e3:.3f} ms scan16={ms.value/steps:.3f} ms launches/step={n.value//steps} "
        f"scan16 TFLOP/s={fl.value/ms.value/1e9:.0f}", flush=True)
