"""Kernel time of the wide block-fed filter (HIP events of tfrs_profile kind 2 = raw16 launches) at 512 queries over 12.5M x 128."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd import _lib
from recommenders_amd.layers import factorized_top_k as ftk
dev = torch.device("cuda", 0); lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(5)
n, d, k, bs = 12_500_000, 128, 100, 65536
corpus = torch.randn((n, d), generator=g, device=dev) / d ** 0.5
class Blocks:
  def __iter__(self):
    for lo in range(0, n, bs): yield corpus[lo:lo + bs]
st = ftk.Streaming(k=k).index_from_dataset(Blocks())
for nq in (512, 1024):
  q = torch.randn((nq, d), generator=g, device=dev) / d ** 0.5
  for _ in range(2): st(q)
  torch.cuda.synchronize()
  lib.tfrs_profile_enable(1)
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(4): st(q)
  b.record(); torch.cuda.synchronize()
  res = {}
  for kind in (0, 1, 2):
    ms, cnt, fl = ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
    lib.tfrs_profile_read_kind(kind, ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(fl))
    res[kind] = (round(ms.value / 4, 3), cnt.value // 4)
  lib.tfrs_profile_read(None, None, None); lib.tfrs_profile_enable(0)
  print(json.dumps({"nq": nq, "call_ms": round(a.elapsed_time(b) / 4, 3), "scan_ms_per_call_by_kind": res}), flush=True)
