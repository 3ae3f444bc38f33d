"""In-process sweep of the top-K round/queue tuning knobs on the bench workload
(1M x 64 corpus, batch 8192, top-100).  Development tool; prints one line per setting."""

import ctypes
import itertools
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from recommenders_amd import _lib  # noqa: E402
from recommenders_amd.layers import factorized_top_k as ftk  # noqa: E402


def main():
  dev = torch.device("cuda", 0)
  g = torch.Generator(device=dev).manual_seed(42)
  corpus = torch.randn((1_000_000, 64), generator=g, device=dev) / 8.0
  queries = torch.randn((8192, 64), generator=g, device=dev) / 8.0
  index = ftk.BruteForce(k=100).index(corpus)
  lib = _lib.load()
  ref = None
  grid = list(itertools.product([1, 0], [4096, 16384], [4, 8, 16], [1024, 2048]))
  if len(sys.argv) > 1 and sys.argv[1] == "quick":
    grid = [(1, 4096, 8, 1024), (0, 4096, 8, 1024), (1, 4096, 4, 1024), (1, 4096, 16, 1024),
            (1, 4096, 8, 512), (1, 4096, 8, 2048), (1, 2048, 8, 1024), (1, 8192, 8, 1024),
            (1, 16384, 16, 1024), (1, 4096, 32, 1024)]
  for glds, prefix, rho, wgs in grid:
    os.environ["TFRS_SCAN_GLDS"] = str(glds)
    os.environ["TFRS_TOPK_PREFIX"] = str(prefix)
    os.environ["TFRS_TOPK_RHO"] = str(rho)
    os.environ["TFRS_TOPK_WGS"] = str(wgs)
    for _ in range(2):
      out = index(queries)
    torch.cuda.synchronize()
    lib.tfrs_profile_enable(1)
    t0 = time.perf_counter()
    steps = 8
    for _ in range(steps):
      out = index(queries)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ms, n, fl = ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
    lib.tfrs_profile_read(ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl))
    lib.tfrs_profile_enable(0)
    if ref is None:
      ref = out
    same = bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]))
    print(f"glds={glds} prefix={prefix:6d} rho={rho:3d} wgs={wgs:5d}  step={dt*1e3:7.3f} ms  "
          f"{8192/dt/1e3:8.1f} kq/s  scan={ms.value/steps:7.3f} ms  "
          f"scan_tflops={fl.value/ms.value/1e9:6.1f}  launches/step={n.value//steps}  same={same}",
          flush=True)


if __name__ == "__main__":
  main()
