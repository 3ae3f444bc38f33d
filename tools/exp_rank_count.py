"""Decomposes the fused metric update at the quickstart shapes (4096 queries x 1682 candidates x 64):
rank_count alone, rank_count + folded hits update, the separate hits kernel; grid sizes via
TFRS_RANK_WGS (development tool)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
nq, d, vocab, nc = 4096, 64, 2000, 1682
q = torch.randn((nq, d), generator=g, device=dev) * 0.05
c = torch.randn((nq, d), generator=g, device=dev) * 0.05
table = torch.randn((vocab, d), generator=g, device=dev) * 0.05
ids = torch.arange(nc, device=dev)
counts = torch.zeros((nq + 1,), dtype=torch.int32, device=dev)
state = torch.zeros((10,), device=dev); results = torch.zeros((5,), device=dev)
ks = (ctypes.c_int32 * 5)(1, 5, 10, 50, 100)
def acc():
  _lib.check(lib.tfrs_rank_count_accumulate(_lib.ptr(q), _lib.ptr(c), nq, d, _lib.ptr(table), _lib.ptr(ids), 1, nc, vocab, _lib.ptr(counts), 1, _lib.current_stream()))
def hits():
  _lib.check(lib.tfrs_topk_hits_update(_lib.ptr(counts), nq, ks, 5, None, _lib.ptr(state), _lib.ptr(results), None, _lib.current_stream()))
def timeit(fn, n=200):
  for _ in range(10): fn()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  graph = torch.cuda.CUDAGraph()
  side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(side):
    fn()
  torch.cuda.current_stream().wait_stream(side)
  if True:
    with torch.cuda.graph(graph):
      for _ in range(20): fn()
  graph.replay(); torch.cuda.synchronize()
  a.record()
  for _ in range(n // 20): graph.replay()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b) / (n // 20 * 20) * 1e3

print(f"accumulate {timeit(acc):.1f} us, hits kernel {timeit(hits):.1f} us, both {timeit(lambda: (acc(), hits())):.1f} us")
