"""In-batch softmax chain at the quickstart shapes, launched back to back through the C ABI and timed with events:
forward (prep + fwd + finalize) and backward (bwd + reduce) separately.  Env: TFRS_SOFTMAX_NW, TFRS_SOFTMAX_WGS.
python tools/exp_sm16_ms.py [B] [D] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd import _lib
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn((B, D), generator=g, device=dev) * 0.05
c = torch.randn((B, D), generator=g, device=dev) * 0.05
lib = _lib.load()
ws = torch.empty((lib.tfrs_inbatch_softmax_workspace_bytes(B, B, D),), dtype=torch.uint8, device=dev)
loss = torch.empty((), dtype=torch.float32, device=dev)
lse = torch.empty((B,), dtype=torch.float32, device=dev)
pos = torch.empty((B,), dtype=torch.float32, device=dev)
dq, dc = torch.empty_like(q), torch.empty_like(c)
one = torch.ones((), dtype=torch.float32, device=dev)
s = _lib.current_stream()


def fwd():
  _lib.check(lib.tfrs_inbatch_softmax_ce_fwd(_lib.ptr(q), _lib.ptr(c), B, B, D, None, 1.0, None, None, None, _lib.ptr(loss),
                                             _lib.ptr(lse), _lib.ptr(pos), _lib.ptr(ws), ws.numel(), s))


def bwd():
  _lib.check(lib.tfrs_inbatch_softmax_ce_bwd(_lib.ptr(q), _lib.ptr(c), B, B, D, None, 1.0, None, None, None, _lib.ptr(lse),
                                             _lib.ptr(one), _lib.ptr(dq), _lib.ptr(dc), _lib.ptr(ws), ws.numel(), 1, s))


def timed(fn):
  for _ in range(10):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e3


fwd()
print("B %d D %d NW=%s WGS=%s: fwd chain %.1f us, bwd chain %.1f us, both %.1f us; loss %.6f dq %.6e" % (
    B, D, os.environ.get("TFRS_SOFTMAX_NW", "-"), os.environ.get("TFRS_SOFTMAX_WGS", "-"),
    timed(fwd), timed(bwd), timed(lambda: (fwd(), bwd())), float(loss), float(dq.abs().sum())))
