"""Cycle-counter trace of sm16_bwd_kernel (ablation build with TFRS_SM16_ABLATE & 256): per traced workgroup the
prologue, per-step and epilogue cycles of wave 0.  python tools/exp_sm16_trace.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TFRS_SOFTMAX_BWD_V", "1")
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
import numpy as np, torch
from recommenders_amd import _lib
dev = torch.device("cuda", 0)
B, D = 4096, 64
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn((B, D), generator=g, device=dev) * 0.05
c = torch.randn((B, D), generator=g, device=dev) * 0.05
lib = _lib.load()
ws = torch.empty((lib.tfrs_inbatch_softmax_workspace_bytes(B, B, D),), dtype=torch.uint8, device=dev)
loss = torch.empty((), dtype=torch.float32, device=dev); lse = torch.empty((B,), dtype=torch.float32, device=dev)
pos = torch.empty((B,), dtype=torch.float32, device=dev); dq, dc = torch.empty_like(q), torch.empty_like(c)
one = torch.ones((), dtype=torch.float32, device=dev); s = _lib.current_stream()
_lib.check(lib.tfrs_inbatch_softmax_ce_fwd(_lib.ptr(q), _lib.ptr(c), B, B, D, None, 1.0, None, None, None, _lib.ptr(loss),
                                           _lib.ptr(lse), _lib.ptr(pos), _lib.ptr(ws), ws.numel(), s))
for _ in range(5):
  _lib.check(lib.tfrs_inbatch_softmax_ce_bwd(_lib.ptr(q), _lib.ptr(c), B, B, D, None, 1.0, None, None, None, _lib.ptr(lse),
                                             _lib.ptr(one), _lib.ptr(dq), _lib.ptr(dc), _lib.ptr(ws), ws.numel(), 1, s))
torch.cuda.synchronize()
out = np.zeros((64, 32), dtype=np.uint64)
raw = ctypes.CDLL(_lib.__dict__.get("_PATH", None) or os.path.join(os.path.dirname(_lib.__file__), "libtfrs_hip.so"))
raw.tfrs_debug_sm16_trace(out.ctypes.data_as(ctypes.c_void_p))
t = out.astype(np.int64)
t0 = t[:32, 0].min()
print("blk  start   prolog  steps...                                      epilog  total   hwid")
for i in range(32):
  r = t[i]
  steps = [int(r[3 + k] - r[2 + k]) for k in range(NS - 1)] + [int(r[20] - r[1 + NS])]
  print("%3d %7d %7d  %s  %6d %7d  %08x" % (i * 16, r[0] - t0, r[1] - r[0], " ".join("%5d" % x for x in steps), r[21] - r[20],
                                          r[21] - r[0], int(r[22]) & 0xffffffff))
print("first step after prologue: ", [int(t[i, 2] - t[i, 1]) for i in range(8)])
