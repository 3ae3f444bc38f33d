"""fp16 filter pass per launch for the shapes of scan16f_kernel at dims 32 / 64 / 128 (2M rows, 8192 and 2048 queries),
alternating inside one process: which instantiation the launcher should pick per dim (profiles/r05_scan16f_shapes.txt)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd import _lib
from recommenders_amd.layers import factorized_top_k as ftk
dev = torch.device("cuda", 0)
lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(1)
for d in (128, 32, 64):
  corpus = torch.randn((2_000_000, d), generator=g, device=dev) / d ** 0.5
  index = ftk.BruteForce(k=100).index(corpus)
  for nq in (8192, 2048):
    q = torch.randn((nq, d), generator=g, device=dev) / d ** 0.5
    ref = None
    for shape in ("8x2", "16x2", None, "8x2", "16x2", None):
      _lib.set_option("TFRS_SCAN16_SHAPE", shape)
      for _ in range(3):
        out = index(q)
      torch.cuda.synchronize()
      if ref is None:
        ref = (out[0].clone(), out[1].clone())
      same = bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]))
      lib.tfrs_profile_enable(1)
      for _ in range(10):
        index(q)
      torch.cuda.synchronize()
      ms, n, fl = ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
      lib.tfrs_profile_read_kind(1, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl))
      lib.tfrs_profile_read(None, None, None)
      lib.tfrs_profile_enable(0)
      _lib.set_option("TFRS_SCAN16_SHAPE", None)
      print(json.dumps({"dim": d, "nq": nq, "shape": shape or "default", "filter_ms": round(ms.value / max(n.value, 1), 4),
                        "tflops": round(fl.value / max(ms.value, 1e-9) / 1e9, 1), "same": same}), flush=True)
  del index, corpus
  torch.cuda.empty_cache()
