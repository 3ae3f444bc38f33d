"""Kernel-trace / PMC target for the north_star gather claim: tfrs::gather_kernel at BASELINE configs[3]
shapes (65536 x 26 lookups of dim 128 out of 26 x 1M-row tables = 13.3 GB) and at configs[4]'s row
size (dim 32), called through the C ABI into a PRE-ALLOCATED output (no allocation inside the timed
or profiled region).  Prints HIP-event timings; under rocprofv3 the kernel rows / FETCH_SIZE /
WRITE_SIZE of the same launches are the counter evidence (tools/run_gather_evidence.sh)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
g = torch.Generator(device=dev).manual_seed(3)
for rows, d, n, what in ((26_000_000, 128, 65536 * 26, "configs[3]: 65536 x 26 rows of dim 128, 26 x 1M-row tables"),
                         (100_000_000, 32, 131072 * 13, "configs[4] shard: 131072 x 13 rows of dim 32, 100M-row store")):
  table = torch.empty((rows, d), dtype=torch.float32, device=dev).uniform_(-0.05, 0.05)
  ids = torch.randint(0, rows, (n,), generator=g, device=dev)
  out = torch.empty((n, d), dtype=torch.float32, device=dev)
  stream = _lib.current_stream()

  def call():
    _lib.check(lib.tfrs_embedding_gather_fwd(_lib.ptr(table), rows, d, _lib.ptr(ids), 1, n, _lib.ptr(out), None, stream))

  for _ in range(5):
    call()
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
  for a, b in ev:
    a.record(); call(); b.record()
  torch.cuda.synchronize()
  ms = sorted(a.elapsed_time(b) for a, b in ev)
  nbytes = n * (2 * d * 4 + 8)          # rows * (D*4 read + D*4 write) + ids
  med = ms[len(ms) // 2]
  print(json.dumps({"kernel": "tfrs::gather_kernel", "workload": what, "rows": n, "dim": d,
                    "algorithmic_bytes": nbytes, "ms_median": med, "ms_p10": ms[len(ms) // 10], "ms_p90": ms[9 * len(ms) // 10],
                    "GBps": nbytes / (med * 1e-3) / 1e9, "frac_of_8TBps": nbytes / (med * 1e-3) / 8e12}), flush=True)
  del table, ids, out
  torch.cuda.empty_cache()
