"""Kernel-trace / PMC target: Streaming top-100 over a 12.5M x 128 stream handed over as a dataset object
(one GPU's shard of BASELINE.json configs[2]), at ONE batch size per run (BATCH=1 | 64 | 8192):
rocprofv3 then shows the kernels of tfrs_streaming_topk_update_blocks -- rawscan_kernel + select_kernel
(small batches), pack16_raw_kernel + scan16f_kernel + list_topk16_kernel (large ones)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.layers import factorized_top_k as ftk

dev = torch.device("cuda", 0)
n, d, k, bs = int(os.environ.get("ROWS", 12_500_000)), 128, 100, 65536
nq = int(os.environ.get("BATCH", 1))
g = torch.Generator(device=dev).manual_seed(1)
corpus = torch.randn((n, d), generator=g, device=dev) / (d ** 0.5)
q = torch.randn((nq, d), generator=g, device=dev) / (d ** 0.5)


class Blocks:
  def __iter__(self):
    for lo in range(0, n, bs):
      yield corpus[lo:lo + bs]


st = ftk.Streaming(k=k).index_from_dataset(Blocks())
for _ in range(int(os.environ.get("CALLS", 6))):
  st(q)
torch.cuda.synchronize()
