"""Times tfrs_topk_merge_strided (the post-all-gather merge of ShardedBruteForce) for 2/4/8 parts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd import _lib
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
lib = _lib.load()
nq, k = 8192, 100
for world in (2, 4, 8):
  scores = torch.randn((world, nq, k), generator=g, device=dev).sort(dim=2, descending=True).values
  rows = torch.randint(0, 1_000_000, (world, nq, k), generator=g, device=dev, dtype=torch.int32)
  rows = rows + (torch.arange(world, device=dev, dtype=torch.int32) * 1_000_000).view(-1, 1, 1)
  gathered = torch.stack([scores.view(torch.int32), rows], dim=1).contiguous()   # [world, 2, nq, k]
  out_s = torch.empty((nq, k), dtype=torch.float32, device=dev)
  out_i = torch.empty((nq, k), dtype=torch.int32, device=dev)
  base = gathered.view(-1)
  def run():
    _lib.check(lib.tfrs_topk_merge_strided(base.data_ptr(), base.data_ptr() + nq * k * 4, world,
                                           2 * nq * k, nq, k, k, _lib.ptr(out_s), _lib.ptr(out_i),
                                           _lib.current_stream()))
  for _ in range(3): run()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(20): run()
  b.record(); torch.cuda.synchronize()
  ref = torch.topk(scores.permute(1, 0, 2).reshape(nq, world * k), k, dim=1).values
  print(f"world={world} merge {a.elapsed_time(b) / 20 * 1e3:.1f} us  max|diff|={float((ref - out_s).abs().max()):.1e}", flush=True)
