"""Row-sharded embedding lookup at one GPU's share of BASELINE configs[4] (100 tables x 10M rows x dim
32 over 8 GPUs: a 125M-row shard; batch 131072 x 100 features = 13.1M lookups per step, 1.64M per
rank under data parallelism).  One rank (no exchange: what is timed is everything AROUND the two
all-to-alls): the HIP owner bucketing (csrc/shard_route.hip), the owner-side gather, the gather
through the inverse permutation, and the backward's re-ordering + fused sparse Adagrad -- next to
the torch routing it replaces (argsort + bincount + index ops)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recommenders_amd as tfrs
from recommenders_amd.layers import sharded_embedding as se
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)


def events(fn, iters=20, warmup=3):
  for _ in range(warmup): fn()
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
  for a, b in ev:
    a.record(); fn(); b.record()
  torch.cuda.synchronize()
  ms = sorted(a.elapsed_time(b) for a, b in ev)
  return ms[len(ms) // 2]

rows, d = 125_000_000, 32
layer = se.ShardedEmbedding(rows, d)
opt = tfrs.optimizers.Adagrad(layer.parameters(), learning_rate=0.01)
for n, what in ((1_638_400, "one rank's lookups per step (16384 x 100)"), (13_107_200, "the whole batch (131072 x 100)")):
  ids = torch.randint(0, rows, (n,), generator=g, device=dev)
  w = torch.randn((n, d), generator=g, device=dev)
  for world in (1, 8):
    rpr = (rows + world - 1) // world
    t_hip = events(lambda: se._route_hip(ids, rows, rpr, world))
    t_torch = events(lambda: se._route_torch(ids, rows, rpr, world), iters=5, warmup=1)
    print(json.dumps({"op": "owner bucketing", "lookups": n, "world": world, "what": what,
                      "hip_ms": t_hip, "torch_argsort_bincount_ms": t_torch}), flush=True)

  def fwd():
    return layer(ids)

  def fwd_bwd():
    out = layer(ids)
    opt.zero_grad()
    out.backward(w)
    opt.step()

  t_f = events(fwd)
  t_fb = events(fwd_bwd, iters=10)
  fwd_bytes = n * (2 * d * 4 + 8) * 2          # owner gather + gather through perm (rows read + written twice)
  print(json.dumps({"op": "ShardedEmbedding lookup (1 rank, no exchange)", "lookups": n, "dim": d, "what": what,
                    "forward_ms": t_f, "forward_GBps_of_moved_bytes": fwd_bytes / (t_f * 1e-3) / 1e9,
                    "forward_backward_adagrad_ms": t_fb}), flush=True)
