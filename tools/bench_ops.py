"""Secondary measurements for the other rows of the hot path: embedding
gather / segment-sum / scatter-add (HBM roof), the in-batch softmax train step (C1),
Cross (C4) and DotInteraction (C5), each as absolute rate and fraction of its roofline.
Writes one JSON object per line.  Development / evidence tool; bench.py stays the headline."""

import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import recommenders_amd as tfrs  # noqa: E402
from recommenders_amd.layers import embedding as emb  # noqa: E402
from recommenders_amd.tasks.retrieval import in_batch_softmax_loss  # noqa: E402

HBM_PEAK = 8.0e12
F32_MFMA_PEAK = 157.3e12
F16_MFMA_PEAK = 2500e12
# the default in-batch softmax path runs every GEMM as three fp16 MFMA products (hi*hi + hi*lo +
# lo*hi): `tflops` stays ALGORITHMIC (2 B^2 D forward, 10 B^2 D forward + backward); the MFMA
# pipe executes 3x that, which `frac_f16_mfma_peak` prices against the dense fp16 peak


def timeit(fn, warmup=3, iters=20):
  for _ in range(warmup):
    fn()
  torch.cuda.synchronize()
  start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  start.record()
  for _ in range(iters):
    fn()
  stop.record()
  torch.cuda.synchronize()
  return start.elapsed_time(stop) / iters * 1e-3


def emit(**kw):
  print(json.dumps(kw), flush=True)


def main():
  dev = torch.device("cuda", 0)
  g = torch.Generator(device=dev).manual_seed(0)
  small = len(sys.argv) > 1 and sys.argv[1] == "small"

  # ---- embedding gather, C4-shaped: 26 tables x 1M x 128 (one fused 26M-row table) ----
  vocab, d, n = (26_000_000, 128, 65536 * 26) if not small else (1_000_000, 128, 65536)
  table = torch.empty((vocab, d), device=dev).uniform_(-0.05, 0.05)
  ids = torch.randint(0, vocab, (n,), generator=g, device=dev)
  t = timeit(lambda: emb.gather_rows(table, ids))
  byts = n * (2 * d * 4 + 8)
  emit(op="embedding_gather", rows=n, dim=d, vocab=vocab, ms=t * 1e3, gbps=byts / t / 1e9,
       frac_hbm_peak=byts / t / HBM_PEAK, algorithmic_bytes=byts)
  ids32 = ids.to(torch.int32)
  t = timeit(lambda: emb.gather_rows(table, ids32))
  byts = n * (2 * d * 4 + 4)
  emit(op="embedding_gather_i32", rows=n, dim=d, ms=t * 1e3, gbps=byts / t / 1e9,
       frac_hbm_peak=byts / t / HBM_PEAK)
  # D = 64 and D = 32 rows (C2 / C5 widths)
  for dd in (64, 32):
    tb = table.view(-1, dd)
    idd = torch.randint(0, tb.shape[0], (n * 2,), generator=g, device=dev)
    t = timeit(lambda: emb.gather_rows(tb, idd))
    byts = idd.numel() * (2 * dd * 4 + 8)
    emit(op=f"embedding_gather_d{dd}", rows=idd.numel(), dim=dd, ms=t * 1e3, gbps=byts / t / 1e9,
         frac_hbm_peak=byts / t / HBM_PEAK)
  # segment sum: bags of 8 ids
  bag = 8
  nb = n // bag
  splits = torch.arange(0, nb + 1, device=dev) * bag
  t = timeit(lambda: emb.embedding_lookup_sparse(table, ids, splits, combiner="sum"))
  byts = n * d * 4 + nb * d * 4 + n * 8 + (nb + 1) * 8
  emit(op="embedding_segment_sum", nnz=n, bags=nb, dim=d, ms=t * 1e3, gbps=byts / t / 1e9,
       frac_hbm_peak=byts / t / HBM_PEAK)
  # scatter-add backward (incl. the torch sort it currently relies on)
  go = torch.randn((n, d), generator=g, device=dev)
  t = timeit(lambda: emb.scatter_add_rows(go, ids, vocab), iters=5)
  emit(op="embedding_scatter_add_bwd(dense grad, incl. zero-fill of the 13 GB table + own sort)", rows=n, dim=d, ms=t * 1e3)
  acc = torch.full_like(table, 0.1)
  t = timeit(lambda: emb.adagrad_sparse_update_(table, acc, go, ids, 0.5), iters=5)
  byts = n * d * 4 + 4 * n * d * 4
  emit(op="embedding_sparse_adagrad(incl. own radix sort)", rows=n, dim=d, ms=t * 1e3, gbps=byts / t / 1e9)
  del table, acc, go

  # ---- C1: MovieLens-shaped in-batch softmax train step ----
  B, D, V = 4096, 64, 2000

  class TwoTower(tfrs.Model):
    def __init__(self):
      super().__init__()
      self.user_model = emb.Embedding(V, D)
      self.item_model = emb.Embedding(V, D)
      self.task = tfrs.tasks.Retrieval()

    def compute_loss(self, inputs, training=False):
      return self.task(self.user_model(inputs[0]), self.item_model(inputs[1]), compute_metrics=False)

  model = TwoTower()
  model.compile(optimizer=tfrs.optimizers.Adagrad(model.parameters(), learning_rate=0.5))
  batch = (torch.randint(0, 943, (B,), generator=g, device=dev),
           torch.randint(0, 1682, (B,), generator=g, device=dev))
  t = timeit(lambda: model.train_step(batch), iters=50)
  emit(op="C1 tfrs.Model.train_step (gather + in-batch softmax fwd/bwd + sparse Adagrad), compute_metrics=False",
       batch=B, dim=D, ms=t * 1e3, steps_per_s=1.0 / t)
  q = torch.randn((B, D), generator=g, device=dev, requires_grad=True)
  c = torch.randn((B, D), generator=g, device=dev, requires_grad=True)
  t = timeit(lambda: in_batch_softmax_loss(q, c), iters=50)
  emit(op="inbatch_softmax_fwd", batch=B, dim=D, ms=t * 1e3, tflops=2.0 * B * B * D / t / 1e12,
       frac_f16_mfma_peak=3 * 2.0 * B * B * D / t / F16_MFMA_PEAK)

  def fb():
    q.grad = None
    c.grad = None
    in_batch_softmax_loss(q, c).backward()

  t = timeit(fb, iters=50)
  emit(op="inbatch_softmax_fwd+bwd", batch=B, dim=D, ms=t * 1e3,
       tflops=10.0 * B * B * D / t / 1e12, frac_f16_mfma_peak=3 * 10.0 * B * B * D / t / F16_MFMA_PEAK)
  for Bb in (16384, 65536):
    qq = torch.randn((Bb, D), generator=g, device=dev)
    cc = torch.randn((Bb, D), generator=g, device=dev)
    t = timeit(lambda: in_batch_softmax_loss(qq, cc), iters=5)
    emit(op="inbatch_softmax_fwd", batch=Bb, dim=D, ms=t * 1e3, tflops=2.0 * Bb * Bb * D / t / 1e12,
         frac_f16_mfma_peak=3 * 2.0 * Bb * Bb * D / t / F16_MFMA_PEAK)

  # ---- C4: Cross layer, B = 65536, d = 3456 ----
  Bc, dc = (65536, 3456) if not small else (8192, 1024)
  x0 = torch.randn((Bc, dc), generator=g, device=dev)
  layer = tfrs.layers.feature_interaction.Cross()
  with torch.no_grad():
    t = timeit(lambda: layer(x0, x0), iters=5)
  fl = 2.0 * Bc * dc * dc
  # large Cross products run on the split-fp16 GEMM: 3 fp16 MFMA products per f32 product
  emit(op="cross_fwd", batch=Bc, dim=dc, ms=t * 1e3, tflops=fl / t / 1e12,
       frac_f16_mfma_peak=3 * fl / t / F16_MFMA_PEAK)
  x0g = x0.clone().requires_grad_(True)

  def cross_fb():
    x0g.grad = None
    layer.zero_grad(set_to_none=True)
    layer(x0g, x0g).sum().backward()

  t = timeit(cross_fb, warmup=1, iters=3)
  emit(op="cross_fwd+bwd through autograd (training pair: 3 GEMMs, u saved by the forward; incl. sum() and its backward)",
       batch=Bc, dim=dc, ms=t * 1e3, tflops=3 * fl / t / 1e12, frac_f16_mfma_peak=3 * 3 * fl / t / F16_MFMA_PEAK)
  del x0, x0g, layer

  # ---- C5: DotInteraction, B = 131072, F = 101, D = 32 ----
  Bd, F, Dd = (131072, 101, 32) if not small else (16384, 27, 16)
  x = torch.randn((Bd, F, Dd), generator=g, device=dev)
  from recommenders_amd.layers.feature_interaction.dot_interaction import _DotInteractionFn
  t = timeit(lambda: _DotInteractionFn.apply(x, False, False), iters=5)
  out_dim = F * (F - 1) // 2
  byts = Bd * F * Dd * 4 + Bd * out_dim * 4
  emit(op="dot_interaction_fwd", batch=Bd, features=F, dim=Dd, ms=t * 1e3, gbps=byts / t / 1e9,
       frac_hbm_peak=byts / t / HBM_PEAK, gflop_full_gram=2.0 * Bd * F * F * Dd / 1e9)
  xg = x.clone().requires_grad_(True)

  def dot_fb():
    xg.grad = None
    _DotInteractionFn.apply(xg, False, False).sum().backward()

  t = timeit(dot_fb, warmup=1, iters=3)
  emit(op="dot_interaction_fwd+bwd", batch=Bd, features=F, dim=Dd, ms=t * 1e3)
  del x, xg

  # ---- FactorizedTopK metric update on a BruteForce index (C1-sized eval batch) ----
  cand = torch.randn((1682, 64), generator=g, device=dev) / 8.0
  metric = tfrs.metrics.FactorizedTopK(tfrs.layers.factorized_top_k.BruteForce(k=100).index(cand))
  qe = torch.randn((4096, 64), generator=g, device=dev) / 8.0
  ce = cand[torch.randint(0, 1682, (4096,), generator=g, device=dev)]
  t = timeit(lambda: metric.update_state(qe, ce), iters=20)
  emit(op="FactorizedTopK.update_state (batch 4096 vs 1682 candidates, ks=(1,5,10,50,100))", ms=t * 1e3)


if __name__ == "__main__":
  main()
