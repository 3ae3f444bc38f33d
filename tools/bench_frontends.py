"""Measurements for the embedding front-end kernels: salted SipHash
bucketing of ids / strings and the combiner-lookup backward, each against the HBM roof, plus a
UnifiedEmbedding forward at C4-like batch.  One JSON object per line (evidence tool)."""

import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import recommenders_amd as tfrs  # noqa: E402
from recommenders_amd.layers import tpu_embedding_layer as tel  # noqa: E402
from recommenders_amd.layers.feature_multiplexing import UnifiedEmbedding, UnifiedEmbeddingConfig  # noqa: E402

HBM_PEAK = 8.0e12


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
  lib_hash = tfrs.layers.hashing.Hashing(num_bins=1_000_003, salt=[3, 1])
  n = 1 << 26
  for name, hi, dt in (("ids<10M i64", 10_000_000, torch.int64), ("ids<10M i32", 10_000_000, torch.int32),
                       ("ids<2^62 i64", 2**62, torch.int64)):
    ids = torch.randint(0, hi, (n,), generator=g, device=dev, dtype=torch.int64).to(dt)
    t = timeit(lambda: lib_hash(ids))
    byts = n * (ids.element_size() + 8)
    emit(op=f"hash_bucket_strong_ids ({name})", n=n, ms=t * 1e3, gids_per_s=n / t / 1e9,
         gbps=byts / t / 1e9, frac_hbm_peak=byts / t / HBM_PEAK, algorithmic_bytes=byts)
  # strings: 1M strings of 8..23 bytes, packed once on the host; time the kernel call only
  rng = np.random.default_rng(0)
  ns = 1_000_000
  lens = rng.integers(8, 24, size=ns)
  offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
  blob = torch.from_numpy(rng.integers(32, 127, size=offsets[-1], dtype=np.uint8)).to(dev)
  d_off = torch.from_numpy(offsets).to(dev)
  out = torch.empty((ns,), dtype=torch.int64, device=dev)
  from recommenders_amd import _lib
  lib = _lib.load()
  t = timeit(lambda: _lib.check(lib.tfrs_hash_bucket_strong_bytes(
      _lib.ptr(blob), _lib.ptr(d_off), ns, 1_000_003, 3, 1, _lib.ptr(out), _lib.current_stream())))
  byts = int(offsets[-1]) + ns * 16
  emit(op="hash_bucket_strong_bytes (1M strings, 8..23 B)", n=ns, ms=t * 1e3,
       gstrings_per_s=ns / t / 1e9, gbps=byts / t / 1e9, frac_hbm_peak=byts / t / HBM_PEAK)

  # combiner backward: bags of 8, D = 128
  n = 65536 * 26
  d, bag = 128, 8
  nb = n // bag
  splits = torch.arange(0, nb + 1, device=dev) * bag
  go = torch.randn((nb, d), generator=g, device=dev)
  w = torch.rand((n,), generator=g, device=dev) + 0.5
  for comb, ww in (("sum", None), ("mean", w)):
    t = timeit(lambda: tel.segment_reduce_grad_rows(go, splits, ww, comb, n))
    byts = nb * d * 4 + n * d * 4 + (nb + 1) * 8 + (n * 4 if ww is not None else 0)
    emit(op=f"embedding_segment_reduce_bwd ({comb}{', weighted' if ww is not None else ''})",
         nnz=n, bags=nb, dim=d, ms=t * 1e3, gbps=byts / t / 1e9, frac_hbm_peak=byts / t / HBM_PEAK)

  # UnifiedEmbedding forward: 26 integer features x 2 chunks of 64 dims, 8 shared 1M-row tables
  cfg = UnifiedEmbeddingConfig(buckets_per_table=1_000_000, dim_per_table=64, num_tables=8, name="ue")
  B = 65536
  feats = {}
  for i in range(26):
    cfg.add_feature(f"f{i}", 2)
    feats[f"f{i}"] = torch.randint(0, 10_000_000, (B,), generator=g, device=dev)
  layer = UnifiedEmbedding(cfg, optimizer=None)
  lookups = 26 * 2 * B
  byts = 26 * B * 8 + lookups * 64 * 4 * 2          # ids once + row read + row written
  for fuse in (True, False):
    layer.fuse = fuse
    with torch.no_grad():
      t = timeit(lambda: layer(feats), iters=10)
    emit(op=f"UnifiedEmbedding forward, fuse={fuse} (26 features x 2 chunks x 64, B=65536, 8 tables x 1M)",
         ms=t * 1e3, lookups=lookups, lookups_per_s=lookups / t, algorithmic_bytes=byts,
         gbps=byts / t / 1e9, frac_hbm_peak=byts / t / HBM_PEAK)
  # one big feature: the kernel alone, no per-feature launch overhead
  cfg1 = UnifiedEmbeddingConfig(buckets_per_table=4_000_000, dim_per_table=64, num_tables=4, name="one")
  cfg1.add_feature("f", 4)
  big = UnifiedEmbedding(cfg1, optimizer=None)
  ids = torch.randint(0, 2**40, (1 << 21,), generator=g, device=dev)
  with torch.no_grad():
    t = timeit(lambda: big({"f": ids}))
  lookups = 4 * ids.numel()
  byts = ids.numel() * 8 + lookups * (64 * 4 * 2 + 8)     # + the kept buckets
  emit(op="unified_embedding_fwd kernel (2M ids x 4 chunks x 64, 4 tables x 4M)", ms=t * 1e3,
       lookups_per_s=lookups / t, algorithmic_bytes=byts, gbps=byts / t / 1e9,
       frac_hbm_peak=byts / t / HBM_PEAK)


if __name__ == "__main__":
  main()
