"""GPU parity tests at the sizes BASELINE.json's configs name (default kernel configuration
only; `tests/test_topk_gpu.py` runs the small cases in all three filter modes).  Oracle
comparisons are on sampled rows so that every test stays well under a minute."""

import numpy as np
import pytest

from oracle import topk as o_topk

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _np(x):
  return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)


def _ftk():
  from recommenders_amd.layers import factorized_top_k
  return factorized_top_k


def test_config3_shard_bruteforce_and_streaming_vs_oracle():
  """BASELINE configs[2], one GPU's shard: 12.5M x 128 corpus, top-100.  BruteForce, Streaming
  over 65536-row device blocks (packed images cached -> one search), and Streaming block by
  block (cache off) must agree with each other on the whole 8192-query batch and with the
  oracle (layers/factorized_top_k.py:404-509,586-607) on 16 sampled queries, bit for bit."""
  ftk = _ftk()
  g = torch.Generator(device="cuda").manual_seed(1234)
  n, d, nq, k, bs = 12_500_000, 128, 8192, 100, 65536
  blocks = [torch.randn((min(bs, n - lo), d), generator=g, device="cuda") / 11.3
            for lo in range(0, n, bs)]
  q = torch.randn((nq, d), generator=g, device="cuda") / 11.3
  bf = ftk.BruteForce(k=k).index_from_dataset(blocks, total_rows=n)     # reserve + append ingest
  s, i = bf(q)
  assert bf.last_redo_count() == 0
  st = ftk.Streaming(k=k).index_from_dataset(blocks)
  s1, i1 = st(q)
  assert st._cache is not None                                           # cached path taken
  assert torch.equal(s, s1) and torch.equal(i, i1)
  s1b, i1b = st(q)                                                       # second call: cache hit
  assert torch.equal(s, s1b) and torch.equal(i, i1b)
  st2 = ftk.Streaming(k=k, cache_packed_blocks=False).index_from_dataset(blocks)
  s2, i2 = st2(q[:512])                                                  # block-by-block path
  assert torch.equal(s[:512], s2) and torch.equal(i[:512], i2)
  sample = np.r_[0:8, 300:304, nq - 4:nq]
  corpus_host = torch.cat(blocks).cpu().numpy()
  es, ei = o_topk.brute_force(q[sample].cpu().numpy(), corpus_host, k)
  np.testing.assert_array_equal(_np(i)[sample], ei)
  np.testing.assert_array_equal(_np(s)[sample], es)


def test_streaming_cache_follows_block_changes():
  """The packed-image cache is keyed by (storage, shape, version) of every block: an in-place
  write, a replaced block or a different block count rebuilds it; identifiers ride along."""
  ftk = _ftk()
  rng = np.random.default_rng(3)
  n, d, nq, k = 50_000, 32, 40, 20
  c = (rng.normal(size=(n, d)) / 5).astype(np.float32)
  q = (rng.normal(size=(nq, d)) / 5).astype(np.float32)
  ids = (np.arange(n) * 5 + 1).astype(np.int64)
  blocks = [torch.as_tensor(c[lo:lo + 7000]).cuda() for lo in range(0, n, 7000)]
  id_blocks = [torch.as_tensor(ids[lo:lo + 7000]).cuda() for lo in range(0, n, 7000)]
  layer = ftk.Streaming(k=k).index_from_dataset(list(zip(id_blocks, blocks)))
  s, got = layer(q)
  es, ei = o_topk.brute_force(q, c, k, ids)
  np.testing.assert_array_equal(_np(got), ei)
  np.testing.assert_array_equal(_np(s), es)
  first = layer._cache
  layer(q)
  assert layer._cache is first
  blocks[2].mul_(-1.0)                                   # in-place change of one block
  c[14000:21000] *= -1.0
  s, got = layer(q)
  assert layer._cache is not first
  es, ei = o_topk.brute_force(q, c, k, ids)
  np.testing.assert_array_equal(_np(got), ei)
  np.testing.assert_array_equal(_np(s), es)
  # exclusions go through the cached index as well
  excl = ei[:, :3]
  s3, g3 = layer.query_with_exclusions(q, excl, k=5)
  sc = o_topk.scores(q, c)
  es3, ei3 = o_topk.exclude(*o_topk.brute_force(q, c, 5 + 3, ids), excl, 5)
  np.testing.assert_array_equal(_np(g3), ei3)
  np.testing.assert_array_equal(_np(s3), es3)
  del sc


def test_index_reserve_append_equals_index_set():
  """tfrs_index_reserve / tfrs_index_append (streamed ingest, ragged blocks that straddle the
  128-row stage boundaries) build the same index as tfrs_index_set."""
  ftk = _ftk()
  rng = np.random.default_rng(8)
  n, d, k = 70_001, 40, 50
  c = (rng.normal(size=(n, d)) * np.exp(0.3 * rng.normal(size=(n, 1)))).astype(np.float32)
  q = rng.normal(size=(33, d)).astype(np.float32)
  cuts = [0, 1, 130, 4097, 20000, 20001, 65536, n]
  blocks = [c[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
  a = ftk.BruteForce(k=k).index(c)
  b = ftk.BruteForce(k=k).index_from_dataset(blocks, total_rows=n)
  np.testing.assert_array_equal(_np(b.candidates()), c)                  # unpack round trip
  sa, ia = a(q)
  sb, ib = b(q)
  assert torch.equal(sa, sb) and torch.equal(ia, ib)
  es, ei = o_topk.brute_force(q, c, k)
  np.testing.assert_array_equal(_np(ib), ei)
  np.testing.assert_array_equal(_np(sb), es)
  with pytest.raises(ValueError, match="more than total_rows"):
    ftk.BruteForce(k=k).index_from_dataset(blocks, total_rows=n - 1)


def test_identical_queries_fill_the_survivor_queue():
  """A batch of identical queries makes all 64 lanes of a wave hot on the same tiles: the
  filter kernel's LDS survivor queue overflows and drains mid-stage.  Results must not
  change (and no query may fall back to the exact-redo path because of it)."""
  ftk = _ftk()
  rng = np.random.default_rng(17)
  n, d, nq, k = 300_000, 64, 1024, 100
  c = (rng.normal(size=(n, d)) / 8).astype(np.float32)
  q1 = (rng.normal(size=(1, d)) / 8).astype(np.float32)
  q = np.repeat(q1, nq, axis=0)
  layer = ftk.BruteForce(k=k).index(c)
  s, i = layer(q)
  es, ei = o_topk.brute_force(q1, c, k)
  assert layer.last_redo_count() == 0
  np.testing.assert_array_equal(_np(i), np.repeat(ei, nq, axis=0))
  np.testing.assert_array_equal(_np(s), np.repeat(es, nq, axis=0))
