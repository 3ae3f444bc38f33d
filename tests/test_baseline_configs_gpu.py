"""GPU parity tests at the sizes BASELINE.json's configs name (default kernel configuration
only; `tests/test_topk_gpu.py` runs the small cases in all three filter modes).  Oracle
comparisons are on sampled rows so that every test stays well under a minute."""

import numpy as np
import pytest

from oracle import topk as o_topk
from tests.conftest import float_gate

pytestmark = pytest.mark.gpu

# Gates relative to the sum of |terms| of each entry (tests/conftest.py float_gate), set at <= 4x the
# largest error observed on MI355X (profiles/r03_observed_errors.md).
# (observed maxima, round 3: Cross at configs[3] y 1.8e-7, dx0 1.9e-7, dx 2.3e-7, dW 5.7e-8, db 8e-9;
# low-rank / MultiLayerDCN 7e-8; in-batch softmax at B = 16384 / 65536 dq, dc 1.3e-6; wide dims
# 1.4e-6; DotInteraction at configs[4] forward 4.2e-7, backward 3.8e-7 on every kernel variant)
GATE_GEMM16 = 8e-7
GATE_SOFTMAX_BIG = 5e-6
GATE_DOT_C4 = {"fwd": 1.6e-6, "bwd": 1.5e-6}

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


def test_streaming_train_eval_train_eval_sees_fresh_candidates():
  """evaluation after further training must score against the CURRENT
  candidate embeddings.  (1) a lazily mapped dataset (`candidates.map(item_model)`) whose blocks
  are fresh tensors on every pass -- very likely at the addresses of the previous pass's freed
  blocks -- is never cached; (2) a list of detached views of the trained table is cached, and the
  fused Adagrad kernels (raw-pointer writes) bump the table's version counter so the cache is
  rebuilt.  Both against the oracle after each of two training phases."""
  import recommenders_amd as tfrs
  ftk = _ftk()
  from recommenders_amd.layers import embedding as emb
  rng = np.random.default_rng(12)
  vocab, d, nq, k = 70_000, 32, 64, 10
  item_model = emb.Embedding(vocab, d).cuda()
  ids = torch.arange(vocab, device="cuda")

  class Mapped:                                   # movies.batch(8192).map(item_model)
    def __iter__(self):
      for lo in range(0, vocab, 8192):
        with torch.no_grad():
          yield item_model(ids[lo:lo + 8192])

  lazy = ftk.Streaming(k=k).index_from_dataset(Mapped())
  views = ftk.Streaming(k=k).index_from_dataset(
      [item_model.embeddings.detach()[lo:lo + 8192] for lo in range(0, vocab, 8192)])
  opt = tfrs.optimizers.Adagrad(item_model.parameters(), learning_rate=0.5)
  q = (rng.normal(size=(nq, d)) / 5).astype(np.float32)
  tq = torch.as_tensor(q).cuda()
  for phase in range(3):
    table = _np(item_model.embeddings.detach())
    es, ei = o_topk.brute_force(q, table, k)
    for layer in (lazy, views):
      s, i = layer(tq)
      np.testing.assert_array_equal(_np(i), ei)
      np.testing.assert_array_equal(_np(s), es)
    assert lazy._cache is None and views._cache is not None
    # "training": a sparse Adagrad step that moves a few thousand rows, among them the current winners
    hit = torch.as_tensor(np.unique(np.r_[ei.ravel(), rng.integers(0, vocab, size=4096)]), device="cuda")
    out = item_model(hit)
    (out * torch.as_tensor(rng.normal(size=(hit.numel(), d)).astype(np.float32), device="cuda")).sum().backward()
    opt.step()
    opt.zero_grad()
    assert not np.array_equal(_np(item_model.embeddings.detach()), table)


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


# ------------------------------------------------------------------------------------------------
# Cross at BASELINE configs[3] (B = 65536, d = 3456): the 256 x 256-tile split-fp16 GEMM at its real
# grid, forward + the fused backward, against float64 (oracle formulas of
# oracle/feature_interaction.py: dcn.py:151-186) on sampled rows and the full dW.
# ------------------------------------------------------------------------------------------------
def test_cross_config4_forward_backward_vs_float64():
  import recommenders_amd as tfrs
  g = torch.Generator(device="cuda").manual_seed(77)
  b, d, diag = 65536, 3456, 0.25
  x0 = torch.randn((b, d), generator=g, device="cuda") * 0.5
  x = torch.randn((b, d), generator=g, device="cuda") * 0.5
  layer = tfrs.layers.feature_interaction.Cross(diag_scale=diag)
  layer.build((b, d), torch.device("cuda"))
  with torch.no_grad():
    layer.bias.uniform_(-0.1, 0.1, generator=g)
  x0g, xg = x0.clone().requires_grad_(True), x.clone().requires_grad_(True)
  y = layer(x0g, xg)
  dy = torch.randn((b, d), generator=g, device="cuda")
  y.backward(dy)
  rows = torch.tensor(np.r_[0:24, 30000:30016, b - 24:b], device="cuda")
  w64, b64 = layer.kernel.detach().double(), layer.bias.detach().double()
  # sampled rows: y, dx0, dx
  xr, x0r, dyr = x[rows].double(), x0[rows].double(), dy[rows].double()
  z = xr @ w64 + b64 + diag * xr
  # yardsticks: the sums of |terms| (oracle/feature_interaction.py cross_yardsticks, on the GPU)
  za = xr.abs() @ w64.abs() + b64.abs() + diag * xr.abs()
  float_gate("cross_c3.y", y[rows], x0r * z + xr, x0r.abs() * za + xr.abs(), GATE_GEMM16)
  dz = dyr * x0r
  float_gate("cross_c3.dx0", x0g.grad[rows], dyr * z, dyr.abs() * za, GATE_GEMM16)
  float_gate("cross_c3.dx", xg.grad[rows], dz @ w64.t() + dyr + diag * dz,
             dz.abs() @ w64.abs().t() + dyr.abs() + diag * dz.abs(), GATE_GEMM16)
  # full dW and db in float64 on the GPU, in row chunks (x^T dz)
  dw = torch.zeros((d, d), dtype=torch.float64, device="cuda")
  dwa = torch.zeros((d, d), dtype=torch.float64, device="cuda")
  dbias = torch.zeros((d,), dtype=torch.float64, device="cuda")
  dba = torch.zeros((d,), dtype=torch.float64, device="cuda")
  for lo in range(0, b, 8192):
    dzc = (dy[lo:lo + 8192] * x0[lo:lo + 8192]).double()
    xc = x[lo:lo + 8192].double()
    dw += xc.t() @ dzc
    dwa += xc.abs().t() @ dzc.abs()
    dbias += dzc.sum(dim=0)
    dba += dzc.abs().sum(dim=0)
  float_gate("cross_c3.dW", layer.kernel.grad, dw, dwa, GATE_GEMM16)
  float_gate("cross_c3.db", layer.bias.grad, dbias, dba, GATE_GEMM16)


@pytest.mark.parametrize("gemm_mode", ["f32", "f16"])
def test_cross_lowrank_and_multilayer_gradients(gemm_mode, monkeypatch):
  """Low-rank Cross (dcn.py:176-180) and MultiLayerDCN (multi_layer_dcn.py:136-153): every
  parameter and input gradient against float64 autograd of the reference formula."""
  import recommenders_amd as tfrs
  monkeypatch.setenv("TFRS_GEMM_MODE", gemm_mode)
  g = torch.Generator(device="cuda").manual_seed(5)
  b, d, p = 700, 160, 24
  x0 = torch.randn((b, d), generator=g, device="cuda")
  x = torch.randn((b, d), generator=g, device="cuda")
  layer = tfrs.layers.feature_interaction.Cross(projection_dim=p, diag_scale=0.1)
  layer.build((b, d), torch.device("cuda"))
  with torch.no_grad():
    layer.bias.uniform_(-0.1, 0.1, generator=g)
  x0g, xg = x0.clone().requires_grad_(True), x.clone().requires_grad_(True)
  dy = torch.randn((b, d), generator=g, device="cuda")
  layer(x0g, xg).backward(dy)
  u, v, bb = (t.detach().double().requires_grad_(True) for t in (layer.kernel_u, layer.kernel_v, layer.bias))
  x0d, xd = x0.double().requires_grad_(True), x.double().requires_grad_(True)
  ref = x0d * (xd @ u @ v + bb + 0.1 * xd) + xd
  ref.backward(dy.double())
  # yardsticks: the same network on absolute values with |dy| upstream -- every operation is a
  # product or a sum of non-negative numbers, so autograd returns the sum of |terms| per entry
  ua, va, ba, x0a, xa = (t.detach().abs().requires_grad_(True) for t in (u, v, bb, x0d, xd))
  (x0a * (xa @ ua @ va + ba + 0.1 * xa) + xa).backward(dy.double().abs())
  for got, want, yard, what in ((x0g.grad, x0d.grad, x0a.grad, "dx0"), (xg.grad, xd.grad, xa.grad, "dx"),
                                (layer.kernel_u.grad, u.grad, ua.grad, "dU"), (layer.kernel_v.grad, v.grad, va.grad, "dV"),
                                (layer.bias.grad, bb.grad, ba.grad, "db")):
    assert got is not None, what
    float_gate(f"cross_lowrank_{gemm_mode}.{what}", got, want, yard, GATE_GEMM16)
  # MultiLayerDCN: 3 stacked low-rank layers, gradients of every parameter
  mdcn = tfrs.layers.feature_interaction.MultiLayerDCN(projection_dim=8, num_layers=3)
  xin = torch.randn((b, d), generator=g, device="cuda").requires_grad_(True)
  out = mdcn(xin)
  out.backward(dy)
  params = [p_ for p_ in mdcn.parameters()]
  refs = [p_.detach().double().requires_grad_(True) for p_ in params]
  names = [n for n, _ in mdcn.named_parameters()]
  byname = dict(zip(names, refs))
  xd = xin.detach().double().requires_grad_(True)
  xl = xd
  for i in range(3):                                          # multi_layer_dcn.py:147-153
    ui = byname[[n for n in names if "u_kernels" in n and n.endswith(str(i))][0]]
    vi = byname[[n for n in names if "v_kernels" in n and n.endswith(str(i))][0]]
    bi = [n for n in names if "bias" in n and n.endswith(str(i))]
    prod = xl @ ui @ vi
    if bi:
      prod = prod + byname[bi[0]]
    xl = xd * prod + xl
  xl.backward(dy.double())
  arefs = [r.detach().abs().requires_grad_(True) for r in refs]      # the abs-network again
  abyname = dict(zip(names, arefs))
  xa = xd.detach().abs().requires_grad_(True)
  xla = xa
  for i in range(3):
    ui = abyname[[n for n in names if "u_kernels" in n and n.endswith(str(i))][0]]
    vi = abyname[[n for n in names if "v_kernels" in n and n.endswith(str(i))][0]]
    bi = [n for n in names if "bias" in n and n.endswith(str(i))]
    prod = xla @ ui @ vi
    if bi:
      prod = prod + abyname[bi[0]]
    xla = xa * prod + xla
  xla.backward(dy.double().abs())
  float_gate(f"mdcn_{gemm_mode}.y", out.detach(), xl.detach(), xla.detach(), GATE_GEMM16)
  for p_, r, ra, n in zip(params, refs, arefs, names):
    assert p_.grad is not None, n
    float_gate(f"mdcn_{gemm_mode}.d{n}", p_.grad, r.grad, ra.grad, GATE_GEMM16)
  float_gate(f"mdcn_{gemm_mode}.dx", xin.grad, xd.grad, xa.grad, GATE_GEMM16)


@pytest.mark.parametrize("bsz", [16384, 65536])
def test_inbatch_softmax_large_batch_vs_float64(bsz):
  """In-batch softmax (tasks/retrieval.py:172-210) at the batch sizes where the 8-wave
  workgroups switch on: loss, and sampled rows of dq / dc, against float64 (chunked on the GPU)."""
  from recommenders_amd.tasks.retrieval import in_batch_softmax_loss
  g = torch.Generator(device="cuda").manual_seed(bsz)
  d = 64
  q = (torch.randn((bsz, d), generator=g, device="cuda") * 0.3).requires_grad_(True)
  c = (torch.randn((bsz, d), generator=g, device="cuda") * 0.3).requires_grad_(True)
  loss = in_batch_softmax_loss(q, c)
  loss.backward()
  q64, c64 = q.detach().double(), c.detach().double()
  lse = torch.empty((bsz,), dtype=torch.float64, device="cuda")
  for lo in range(0, bsz, 4096):
    lse[lo:lo + 4096] = torch.logsumexp(q64[lo:lo + 4096] @ c64.t(), dim=1)
  pos = (q64 * c64).sum(dim=1)
  ref_loss = (lse - pos).sum().item()
  assert abs(loss.item() - ref_loss) <= 1e-5 * abs(ref_loss)
  rows = torch.tensor(np.r_[0:16, bsz // 2:bsz // 2 + 16, bsz - 16:bsz], device="cuda")
  # dq[i] = sum_j p_ij c_j - c_i
  p_rows = torch.exp(q64[rows] @ c64.t() - lse[rows, None])
  dq_ref = p_rows @ c64 - c64[rows]
  # yardstick: first-order propagation of a unit relative rounding error (oracle/retrieval.py
  # loss_grads): terms p_ij c_j and c_i, with p_ij carrying the absolute error of its logit and lse_i
  cond = q64[rows].abs() @ c64.abs().t()
  cond = cond + (p_rows * cond).sum(dim=1, keepdim=True)
  float_gate("softmax_big.dq", q.grad[rows], dq_ref, (p_rows * (1.0 + cond)) @ c64.abs() + c64[rows].abs(),
             GATE_SOFTMAX_BIG)
  # dc[j] = sum_i p_ij q_i - q_j  for sampled j
  p_cols = torch.exp(q64 @ c64[rows].t() - lse[:, None])            # [bsz, 48]
  dc_ref = p_cols.t() @ q64 - q64[rows]
  cond_c = q64.abs() @ c64[rows].abs().t()                            # [bsz, 48]
  lse_cond = torch.empty((bsz,), dtype=torch.float64, device="cuda")  # sum_j p_ij A_ij per query
  for lo in range(0, bsz, 4096):
    sl = slice(lo, lo + 4096)
    lse_cond[sl] = (torch.exp(q64[sl] @ c64.t() - lse[sl, None]) * (q64[sl].abs() @ c64.abs().t())).sum(dim=1)
  float_gate("softmax_big.dc", c.grad[rows], dc_ref,
             (p_cols * (1.0 + cond_c + lse_cond[:, None])).t() @ q64.abs() + q64[rows].abs(), GATE_SOFTMAX_BIG)


@pytest.mark.parametrize("vocab", [3_000_000, 400_000, 100_000])
def test_large_vocab_scatter_add_own_sort_and_bad_ids(vocab):
  """The large-vocabulary backward (own radix sort + segmented scatter-add, no torch.sort):
  bit-exact occurrence-order sums on a 3M-row table with heavy duplicates, int32 and int64 ids;
  ids outside [0, vocab) -- sequence padding, corrupt input -- are ignored and can never write
  outside the table (the row right behind the table is checked).  The three sizes take the sort's three
  digit widths: 3 M rows three 8-bit passes, 400 k rows two 10-bit passes, 100 k rows two 9-bit passes
  (26 M and 100 M rows -- three 9- and three 10-bit passes -- are in
  test_embedding_path_at_config_table_sizes)."""
  from recommenders_amd.layers import embedding as emb
  from oracle import embedding as o_emb
  rng = np.random.default_rng(21)
  n, d = 200_000, 32
  ids = np.where(rng.random(n) < 0.5, rng.integers(0, 5000, size=n), rng.integers(0, vocab, size=n))
  ids[::97] = -1
  ids[5::101] = vocab            # one past the end
  ids[7::103] = vocab + 12345
  g = rng.normal(size=(n, d)).astype(np.float32)
  ok = (ids >= 0) & (ids < vocab)
  ref = o_emb.scatter_add_grad(g[ok], ids[ok], vocab)
  # rows looked up fewer than 32 times are summed in occurrence order: bit for bit the oracle; longer runs are
  # cut into pieces of 32 sorted positions that are summed in parallel: float association only
  short = np.bincount(ids[ok], minlength=vocab) < 32
  for dtype in (np.int64, np.int32):
    got = _np(emb.scatter_add_rows(torch.as_tensor(g).cuda(), torch.as_tensor(ids.astype(dtype)).cuda(), vocab))
    np.testing.assert_array_equal(got[short], ref[short])
    np.testing.assert_allclose(got[~short], ref[~short], rtol=2e-6, atol=2e-6)
  # fused Adagrad on a table that has a guard row behind it
  backing = torch.zeros((vocab + 1, d), device="cuda")
  acc_backing = torch.full((vocab + 1, d), 0.1, device="cuda")
  table, accum = backing[:vocab], acc_backing[:vocab]
  emb.adagrad_sparse_update_(table, accum, torch.as_tensor(g).cuda(), torch.as_tensor(ids).cuda(), lr=0.5)
  assert float(backing[vocab].abs().max()) == 0.0 and float((acc_backing[vocab] - 0.1).abs().max()) == 0.0
  t_ref, a_ref = o_emb.adagrad_sparse_update(np.zeros((vocab, d), np.float32), np.full((vocab, d), 0.1, np.float32),
                                             g[ok], ids[ok], lr=0.5)
  np.testing.assert_allclose(_np(accum), a_ref, rtol=1e-6)
  np.testing.assert_allclose(_np(table), t_ref, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("d", [129, 200, 384])
def test_wide_embedding_dims_topk_and_retrieval(d):
  """Embedding dims above 128 (outside the fused kernels' register envelope; the reference has no
  limit, layers/factorized_top_k.py:320-333, tasks/retrieval.py:172-180) take the GEMM + select
  path: same top-K as the oracle (scores to 1e-5 relative: they are GEMM sums, not the fma chain),
  BruteForce / Streaming / exclusions / Retrieval loss and gradients."""
  import recommenders_amd as tfrs
  from oracle import retrieval as o_ret
  ftk = _ftk()
  rng = np.random.default_rng(d)
  n, nq, k = 70_000, 50, 60
  c = (rng.normal(size=(n, d)) / np.sqrt(d)).astype(np.float32)
  q = (rng.normal(size=(nq, d)) / np.sqrt(d)).astype(np.float32)
  es, ei = o_topk.brute_force(q, c, k)
  full = o_topk.scores(q, c)
  for layer in (ftk.BruteForce(k=k).index(c),
                ftk.Streaming(k=k).index_from_dataset([c[lo:lo + 9000] for lo in range(0, n, 9000)]),
                ftk.Streaming(k=k).index_from_dataset([torch.as_tensor(c[lo:lo + 9000]).cuda() for lo in range(0, n, 9000)])):
    s, i = layer(q)
    s, i = _np(s), _np(i)
    np.testing.assert_allclose(s, es, rtol=1e-5, atol=1e-6)
    # returned rows carry (to tolerance) the scores returned for them; order may only differ on near-ties
    np.testing.assert_allclose(np.take_along_axis(full, i.astype(np.int64), axis=1), s, rtol=1e-5, atol=1e-6)
    assert (i == ei).mean() > 0.995
  # Retrieval default loss at a wide dim: explicit-logits path
  b = 300
  qe = torch.as_tensor(q[:b % nq + nq][:nq]).cuda()
  qe = torch.as_tensor((rng.normal(size=(b, d)) / np.sqrt(d)).astype(np.float32)).cuda().requires_grad_(True)
  ce = torch.as_tensor((rng.normal(size=(b, d)) / np.sqrt(d)).astype(np.float32)).cuda().requires_grad_(True)
  loss = tfrs.tasks.Retrieval()(qe, ce, compute_metrics=False)
  loss.backward()
  ref = o_ret.loss(_np(qe.detach()), _np(ce.detach()))
  assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))
  dq, dc, dq_y, dc_y = o_ret.loss_grads(_np(qe.detach()), _np(ce.detach()), return_yardsticks=True)
  float_gate("softmax_wide.dq", _np(qe.grad), dq, dq_y, GATE_SOFTMAX_BIG)
  float_gate("softmax_wide.dc", _np(ce.grad), dc, dc_y, GATE_SOFTMAX_BIG)


def test_cluster_ordered_corpus_stays_exact():
  """Rows grouped by cluster are the adversarial order for the sampled-bin threshold: a query's
  whole top-K sits in a few consecutive stages, so its survivors overflow their list segments
  (per-query overflow lists) or the list itself (exact-redo path).  Slow, but results stay exact.
  A moderately clustered corpus (second part: clusters of ~60 rows) must not need the redo path."""
  ftk = _ftk()
  rng = np.random.default_rng(99)
  n, d, nq, k, ncl = 400_000, 64, 512, 100, 200
  centers = rng.normal(size=(ncl, d)) / np.sqrt(d)
  cl = np.sort(rng.integers(0, ncl, size=n))                       # grouped by cluster
  c = (centers[cl] + 0.35 * rng.normal(size=(n, d)) / np.sqrt(d)).astype(np.float32)
  qcl = rng.integers(0, ncl, size=nq)
  q = (centers[qcl] + 0.35 * rng.normal(size=(nq, d)) / np.sqrt(d)).astype(np.float32)
  layer = ftk.BruteForce(k=k).index(c)
  s, i = layer(q)
  es, ei = o_topk.brute_force(q, c, k)
  np.testing.assert_array_equal(_np(i), ei)
  np.testing.assert_array_equal(_np(s), es)
  # many small clusters (rows still grouped): survivors concentrate in a few segments only
  ncl2 = 6000
  centers2 = rng.normal(size=(ncl2, d)) / np.sqrt(d)
  cl2 = np.sort(rng.integers(0, ncl2, size=n))
  c2 = (centers2[cl2] + 0.6 * rng.normal(size=(n, d)) / np.sqrt(d)).astype(np.float32)
  q2 = (centers2[rng.integers(0, ncl2, size=nq)] + 0.6 * rng.normal(size=(nq, d)) / np.sqrt(d)).astype(np.float32)
  layer2 = ftk.BruteForce(k=k).index(c2)
  s2, i2 = layer2(q2)
  es2, ei2 = o_topk.brute_force(q2, c2, k)
  np.testing.assert_array_equal(_np(i2), ei2)
  np.testing.assert_array_equal(_np(s2), es2)
  assert layer2.last_redo_count() <= nq // 20


@pytest.mark.parametrize("k", [1, 10, 100, 300])
def test_statistical_threshold_plan_is_exact(k, monkeypatch):
  """Shuffled indexes take the filter bound from a sparser sample and a statistically chosen rank
  of the bin maxima (csrc/topk_api.hip plan_sample).  Results must not depend on that choice: the
  guaranteed plan (TFRS_TOPK_STAT=0), the default one, and a deliberately reckless one whose
  bound fails for many queries (they are flagged by the list kernel and redone exactly) all
  equal the oracle bit for bit."""
  ftk = _ftk()
  rng = np.random.default_rng(7 + k)
  n, d, nq = 300_000, 32, 384
  c = (rng.normal(size=(n, d)) / np.sqrt(d)).astype(np.float32)
  q = (rng.normal(size=(nq, d)) / np.sqrt(d)).astype(np.float32)
  es, ei = o_topk.brute_force(q, c, k)
  layer = ftk.BruteForce(k=k).index(c)
  for env in ({"TFRS_TOPK_STAT": "0"}, {}, {"TFRS_TOPK_STAT_PFAIL": "0.4"},
              {"TFRS_TOPK_SAMPLE_STAT": "32", "TFRS_TOPK_STAT_PFAIL": "0.2"}):
    for key in ("TFRS_TOPK_STAT", "TFRS_TOPK_STAT_PFAIL", "TFRS_TOPK_SAMPLE_STAT"):
      monkeypatch.delenv(key, raising=False)
    for key, val in env.items():
      monkeypatch.setenv(key, val)
    s, i = layer(q)
    np.testing.assert_array_equal(_np(i), ei)
    np.testing.assert_array_equal(_np(s), es)
    reasons = layer.last_redo_reasons()
    assert layer.last_redo_count() == reasons["list_overflow"] + reasons["statistical_bound"]   # (large retained
    # sets are re-scored inside the list kernel since round 3: counted, not redone)
    if not env or env == {"TFRS_TOPK_STAT": "0"}:
      assert layer.last_redo_count() == 0, reasons
    if env.get("TFRS_TOPK_STAT_PFAIL") == "0.4" and k >= 10:
      assert reasons["statistical_bound"] > 0, reasons     # the verification is what kept it exact


@pytest.mark.parametrize("d", [16, 32, 64, 100, 128])
def test_fp16_prefiltered_search_equals_all_f32_search_on_whole_batches(d, monkeypatch):
  """The fp16-prefiltered search (threshold pass, filter pass with the LDS survivor queue, exact
  re-scoring) against the all-f32 scan (`TFRS_TOPK_FILTER=f32`: no prefilter, no queue) on every
  query of a 4096 batch and every kernel instantiation (DP = 16 .. 128), bit for bit.  The
  oracle comparisons elsewhere look at tens of queries; a survivor lost in the queue shows up
  in a few percent of the queries only."""
  ftk = _ftk()
  g = torch.Generator(device="cuda").manual_seed(500 + d)
  n, nq, k = 600_000, 4096, 100
  c = torch.randn((n, d), generator=g, device="cuda") / d ** 0.5
  q = torch.randn((nq, d), generator=g, device="cuda") / d ** 0.5
  layer = ftk.BruteForce(k=k).index(c)
  monkeypatch.setenv("TFRS_TOPK_FILTER", "f32")
  s32, i32 = layer(q)
  monkeypatch.delenv("TFRS_TOPK_FILTER")
  for env in ({}, {"TFRS_TOPK_STAT": "0"}, {"TFRS_SCAN16_DRAIN_EVERY": "1"}, {"TFRS_SCAN16_DRAIN": "1"}):
    for key, val in env.items():
      monkeypatch.setenv(key, val)
    s, i = layer(q)
    assert layer.last_redo_count() == 0
    assert torch.equal(i, i32) and torch.equal(s, s32), env
    for key in env:
      monkeypatch.delenv(key)


def test_small_cluster_blocks_through_index_from_dataset():
  """`index_from_dataset(total_rows=...)` fed with small batches, one cluster per batch (the row
  order a catalogue sorted by category gives).  The index mixes rows within an appended block
  only, so the batches are gathered into large chunks first: results exact and (nearly) no query
  on the exact-redo path."""
  ftk = _ftk()
  rng = np.random.default_rng(321)
  ncl, per, d, nq, k = 400, 1000, 64, 512, 100
  centers = rng.normal(size=(ncl, d)) / np.sqrt(d)
  blocks = [(centers[c] + 0.35 * rng.normal(size=(per, d)) / np.sqrt(d)).astype(np.float32)
            for c in range(ncl)]
  q = (centers[rng.integers(0, ncl, size=nq)] + 0.35 * rng.normal(size=(nq, d)) / np.sqrt(d)).astype(np.float32)
  layer = ftk.BruteForce(k=k).index_from_dataset([torch.from_numpy(b).cuda() for b in blocks], total_rows=ncl * per)
  s, i = layer(q)
  es, ei = o_topk.brute_force(q, np.concatenate(blocks), k)
  np.testing.assert_array_equal(_np(i), ei)
  np.testing.assert_array_equal(_np(s), es)
  assert layer.last_redo_count() <= nq // 50, layer.last_redo_reasons()


@pytest.mark.parametrize("batch,din,dout", [(131072, 512, 1), (65536, 13, 512), (131072, 256, 32),
                                            (20000, 200, 3), (8191, 129, 130)])
def test_skinny_dense_backward_split_k(batch, din, dout, monkeypatch):
  """Weight gradients of layers with a small input or output width (the first / last layers of the
  ranking models' MLPs at BASELINE batch sizes): x^T dy is a few output tiles with K = batch and is
  cut into K slices (csrc/interaction.hip launch_gemm).  dW, dx, db against float64."""
  from recommenders_amd.layers.feature_interaction import dcn
  monkeypatch.delenv("TFRS_GEMM_MODE", raising=False)
  g = torch.Generator(device="cuda").manual_seed(batch + din)
  x = torch.randn((batch, din), generator=g, device="cuda")
  w = torch.randn((din, dout), generator=g, device="cuda") / din ** 0.5
  dy = torch.randn((batch, dout), generator=g, device="cuda")
  dx, dw, db = dcn.dense_backward(x, w, dy, True, True, True)
  x64, w64, dy64 = x.double(), w.double(), dy.double()
  ref_dw = x64.t() @ dy64
  ref_dx = dy64 @ w64.t()
  ref_db = dy64.sum(0)
  scale_w = float((x64.abs().t() @ dy64.abs()).max())          # the sum of |terms| bounds the rounding
  assert float((dw.double() - ref_dw).abs().max()) <= 4e-6 * scale_w
  assert float((db.double() - ref_db).abs().max()) <= 4e-6 * float(dy64.abs().sum(0).max())
  scale_x = float((dy64.abs() @ w64.abs().t()).max())
  assert float((dx.double() - ref_dx).abs().max()) <= 4e-6 * scale_x


@pytest.mark.parametrize("self_interaction", [False, True])
def test_dot_interaction_forward_concat_equals_cat(self_interaction):
  """`DotInteraction.forward_concat` (pairs written into / read from the wider concat matrix by the
  strided kernels) against `torch.cat([prefix, layer(inputs)])`: values bit for bit, gradients of
  every input and of the prefix (which is also the last input, as in the ranking model)."""
  from recommenders_amd.layers.feature_interaction import DotInteraction
  g = torch.Generator(device="cuda").manual_seed(77)
  b, f, d = 2048, 27, 32
  layer = DotInteraction(self_interaction=self_interaction)
  base = [torch.randn((b, d), generator=g, device="cuda") for _ in range(f)]
  outs = []
  for fused in (False, True):
    xs = [t.clone().requires_grad_(True) for t in base]
    if fused:
      y = layer.forward_concat(xs, xs[-1])
    else:
      y = torch.cat([xs[-1], layer(xs)], dim=1)
    w = torch.linspace(-1.0, 1.0, y.shape[1], device="cuda")
    (y * w).sum().backward()
    outs.append((y.detach(), [t.grad.clone() for t in xs]))
  assert torch.equal(outs[0][0], outs[1][0])
  for ga, gb in zip(outs[0][1], outs[1][1]):
    assert torch.allclose(ga, gb, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("f,self_interaction", [(2, False), (3, True), (17, True), (32, False), (33, True), (64, True),
                                                (65, False), (96, True), (97, False), (101, True), (112, False),
                                                (113, True), (127, True), (128, False)])
def test_dot_producer_consumer_kernels_shapes(f, self_interaction):
  """`dot_interaction_fwd_pc_kernel` and `dot_interaction_bwd_h16_kernel` (the defaults at D = 32, batch >= 512) at every
  instantiation and at the edges of their tilings -- feature counts around the 16-wide k steps and the
  32-row blocks, the last chunk of a packed row cut at 1, 2 and 3 elements, a batch that is neither a
  multiple of the grid nor of the register-set rotation -- against the float64 oracle on EVERY sample
  (`layers/feature_interaction/dot_interaction.py:53-104` differentiated).  Samples span eight
  decades of gradient and input magnitude (the power-of-two scale is per sample), one sample has an
  all-zero gradient and one an all-zero input."""
  from oracle import feature_interaction as o_fi
  from recommenders_amd.layers.feature_interaction import DotInteraction
  g = torch.Generator(device="cuda").manual_seed(100 * f + int(self_interaction))
  b, d = 777, 32
  x = torch.randn((b, f, d), generator=g, device="cuda")
  x *= torch.exp(4.0 * torch.randn((b, 1, 1), generator=g, device="cuda"))
  x[5] = 0.0
  x.requires_grad_(True)
  pairs = f * (f + 1) // 2 if self_interaction else f * (f - 1) // 2
  dy = torch.randn((b, pairs), generator=g, device="cuda")
  dy *= torch.exp(4.0 * torch.randn((b, 1), generator=g, device="cuda"))
  dy[3] = 0.0
  out = DotInteraction(self_interaction=self_interaction).forward_stacked(x)
  out.backward(dy)
  feats = [_np(x.detach()[:, j, :]) for j in range(f)]
  dref = o_fi.dot_interaction_grad(feats, _np(dy), self_interaction, False)
  yf, yb = o_fi.dot_interaction_yardsticks(feats, _np(dy), self_interaction, False)
  assert torch.isfinite(out).all() and float(out[5].abs().max()) == 0.0
  float_gate(f"dot_pc.f{f}.fwd", _np(out.detach()), o_fi.dot_interaction(feats, self_interaction, False), yf,
             GATE_DOT_C4["fwd"])
  assert torch.isfinite(x.grad).all()
  assert float(x.grad[3].abs().max()) == 0.0 and float(x.grad[5].abs().max()) == 0.0
  float_gate(f"dot_h16.f{f}.bwd", _np(x.grad), dref, yb, GATE_DOT_C4["bwd"])


@pytest.mark.parametrize("b", [512, 513, 600, 767, 1024, 1025, 1290, 1537])
def test_dot_producer_consumer_kernels_pipeline_tails(b):
  """The software pipelines of the two default DotInteraction kernels rotate four register sets and two
  LDS buffers by name; a workgroup that owns 2, 3, 4, 5, 6 or 7 samples leaves the unrolled loop through
  a different remainder path each time (256 workgroups for the backward, 512 for the forward).  Every
  sample of each batch size against the float64 oracle, F = 101, D = 32, both triangle variants."""
  from oracle import feature_interaction as o_fi
  from recommenders_amd.layers.feature_interaction import DotInteraction
  f, d = 101, 32
  for self_interaction in (False, True):
    g = torch.Generator(device="cuda").manual_seed(b + int(self_interaction))
    x = torch.randn((b, f, d), generator=g, device="cuda")
    x *= torch.exp(torch.randn((b, 1, 1), generator=g, device="cuda"))
    x.requires_grad_(True)
    pairs = f * (f + 1) // 2 if self_interaction else f * (f - 1) // 2
    dy = torch.randn((b, pairs), generator=g, device="cuda")
    out = DotInteraction(self_interaction=self_interaction).forward_stacked(x)
    out.backward(dy)
    feats = [_np(x.detach()[:, j, :]) for j in range(f)]
    yf, yb = o_fi.dot_interaction_yardsticks(feats, _np(dy), self_interaction, False)
    float_gate("dot_tails.fwd", _np(out.detach()), o_fi.dot_interaction(feats, self_interaction, False), yf,
               GATE_DOT_C4["fwd"])
    float_gate("dot_tails.bwd", _np(x.grad), o_fi.dot_interaction_grad(feats, _np(dy), self_interaction, False), yb,
               GATE_DOT_C4["bwd"])


@pytest.mark.parametrize("f", [122, 123, 128])
def test_dot_interaction_forward_concat_envelope_edge(f):
  """F = 123..128 at D = 32 are inside the strided forward's envelope but outside the strided
  backward's: `forward_concat` must fall back to the contiguous kernels + cat for them (it used
  to crash in the first backward) and agree with the float64 oracle either way."""
  from oracle import feature_interaction as o_fi
  from recommenders_amd.layers.feature_interaction import DotInteraction
  g = torch.Generator(device="cuda").manual_seed(f)
  b, d = 1024, 32
  layer = DotInteraction()
  xs = [torch.randn((b, d), generator=g, device="cuda").requires_grad_(True) for _ in range(f)]
  y = layer.forward_concat(xs, xs[-1])
  dy = torch.randn(y.shape, generator=g, device="cuda")
  y.backward(dy)
  rows = np.r_[0:8, b - 8:b]
  feats = [_np(t.detach())[rows] for t in xs]
  ref = o_fi.dot_interaction(feats)
  yf, yb = o_fi.dot_interaction_yardsticks(feats, _np(dy)[rows, d:])
  float_gate("dot_edge.fwd", _np(y.detach())[rows, d:], ref, yf, GATE_DOT_C4["fwd"])
  dref = o_fi.dot_interaction_grad(feats, _np(dy)[rows, d:])
  dref[:, -1, :] += _np(dy)[rows, :d]                   # the prefix is also the last input
  got = np.stack([_np(t.grad)[rows] for t in xs], axis=1)
  float_gate("dot_edge.bwd", got, dref, yb + np.abs(_np(dy)[rows, None, :d]) * (np.arange(f) == f - 1)[None, :, None],
             GATE_DOT_C4["bwd"])


def test_ranking_dlrm_fast_path_equals_generic_path():
  """`experimental.models.Ranking` with `EmbeddingDict` + `DotInteraction`: ids given as `[B]`
  vectors take the fast path (one gather into the `[B, F + 1, D]` block incl. the invalid-id slot,
  strided DotInteraction + concat), ids given as `[B, 1]` the generic one.  Same predictions and,
  after one Adagrad step from identical weights, the same parameters."""
  import copy
  import recommenders_amd as tfrs
  from recommenders_amd.experimental.models import ranking as rk
  g = torch.Generator(device="cuda").manual_seed(5)
  B, D, nd = 1024, 32, 13
  vocab = {"a": 300, "b": 50, "c": 1000, "d": 7}

  def make():
    torch.manual_seed(123)
    emb = rk.EmbeddingDict(vocab, D)
    model = rk.Ranking(emb, bottom_stack=tfrs.layers.blocks.MLP(units=[64, D], final_activation="relu"),
                       feature_interaction=tfrs.layers.feature_interaction.DotInteraction(),
                       top_stack=tfrs.layers.blocks.MLP(units=[64, 1], final_activation="sigmoid"))
    return model

  dense = torch.rand((B, nd), generator=g, device="cuda")
  ids = {k: torch.randint(0, v, (B,), generator=g, device="cuda") for k, v in vocab.items()}
  labels = torch.randint(0, 2, (B,), generator=g, device="cuda")
  fast, generic = make(), make()
  f_in = {"dense_features": dense, "sparse_features": ids}
  g_in = {"dense_features": dense, "sparse_features": {k: v[:, None] for k, v in ids.items()}}
  p_fast = fast(f_in)
  generic(g_in)
  generic.load_state_dict(copy.deepcopy(fast.state_dict()))
  p_gen = generic(g_in)
  assert torch.allclose(p_fast, p_gen, rtol=1e-6, atol=1e-7)
  for m, batch in ((fast, f_in), (generic, g_in)):
    m.compile(optimizer=tfrs.optimizers.Adagrad(m.parameters(), learning_rate=0.1))
    m.train_step((batch, labels))
  for (n1, a), (n2, b) in zip(fast.named_parameters(), generic.named_parameters()):
    assert n1 == n2
    assert torch.allclose(a, b, rtol=2e-5, atol=2e-6), n1


def test_survivor_workspace_covers_every_split_count():
  """Found by tools/fuzz_topk.py: the survivor buffer was sized for the LARGEST segment count, but
  `segments x truncated capacity` is not monotone in the segment count (254 x 57 > 256 x 56), so
  K = 400 with 2048 queries on ~2530 stages failed with "survivor workspace too small"."""
  ftk = _ftk()
  g = torch.Generator(device="cuda").manual_seed(4)
  n, d, nq, k = 323_800, 16, 2048, 400
  c = torch.randn((n, d), generator=g, device="cuda") / d ** 0.5
  q = torch.randn((nq, d), generator=g, device="cuda") / d ** 0.5
  layer = ftk.BruteForce(k=k).index(c)
  s, i = layer(q)
  es, ei = o_topk.brute_force(_np(q[:8]), _np(c), k)
  np.testing.assert_array_equal(_np(i[:8]), ei)
  np.testing.assert_array_equal(_np(s[:8]), es)


# ------------------------------------------------------------------------------------------------
# DotInteraction at BASELINE configs[4] (B = 131072, F = 101, D = 32): the persistent multi-sample
# loops -- `dot_interaction_f16x3_direct_kernel`'s wave-stride loop and the double-buffered steady
# state of `dot_interaction_bwd_pc_kernel` (one 8-wave workgroup per CU, 512 samples each) --
# against the float64 oracle (layers/feature_interaction/dot_interaction.py:53-104, output order of
# dot_interaction_test.py:25-64) on 384 samples spread over the first / middle / last iterations of
# a workgroup and over every residue class of b mod 256.
# ------------------------------------------------------------------------------------------------
def _dot_c4_samples(b):
  rng = np.random.default_rng(11)
  spread = np.arange(256) * (b // 256) + (np.arange(256) * 37) % 256      # every b mod 256 class once
  edge = np.r_[0:32, b // 2 - 16:b // 2 + 16, b - 32:b]
  extra = rng.integers(0, b, size=32)
  return np.unique(np.clip(np.r_[spread, edge, extra], 0, b - 1))


@pytest.mark.parametrize("variant", ["default", "strided", "pc_bwd", "dense_bwd", "direct_fwd", "staged_fwd", "f32_fwd"])
@pytest.mark.parametrize("self_interaction", [False, True])
def test_dot_interaction_config5_vs_float64(self_interaction, variant, monkeypatch):
  from oracle import feature_interaction as o_fi
  from recommenders_amd.layers.feature_interaction import DotInteraction
  b, f, d = 131072, 101, 32
  if variant == "pc_bwd":
    monkeypatch.setenv("TFRS_DOT_BWD", "pc")          # the f32 producer / consumer backward kernel (round 2)
  elif variant == "dense_bwd":
    monkeypatch.setenv("TFRS_DOT_BWD", "d")           # the single-role dense-S backward kernel
  elif variant == "direct_fwd":
    monkeypatch.setenv("TFRS_DOT_FWD", "direct")      # split-fp16 forward, every wave loads and stores (round 2)
  elif variant == "staged_fwd":
    monkeypatch.setenv("TFRS_DOT_FWD", "staged")      # LDS-staged split-fp16 forward
  elif variant == "f32_fwd":
    monkeypatch.setenv("TFRS_DOT_FWD", "f32")         # exact-f32 MFMA forward
  g = torch.Generator(device="cuda").manual_seed(2025 + int(self_interaction))
  # per-sample magnitudes spread over two decades so that a sample served from a stale buffer
  # (another sample's S / X tile) cannot pass by accident
  x = torch.randn((b, f, d), generator=g, device="cuda")
  x *= torch.exp(torch.randn((b, 1, 1), generator=g, device="cuda"))
  x.requires_grad_(True)
  layer = DotInteraction(self_interaction=self_interaction)
  pairs = f * (f + 1) // 2 if self_interaction else f * (f - 1) // 2
  if variant == "strided":
    prefix = torch.randn((b, d), generator=g, device="cuda")
    out_full = layer.forward_stacked(x, prefix)        # pairs written into the wider [B, D + pairs] rows
    assert out_full.shape == (b, d + pairs)
    assert torch.equal(out_full[:, :d], prefix)
    dy_full = torch.randn((b, d + pairs), generator=g, device="cuda")
    out_full.backward(dy_full)
    out, dy = out_full[:, d:], dy_full[:, d:]
  else:
    out = layer.forward_stacked(x)
    dy = torch.randn((b, pairs), generator=g, device="cuda")
    out.backward(dy)
  rows = _dot_c4_samples(b)
  rt = torch.as_tensor(rows, device="cuda")
  xs = _np(x.detach()[rt])                                              # [n, f, d]
  feats = [xs[:, j, :] for j in range(f)]
  dys = _np(dy[rt])
  ref = o_fi.dot_interaction(feats, self_interaction, False)
  dref = o_fi.dot_interaction_grad(feats, dys, self_interaction, False)
  yf, yb = o_fi.dot_interaction_yardsticks(feats, dys, self_interaction, False)
  float_gate(f"dot_c4.{variant}.fwd", _np(out.detach()[rt]), ref, yf, GATE_DOT_C4["fwd"])
  float_gate(f"dot_c4.{variant}.bwd", _np(x.grad[rt]), dref, yb, GATE_DOT_C4["bwd"])
  # size-independent property on EVERY sample: sum_j dX[b, j, :] * X[b, j, :] = 2 * <dy, out> (Euler:
  # the packed pairs are homogeneous of degree 2 in X)
  lhs = (x.grad.double() * x.detach().double()).sum(dim=(1, 2))
  rhs = 2.0 * (dy.double() * out.detach().double()).sum(dim=1)
  yard = (x.grad.double().abs() * x.detach().double().abs()).sum(dim=(1, 2)) + 1e-30
  assert float(((lhs - rhs).abs() / yard).max()) < 1e-4


# ------------------------------------------------------------------------------------------------
# Embedding path at the TABLE sizes of BASELINE configs[3] / configs[4]:
# byte offsets beyond 2^32, the last row, int32 and int64 ids.  The table is synthetic with a closed
# form -- value(r, j) is an integer hash of (r, j) scaled by a power of two, exact in float32 -- so
# the host regenerates any row without ever holding the 13 GB table.
# ------------------------------------------------------------------------------------------------
def _synthetic_rows_host(rows, d, salt=0):
  r = np.asarray(rows, dtype=np.int64)[:, None]
  j = np.arange(d, dtype=np.int64)[None, :]
  h = (r * 2654435761 + j * 40503 + 12345 + salt * 7919) & 0x3FFFFF          # 22 bits: exact in f32
  return (h.astype(np.float32) - np.float32(2097152.0)) * np.float32(2.0 ** -26)   # [-1/32, 1/32)


def _synthetic_table_device(rows, d, salt=0, chunk=1 << 20):
  table = torch.empty((rows, d), dtype=torch.float32, device="cuda")
  j = torch.arange(d, dtype=torch.int64, device="cuda")[None, :] * 40503 + 12345 + salt * 7919
  for lo in range(0, rows, chunk):
    hi = min(rows, lo + chunk)
    r = torch.arange(lo, hi, dtype=torch.int64, device="cuda")[:, None] * 2654435761
    h = (r + j) & 0x3FFFFF
    table[lo:hi] = (h.to(torch.float32) - 2097152.0) * (2.0 ** -26)
  return table


def _compact(ids, d, salt=0):
  """(sub-table of the distinct ids, ids remapped into it): the oracle then works on a table of
  a few thousand rows; the ORDER of the ids (hence of every float32 sum) is unchanged."""
  uniq, inv = np.unique(np.asarray(ids, dtype=np.int64), return_inverse=True)
  return _synthetic_rows_host(uniq, d, salt), inv.astype(np.int64), uniq


@pytest.mark.parametrize("rows,d,n", [(26_000_000, 128, 65536 * 26), (100_000_000, 32, 131072 * 13)])
def test_embedding_path_at_config_table_sizes(rows, d, n):
  """configs[3]: 26 x 1M-row tables of dim 128 held as one 26M x 128 store (13.3 GB); configs[4]: one
  GPU's 100M-row share of the dim-32 tables (12.8 GB).  Gather (what bench.py's `gather` leg times),
  weighted segment-sum with the three combiners and the fused sparse Adagrad, each on ids that
  include row 0, the last row and rows on both sides of the 2^32-byte offset, against the oracle
  (`oracle/embedding.py`; reference layers/embedding/tpu_embedding_layer.py:913-919) on sampled /
  touched rows: gather and the Adagrad-untouched rows bit for bit."""
  from recommenders_amd.layers import embedding as emb
  from oracle import embedding as o_emb
  rng = np.random.default_rng(rows % 1000 + d)
  table = _synthetic_table_device(rows, d)
  edge = (1 << 32) // (d * 4)                     # first row whose byte offset is >= 2^32
  special = np.array([0, 1, rows - 1, rows - 2, edge - 1, edge, edge + 1, 2 * edge, rows // 2], dtype=np.int64)
  assert special.max() < rows and rows * d * 4 > (1 << 33)

  # ---- gather: the bench's shape, int64 and int32 ids
  ids = rng.integers(0, rows, size=n)
  ids[:special.size] = special
  ids[-special.size:] = special[::-1]
  sample = np.unique(np.r_[0:special.size, n - special.size:n, rng.integers(0, n, size=4096)])
  sub, remap, _ = _compact(ids[sample], d)
  want = o_emb.gather(sub, remap)
  for dtype in (np.int64, np.int32):
    out = emb.gather_rows(table, torch.as_tensor(ids.astype(dtype)).cuda())
    np.testing.assert_array_equal(_np(out[torch.as_tensor(sample).cuda()]), want)
    # ... and every output row against the table through an independent route (torch indexing)
    assert torch.equal(out, table[torch.as_tensor(ids).cuda()])
    del out

  # ---- weighted segment-sum: ragged rows of 0..8 ids
  nseg = 200_000
  lens = rng.integers(0, 9, size=nseg)
  splits = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
  sids = rng.integers(0, rows, size=int(splits[-1]))
  sids[:special.size] = special
  wts = rng.uniform(0.5, 2.0, size=sids.size).astype(np.float32)
  seg_sample = np.unique(np.r_[0:8, rng.integers(0, nseg, size=4096)])
  # the sampled segments as their own small CSR problem for the oracle
  pieces = [np.arange(splits[b], splits[b + 1]) for b in seg_sample]
  take = np.concatenate(pieces) if pieces else np.zeros((0,), np.int64)
  sub, remap, _ = _compact(sids[take], d)
  sub_splits = np.concatenate([[0], np.cumsum([p.size for p in pieces])]).astype(np.int64)
  for comb in ("sum", "mean", "sqrtn"):
    for w in (None, wts):
      out = emb.embedding_lookup_sparse(table, torch.as_tensor(sids).cuda(), torch.as_tensor(splits).cuda(),
                                        None if w is None else torch.as_tensor(w).cuda(), combiner=comb)
      ref = o_emb.lookup_sparse(sub, remap, sub_splits, None if w is None else w[take], comb)
      np.testing.assert_allclose(_np(out[torch.as_tensor(seg_sample).cuda()]), ref, rtol=2e-6, atol=2e-7)   # fma contraction only
      del out

  # ---- fused sparse Adagrad on the full-size table + accumulator (a guard row behind both)
  del table
  torch.cuda.empty_cache()
  backing = torch.empty((rows + 1, d), dtype=torch.float32, device="cuda")
  backing[:rows] = _synthetic_table_device(rows, d)
  backing[rows] = 0.0
  acc_backing = torch.full((rows + 1, d), 0.1, dtype=torch.float32, device="cuda")
  tab, accum = backing[:rows], acc_backing[:rows]
  m = 300_000
  # half of the gradient rows go to ~20 000 hot ids: runs of ~8 duplicates, all shorter than the kernel's
  # `piece` (32 positions), so every sum is ONE thread's occurrence-order chain = the oracle's, bit for bit
  hot = np.concatenate([special, rng.integers(0, rows, size=20_000)])
  gids = np.where(rng.random(m) < 0.5, hot[rng.integers(0, hot.size, size=m)], rng.integers(0, rows, size=m))
  gids[:special.size] = special
  g = rng.normal(size=(m, d)).astype(np.float32)
  untouched = np.setdiff1d(np.r_[rng.integers(0, rows, size=4096), (special + 3) % rows], gids)
  for dtype in (np.int64, np.int32):
    emb.adagrad_sparse_update_(tab, accum, torch.as_tensor(g).cuda(), torch.as_tensor(gids.astype(dtype)).cuda(),
                               lr=0.5)
  sub, remap, uniq = _compact(gids, d)
  t_ref, a_ref = sub, np.full_like(sub, 0.1)
  for _ in range(2):                                                   # the two updates above, in order
    t_ref, a_ref = o_emb.adagrad_sparse_update(t_ref, a_ref, g, remap, lr=0.5)
  touched = torch.as_tensor(uniq).cuda()
  np.testing.assert_allclose(_np(accum[touched]), a_ref, rtol=1e-6)
  np.testing.assert_allclose(_np(tab[touched]), t_ref, rtol=1e-5, atol=1e-7)
  np.testing.assert_array_equal(_np(tab[torch.as_tensor(untouched).cuda()]), _synthetic_rows_host(untouched, d))
  assert float((accum[torch.as_tensor(untouched).cuda()] - 0.1).abs().max()) == 0.0
  assert float(backing[rows].abs().max()) == 0.0 and float((acc_backing[rows] - 0.1).abs().max()) == 0.0
  # the scatter-add alone (dense gradient of the gather), bit-exact occurrence-order sums on the touched rows
  del backing, acc_backing, tab, accum
  torch.cuda.empty_cache()
  dense = emb.scatter_add_rows(torch.as_tensor(g).cuda(), torch.as_tensor(gids).cuda(), rows)
  ref = o_emb.scatter_add_grad(g, remap, uniq.size)
  np.testing.assert_array_equal(_np(dense[touched]), ref)
  assert float(dense[torch.as_tensor(untouched).cuda()].abs().max()) == 0.0
  # very hot rows (runs of 4000 >> piece: summed as partial sums of pieces, csrc/embedding.hip): float64
  # reference, gate relative to the sum of |terms|
  del dense
  vh = np.array([rows - 1, edge, 5], dtype=np.int64)
  vids = np.repeat(vh[None, :], 4000, axis=0).reshape(-1)
  vg = rng.normal(size=(vids.size, d)).astype(np.float32)
  dense = emb.scatter_add_rows(torch.as_tensor(vg).cuda(), torch.as_tensor(vids).cuda(), rows)
  got = _np(dense[torch.as_tensor(vh).cuda()])
  for j in range(3):
    terms = vg[j::3].astype(np.float64)
    float_gate("scatter.long_runs", got[j], terms.sum(axis=0), np.abs(terms).sum(axis=0), 4e-6)


def test_scale_workload_100m_x_64_search_vs_oracle():
  """The 100M x 64 single-GPU search that bench.py's `scale_workload` times (the N = 1 point of the
  strong-scaling configuration): same row generator (1M-row blocks from seeds 1000 + b), batch 8192,
  top-100; 16 sampled queries against the oracle's Streaming fold (layers/factorized_top_k.py:404-509,
  which equals BruteForce.call :586-607 on the concatenation) over the same blocks downloaded to the
  host, bit for bit; no query may need the exact redo."""
  ftk = _ftk()
  n, d, nq, k, blk = 100_000_000, 64, 8192, 100, 1_000_000

  def block(b):
    g = torch.Generator(device="cuda").manual_seed(1000 + b)
    return torch.randn((blk, d), generator=g, device="cuda", dtype=torch.float32) / (d ** 0.5)

  class Blocks:
    def __iter__(self):
      for b in range(n // blk):
        yield block(b)

  g = torch.Generator(device="cuda").manual_seed(7)
  q = torch.randn((nq, d), generator=g, device="cuda", dtype=torch.float32) / (d ** 0.5)
  bf = ftk.BruteForce(k=k).index_from_dataset(Blocks(), total_rows=n)
  s, i = bf(q)
  assert bf.last_redo_count() == 0, bf.last_redo_reasons()
  sample = np.r_[0:6, 4000:4004, nq - 6:nq]
  qs = q[sample].cpu().numpy()
  es = np.zeros((sample.size, 0), np.float32)
  ei = np.zeros((sample.size, 0), np.int64)
  for b in range(n // blk):
    # Streaming.call's fold: per-block top-k with global row numbers, concat([state, new]) -> top_k
    # (ties -> left-most column = lower global row), reference :424-472
    bs, bi = o_topk.brute_force(qs, block(b).cpu().numpy(), k)
    joined_s, joined_i = np.concatenate([es, bs], axis=1), np.concatenate([ei, bi.astype(np.int64) + b * blk], axis=1)
    es, order = o_topk.top_k(joined_s, k)
    ei = o_topk.take_along_axis(joined_i, order)
  np.testing.assert_array_equal(_np(i)[sample], ei)
  np.testing.assert_array_equal(_np(s)[sample], es)
  # size-independent properties on the whole batch: sorted (score desc, row asc), rows distinct and in range
  sv, iv = _np(s), _np(i).astype(np.int64)
  assert (sv[:, :-1] >= sv[:, 1:]).all()
  tie = sv[:, :-1] == sv[:, 1:]
  assert (iv[:, :-1][tie] < iv[:, 1:][tie]).all()
  assert iv.min() >= 0 and iv.max() < n
  assert (np.sort(iv, axis=1)[:, 1:] != np.sort(iv, axis=1)[:, :-1]).all()


# experimental.models.Ranking composed at the BASELINE configs[3] / configs[4] shapes (missing 3b):
# at these sizes the paths that run are the fast ones -- one gather into the [B, F + 1, D] block through
# EmbeddingDict's row ranges, the 256 x 256-tile split-fp16 GEMMs of the Cross stack and the MLPs, the strided
# DotInteraction into the concat buffer, sparse (ids, rows) gradient slices -- none of which the B = 256, D = 16
# model test exercises.
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("config", ["c3_dcn_v2", "c4_dlrm_shard"])
def test_ranking_model_at_config_shapes_vs_oracle(config):
  """`experimental/models/ranking.py:203-236` forward + one backward + Adagrad step at
  configs[3] (DCN-v2: B = 65536, 26 x 1M-row tables of dim 128 + 13 dense features through the bottom stack,
  3 Cross layers on the 27 * 128 = 3456-wide concatenation) and at one GPU's share of configs[4] (DLRM:
  B = 131072, 100 tables of dim 32 -- 1.25M rows each, the 1/8 row shard -- DotInteraction over 101 vectors)
  against `oracle/ranking.py` on 256 sampled examples: predictions, the mean loss's gradient wrt every sampled
  example's embedding vectors (read from the (ids, rows) slice the lookup's backward hands the optimizer), the
  loss over the batch (float64 on the GPU's own predictions), and the fused Adagrad update of the sampled rows
  (touched once) from that gradient; a never-looked-up guard row stays bit for bit."""
  import recommenders_amd as tfrs
  from recommenders_amd.experimental.models import ranking as rk
  from oracle import ranking as o_rank
  if config == "c3_dcn_v2":
    n_tables, vocab, dim, batch, lr = 26, 1_000_000, 128, 65536, 0.5
    bottom_units, fi, interaction = [512, 256, dim], rk.ConcatCross(num_layers=3), "cross"
  else:
    n_tables, vocab, dim, batch, lr = 100, 1_250_000, 32, 131072, 0.5
    bottom_units, fi, interaction = [512, 256, dim], tfrs.layers.feature_interaction.DotInteraction(), "dot"
  g = torch.Generator(device="cuda").manual_seed(2024)
  emb = rk.EmbeddingDict({str(i): vocab for i in range(n_tables)}, dim)
  bottom = tfrs.layers.blocks.MLP(units=bottom_units, final_activation="relu")
  top = tfrs.layers.blocks.MLP(units=[1024, 512, 1], final_activation="sigmoid")
  model = rk.Ranking(emb, bottom_stack=bottom, feature_interaction=fi, top_stack=top,
                     task=tfrs.tasks.Ranking(loss=tfrs.losses.BinaryCrossentropy(reduction="none")))
  dense = torch.rand((batch, 13), generator=g, device="cuda")
  # the last row of every table is never looked up (the guard), sampled examples get ids nobody else has
  ids = {str(i): torch.randint(0, vocab - 1 - 256, (batch,), generator=g, device="cuda") for i in range(n_tables)}
  sample = torch.tensor(np.r_[0:96, batch // 2:batch // 2 + 64, batch - 96:batch], device="cuda")
  ns = sample.numel()
  for i in range(n_tables):
    ids[str(i)][sample] = vocab - 1 - 256 + torch.arange(ns, device="cuda")
  labels = torch.randint(0, 2, (batch,), generator=g, device="cuda")
  feats = {"dense_features": dense, "sparse_features": ids}
  with torch.no_grad():
    model(feats)                                                   # builds the lazily shaped layers
    if interaction == "cross":                                     # non-zero biases in the cross stack
      for layer in fi.layers:
        layer.bias.uniform_(-0.05, 0.05, generator=g)
    for m in list(bottom._sublayers) + list(top._sublayers):
      m.bias.uniform_(-0.05, 0.05, generator=g)
  opt = tfrs.optimizers.Adagrad(model.parameters(), learning_rate=lr)
  model.compile(optimizer=opt)
  model.train()
  opt.zero_grad(set_to_none=True)
  loss = model.compute_loss((feats, labels), training=True)
  with torch.no_grad():
    pred = model(feats)
  loss.backward()
  table = emb.embeddings
  slices = table._tfrs_slices
  assert len(slices) == 1                                          # one (ids, rows) slice for all features
  rows_idx, rows_grad = slices[0][0].reshape(batch, -1), slices[0][1].reshape(batch, -1, dim)
  assert rows_idx.shape[1] in (n_tables, n_tables + 1)
  starts = torch.arange(n_tables, device="cuda") * vocab
  want_rows = torch.stack([ids[k] for k in sorted(ids, key=str)], dim=1) + starts[
      torch.tensor([int(k) for k in sorted(ids, key=str)], device="cuda")]
  assert torch.equal(rows_idx[:, :n_tables].long(), want_rows)
  # ---- the oracle on the sampled examples
  order = sorted(ids, key=str)
  samp_rows = want_rows[sample]                                    # [ns, F] global rows
  before = table.detach()[samp_rows.reshape(-1)].reshape(ns, n_tables, dim).clone()
  embs = [_np(before[:, j, :]) for j in range(n_tables)]
  bt = ([_np(l.kernel.detach()) for l in bottom._sublayers], [_np(l.bias.detach()) for l in bottom._sublayers],
        "relu", "relu")
  tp = ([_np(l.kernel.detach()) for l in top._sublayers], [_np(l.bias.detach()) for l in top._sublayers],
        "relu", "sigmoid")
  ck = [_np(l.kernel.detach()) for l in fi.layers] if interaction == "cross" else None
  cb = [_np(l.bias.detach()) for l in fi.layers] if interaction == "cross" else None
  p_ref, dx_ref, _ = o_rank.ranking_model_embedding_grads(
      _np(dense[sample]), embs, _np(labels[sample]), bt, tp, interaction, batch, True, ck, cb)
  # predictions: probabilities, error relative to 1 (north_star: 1e-5)
  float_gate("ranking_%s.pred" % config, _np(pred[sample]), p_ref, np.ones_like(p_ref), 1e-6)       # observed 6e-8 (round 5)
  # gradient rows: relative to the largest entry of the example's gradient block
  got = _np(rows_grad[sample][:, :n_tables, :]).astype(np.float64)
  scale = np.abs(dx_ref).max(axis=(1, 2), keepdims=True)
  assert scale.min() > 0
  float_gate("ranking_%s.dembedding" % config, got, dx_ref, np.broadcast_to(scale, dx_ref.shape), 2e-5)   # observed 1.2e-6 / 7.9e-7
  # the loss over the whole batch from the GPU's own predictions (float64; tasks/ranking.py:92-115 + :203-206)
  p64 = pred.double().clamp(1e-7, 1 - 1e-7)
  y64 = labels.double()
  want_loss = float((-(y64 * p64.log() + (1 - y64) * (1 - p64).log())).mean())
  assert abs(float(loss) - want_loss) <= 2e-6 * abs(want_loss), (float(loss), want_loss)
  # ---- the optimizer step: sampled rows (each looked up exactly once) follow Adagrad from the GPU's gradient
  guard = [(j + 1) * vocab - 1 for j in range(n_tables)]
  guard_before = table.detach()[guard].clone()
  opt.step()
  after = table.detach()[samp_rows.reshape(-1)].reshape(ns, n_tables, dim)
  gg = rows_grad[sample][:, :n_tables, :].double()
  acc = 0.1 + gg * gg
  want = before.double() - lr * gg / torch.sqrt(acc + 1e-7)
  step_size = (lr * gg.abs() / torch.sqrt(acc + 1e-7))
  err = (after.double() - want).abs().max()
  assert float(err) <= 2.0 ** -24 * 0.06 * 2 + 1e-6 * float(step_size.max()), (float(err), float(step_size.max()))
  assert float((after - before).abs().max()) > 0
  assert torch.equal(table.detach()[guard], guard_before)
  got_acc = opt.state[table]["accumulator"][samp_rows.reshape(-1)].reshape(ns, n_tables, dim)
  assert torch.allclose(got_acc.double(), acc, rtol=1e-6, atol=0)
