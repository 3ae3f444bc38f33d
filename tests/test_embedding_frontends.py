"""Combiner-rich embedding front-ends: ``TPUEmbedding`` feature/table
configs, ``UnifiedEmbedding`` hashing and ``PartialTPUEmbedding`` placement.

CPU tests pin the oracle's SipHash to the published vectors and cover the host logic; the
``gpu`` tests compare the HIP hashing / combiner-backward kernels and the layers built on them
with the oracle, and restate the reference's own tests
(``layers/embedding/tpu_embedding_layer_test.py``, ``layers/feature_multiplexing/
unified_embedding_test.py``, ``experimental/layers/embedding/partial_tpu_embedding_test.py``).
"""

import math

import numpy as np
import pytest
import torch

import recommenders_amd as tfrs
from oracle import embedding as o_emb
from oracle import hashing as o_hash
from recommenders_amd.layers.embedding import (FeatureConfig, RaggedIds, SparseIds, TableConfig,
                                               TPUEmbedding)
from recommenders_amd.layers.feature_multiplexing import UnifiedEmbedding, UnifiedEmbeddingConfig
from recommenders_amd.layers import tpu_embedding_layer as tel
from tests.conftest import load_golden

gpu = pytest.mark.gpu


def _t(x, dtype=None):
  return torch.as_tensor(np.asarray(x), dtype=dtype).cuda()


def _np(t):
  return t.detach().cpu().numpy()


# ======================================================================== CPU: oracle + host
def test_siphash_known_answers():
  g = load_golden("hashing.json")
  k0, k1 = g["key"]
  for v in g["vectors"]:
    msg = bytes(range(v["len"]))
    assert o_hash.siphash24(k0, k1, msg) == int(v["hash"], 16), v


def test_oracle_hash_bucket_glue():
  # decimal strings for integers, raw bytes for strings, unsigned modulo
  k = (3, 1)
  assert o_hash.hash_bucket_strong([12], 10, k)[0] == o_hash.siphash24(3, 1, b"12") % 10
  assert o_hash.hash_bucket_strong([-7], 97, k)[0] == o_hash.siphash24(3, 1, b"-7") % 97
  assert o_hash.hash_bucket_strong(["drama"], 97, k)[0] == o_hash.siphash24(3, 1, b"drama") % 97
  out = o_hash.hash_bucket_strong(np.arange(12).reshape(3, 4), 5, k)
  assert out.shape == (3, 4) and out.dtype == np.int64 and out.min() >= 0 and out.max() < 5
  with pytest.raises(ValueError):
    o_hash.hash_bucket_strong([1], 0, k)


def test_unified_config_round_robin_and_salts():
  """unified_embedding.py:100-126: chunks go to the tables round-robin across features;
  salt = [feature number, chunk id]; chunk names carry config and feature names."""
  cfg = UnifiedEmbeddingConfig(buckets_per_table=100, dim_per_table=16, num_tables=3,
                               name="unified_table")
  cfg.add_feature("movie_genre", 2)
  cfg.add_feature("movie_id", 3)
  cfg.add_feature("user_zip_code", 1)
  emb_cfg, hash_cfg = cfg.embedding_config, cfg.hashing_config
  assert list(emb_cfg) == ["movie_genre", "movie_id", "user_zip_code"]
  names = [t.name for t in cfg._table_configs]
  assert names == ["unified_table_0", "unified_table_1", "unified_table_2"]
  order = [f.table.name for feat in emb_cfg.values() for f in feat.values()]
  assert order == [names[i % 3] for i in range(6)]
  assert list(emb_cfg["movie_id"]) == [f"unified_table_movie_id_lookup_{c}" for c in range(3)]
  assert hash_cfg["movie_id"]["unified_table_movie_id_lookup_2"] == {"num_bins": 100,
                                                                    "salt": [1, 2]}
  assert hash_cfg["user_zip_code"]["unified_table_user_zip_code_lookup_0"]["salt"] == [2, 0]
  for t in cfg._table_configs:
    assert (t.vocabulary_size, t.dim) == (100, 16)


def test_config_validation():
  with pytest.raises(ValueError):
    TableConfig(vocabulary_size=0, dim=4)
  with pytest.raises(ValueError):
    TableConfig(vocabulary_size=4, dim=4, combiner="max")
  with pytest.raises(ValueError):
    TableConfig(vocabulary_size=4, dim=4, initializer=3)
  t = TableConfig(vocabulary_size=4, dim=2)
  assert t.combiner == "mean"                     # TableConfig default
  with pytest.raises(ValueError):
    FeatureConfig(table="not a table")
  with pytest.raises(ValueError):
    FeatureConfig(table=t, max_sequence_length=-1)
  with pytest.raises(ValueError):
    tfrs.layers.hashing.Hashing(num_bins=0, salt=[0, 0])
  with pytest.raises(NotImplementedError):
    tfrs.layers.hashing.Hashing(num_bins=4)       # unsalted FarmHash mode: off the hot path


def test_nest_helpers_follow_tf_nest_order():
  s = {"b": [1, 2], "a": (3, {"z": 4, "y": 5})}
  assert tel._flatten(s) == [3, 5, 4, 1, 2]
  assert [p for p, _ in tel._flatten_with_paths(s)] == ["a/0", "a/1/y", "a/1/z", "b/0", "b/1"]
  packed = tel._pack_as(s, ["A", "Y", "Z", "B0", "B1"])
  assert packed == {"b": ["B0", "B1"], "a": ("A", {"z": "Z", "y": "Y"})}
  assert list(packed) == ["b", "a"]               # the structure's own key order survives
  assert tel._same_structure(s, packed)
  assert not tel._same_structure(s, {"a": 1, "b": 2})


def test_ragged_and_sparse_containers():
  r = RaggedIds.from_row_lengths([0, 0, 1, 0, 1, 1], [1, 2, 2, 1])
  assert r.nrows == 4 and r.row_splits.tolist() == [0, 1, 3, 5, 6]
  r2 = RaggedIds.from_nested([[3], [], [1, 2]])
  assert r2.row_splits.tolist() == [0, 1, 1, 3] and r2.values.tolist() == [3, 1, 2]
  r3 = RaggedIds.from_nested([["a", "bc"], []])
  assert isinstance(r3.values, np.ndarray) and r3.nrows == 2
  with pytest.raises(ValueError):
    RaggedIds([1, 2, 3], [0, 2])
  s = SparseIds([[0, 0], [1, 0], [1, 1], [3, 2]], [5, 6, 7, 8], (4, 3))
  splits, pos = s._csr()
  assert splits.tolist() == [0, 1, 3, 3, 4] and pos.tolist() == [0, 0, 1, 2]
  with pytest.raises(ValueError):
    SparseIds([[1, 0], [0, 0]], [1, 2], (2, 1))._csr()


def _ref_feature_config(vocab_sizes, dims):
  """partial_tpu_embedding_test.py:25-52."""
  cfg = {}
  for i, (v, d) in enumerate(zip(vocab_sizes, dims)):
    table = TableConfig(vocabulary_size=v, dim=d, combiner="mean", name=f"table_{i}")
    cfg[str(i)] = FeatureConfig(table=table)
  return cfg


def test_partial_embedding_placement_cpu():
  """partial_tpu_embedding_test.py:57-85: tables above the threshold go to TPUEmbedding."""
  cfg = _ref_feature_config([5, 20, 8, 9, 15], [2, 4, 6, 8, 10])
  cpu = torch.device("cpu")
  layer = tfrs.experimental.layers.embedding.PartialTPUEmbedding(cfg, optimizer=None,
                                                                 size_threshold=10, device=cpu)
  tables = layer.tpu_embedding.embedding_tables
  small = layer.keras_embedding_layers
  assert len(tables) == 2 and len(small) == 3
  for tc, w in tables.items():
    assert (tc.vocabulary_size, tc.dim) in ((20, 4), (15, 10))
    assert tuple(w.shape) == (tc.vocabulary_size, tc.dim)
  assert (small["0"].input_dim, small["0"].output_dim) == (5, 2)
  assert (small["2"].input_dim, small["2"].output_dim) == (8, 6)
  assert (small["3"].input_dim, small["3"].output_dim) == (9, 8)
  none = tfrs.experimental.layers.embedding.PartialTPUEmbedding(cfg, None, size_threshold=None,
                                                                device=cpu)
  assert none.tpu_embedding is None and len(none.keras_embedding_layers) == 5
  all_big = tfrs.experimental.layers.embedding.PartialTPUEmbedding(cfg, None, size_threshold=0,
                                                                   device=cpu)
  assert len(all_big.tpu_embedding.embedding_tables) == 5 and not all_big.keras_embedding_layers


def test_tpu_embedding_structure_errors_cpu():
  video = TableConfig(vocabulary_size=2, dim=4, combiner="sum", name="video_table")
  cfg = {"watched": FeatureConfig(table=video), "favorited": FeatureConfig(table=video),
         "seq": FeatureConfig(table=video, max_sequence_length=2)}
  layer = TPUEmbedding(cfg, optimizer=None, device=torch.device("cpu"))
  assert len(layer.embedding_tables) == 1           # shared table, one parameter
  assert cfg["watched"].name == "watched"
  r = RaggedIds.from_row_lengths([0, 1], [1, 1])
  with pytest.raises(ValueError, match="same nested structure"):
    layer({"watched": r})
  feats = {"watched": torch.tensor([0, 1]), "favorited": r, "seq": r}
  with pytest.raises(ValueError, match="input is dense"):
    layer(feats, weights={"watched": torch.ones(2), "favorited": None, "seq": None})
  with pytest.raises(ValueError, match="sequence feature"):
    layer(feats, weights={"watched": None, "favorited": None,
                          "seq": r.with_values(torch.ones(2))})
  with pytest.raises(ValueError, match="dense tensor was passed"):
    layer({"watched": r, "favorited": r, "seq": torch.tensor([0, 1])})
  with pytest.raises(ValueError, match="does not match"):
    layer(feats, weights={"watched": None,
                          "favorited": SparseIds([[0, 0], [1, 0]], [1.0, 1.0], (2, 1)),
                          "seq": None})
  with pytest.raises(ValueError, match="FeatureConfig expected"):
    TPUEmbedding({"x": video})
  other = TableConfig(vocabulary_size=2, dim=4, name="video_table")
  with pytest.raises(ValueError, match="unique"):
    TPUEmbedding({"a": FeatureConfig(table=video), "b": FeatureConfig(table=other)},
                 device=torch.device("cpu"))


# ================================================================================ GPU: hashing
@gpu
@pytest.mark.parametrize("dtype", [np.int64, np.int32])
@pytest.mark.parametrize("num_bins,salt", [(10, [0, 0]), (1, [5, 6]), (1000003, [2, 1]),
                                           (2**31 - 1, [123456789, 987654321]),
                                           (2**62 + 7, [2**63 + 5, 2**64 - 1])])
def test_hash_ids_parity(dtype, num_bins, salt):
  rng = np.random.default_rng(7)
  info = np.iinfo(dtype)
  edge = [0, 1, -1, 9, 10, 99, 100, 12345678, 99999999, 100000000, info.max, info.min,
          info.max - 1, info.min + 1]
  edge += [10**k for k in range(1, 19) if 10**k <= info.max]
  edge += [-(10**k) for k in range(1, 19) if 10**k <= info.max]
  ids = np.concatenate([np.asarray(edge, dtype=dtype),
                        rng.integers(0, 10_000_000, size=3000).astype(dtype),
                        rng.integers(info.min, info.max, size=1000, dtype=dtype)])
  layer = tfrs.layers.hashing.Hashing(num_bins=num_bins, salt=salt)
  got = layer(_t(ids))
  assert got.dtype == torch.int64 and got.is_cuda
  np.testing.assert_array_equal(_np(got), o_hash.hash_bucket_strong(ids, num_bins, salt))
  # shape is preserved
  ids2 = ids[:4000].reshape(40, 100)
  np.testing.assert_array_equal(_np(layer(_t(ids2))),
                                o_hash.hash_bucket_strong(ids2, num_bins, salt))


@gpu
def test_hash_strings_parity_and_empty():
  rng = np.random.default_rng(8)
  alphabet = np.array(list("abcdefghijklmnopqrstuvwxyzABCDEFGH 0123456789-_é漢"))
  strs = ["", "a", "romance", "New York", "Movie 999", "é", "漢字"]
  strs += ["".join(rng.choice(alphabet, size=n)) for n in range(0, 41)]
  strs += ["".join(rng.choice(alphabet, size=rng.integers(0, 300))) for _ in range(200)]
  arr = np.asarray(strs, dtype=object)
  layer = tfrs.layers.hashing.Hashing(num_bins=1009, salt=[4, 2])
  np.testing.assert_array_equal(_np(layer(arr)), o_hash.hash_bucket_strong(arr, 1009, [4, 2]))
  # bytes and str of the same text agree; integers hash as their decimal strings
  assert _np(layer(np.asarray([b"drama"])))[0] == _np(layer(np.asarray(["drama"])))[0]
  assert _np(layer(np.asarray(["1234"])))[0] == _np(layer(_t([1234])))[0]
  assert _np(layer(np.asarray(["-5"])))[0] == _np(layer(_t([-5])))[0]
  # 2-D string arrays keep their shape; empty input is fine
  grid = arr[:48].reshape(6, 8)
  np.testing.assert_array_equal(_np(layer(grid)), o_hash.hash_bucket_strong(grid, 1009, [4, 2]))
  assert tuple(layer(_t(np.zeros((0,), np.int64))).shape) == (0,)
  assert tuple(layer(np.asarray([], dtype=object)).shape) == (0,)
  with pytest.raises(ValueError):
    layer(_t(np.zeros(3, np.float32)))


@gpu
def test_hash_ragged_keeps_splits():
  r = RaggedIds.from_nested([["Movie 1", "Movie 2"], [], ["Movie 3"]])
  out = tfrs.layers.hashing.Hashing(num_bins=10, salt=[0, 1])(r)
  assert isinstance(out, RaggedIds) and out.row_splits.tolist() == [0, 2, 2, 3]
  np.testing.assert_array_equal(
      _np(out.values), o_hash.hash_bucket_strong(["Movie 1", "Movie 2", "Movie 3"], 10, [0, 1]))


# ============================================================== GPU: combiner lookup backward
@gpu
@pytest.mark.parametrize("d", [32, 7])
@pytest.mark.parametrize("combiner", ["sum", "mean", "sqrtn"])
@pytest.mark.parametrize("weighted", [False, True])
def test_segment_reduce_backward_rows(d, combiner, weighted):
  rng = np.random.default_rng(d)
  lens = rng.integers(0, 6, size=400)
  lens[:3] = 0                                     # leading empty rows
  splits = np.concatenate([[0], np.cumsum(lens)])
  w = rng.uniform(0.5, 2.0, size=splits[-1]).astype(np.float32) if weighted else None
  g = rng.normal(size=(400, d)).astype(np.float32)
  got = tel.segment_reduce_grad_rows(_t(g), _t(splits), None if w is None else _t(w), combiner,
                                     int(splits[-1]))
  ref = o_emb.lookup_sparse_grad_rows(g, splits, w, combiner)
  if combiner == "sqrtn":     # sqrt(fma-accumulated sum w^2) vs NumPy's: last-bit differences
    np.testing.assert_allclose(_np(got), ref, rtol=1e-6, atol=0)
  else:
    np.testing.assert_array_equal(_np(got), ref)   # one divide, one multiply: exact


# ========================================================================= GPU: TPUEmbedding
def _fixture_layer(optimizer=None):
  """tpu_embedding_layer_test.py:51-80: video table [[0..3],[4..7]] (sum), user table
  [[0,1],[2,3],[4,5],[6,7]] (mean); watched/favorited share the video table."""
  vals = np.arange(8, dtype=np.float32)
  init = lambda shape: vals.reshape(shape)         # tf.constant_initializer(range(8))
  video = TableConfig(vocabulary_size=2, dim=4, initializer=init, combiner="sum",
                      name="video_table")
  user = TableConfig(vocabulary_size=4, dim=2, initializer=init, combiner="mean",
                     name="user_table")
  cfg = {"watched": FeatureConfig(table=video, name="watched"),
         "favorited": FeatureConfig(table=video, name="favorited"),
         "friends": FeatureConfig(table=user, name="friends")}
  return TPUEmbedding(cfg, optimizer), video, user


_FIX = dict(
    watched=dict(indices=[[0, 0], [1, 0], [1, 1], [2, 0], [2, 1], [3, 0]],
                 values=[0, 0, 1, 0, 1, 1], lengths=[1, 2, 2, 1], width=2),
    favorited=dict(indices=[[0, 0], [0, 1], [1, 0], [2, 0], [3, 0], [3, 1]],
                   values=[0, 1, 1, 0, 0, 1], lengths=[2, 1, 1, 2], width=2),
    friends=dict(indices=[[0, 0], [1, 0], [1, 1], [1, 2], [2, 0], [3, 0], [3, 1], [3, 2]],
                 values=[3, 0, 1, 2, 3, 0, 1, 2], lengths=[1, 3, 1, 3], width=3))


def _fixture_inputs(kind):
  if kind == "ragged":       # tpu_embedding_layer_test.py:250-268
    return {k: RaggedIds.from_row_lengths(torch.tensor(v["values"], dtype=torch.int32),
                                          v["lengths"]) for k, v in _FIX.items()}
  return {k: SparseIds(v["indices"], torch.tensor(v["values"], dtype=torch.int32),
                       (4, v["width"])) for k, v in _FIX.items()}   # :227-248


@gpu
@pytest.mark.parametrize("kind", ["ragged", "sparse"])
def test_tpu_embedding_fixture_activations(kind):
  g = load_golden("embedding.json")
  layer, video, user = _fixture_layer()
  tables = layer.embedding_tables
  np.testing.assert_array_equal(_np(tables[video]), g["video_table"])
  np.testing.assert_array_equal(_np(tables[user]), g["user_table"])
  out = layer(_fixture_inputs(kind))
  assert list(out) == ["watched", "favorited", "friends"]
  for k in out:
    np.testing.assert_allclose(_np(out[k]), g[k]["expected"], rtol=1e-6, err_msg=k)
  # the reference's test only asserts "loss + 3 activations" (:206-208); the loss it forms:
  loss = sum(torch.mean(torch.sum(a * a, dim=1)) for a in out.values())
  assert math.isfinite(float(loss.detach()))


@gpu
def test_tpu_embedding_dense_and_weighted_inputs():
  layer, video, user = _fixture_layer()
  # dense ids (tpu_embedding_layer_test.py:270-283): plain lookup, rank preserved
  dense = {"watched": torch.tensor([1, 1]), "favorited": torch.tensor([[0, 1], [1, 1]]),
           "friends": torch.tensor([1, 2], dtype=torch.int32)}
  out = layer(dense)
  vt, ut = np.arange(8, dtype=np.float32).reshape(2, 4), np.arange(8, dtype=np.float32).reshape(4, 2)
  np.testing.assert_array_equal(_np(out["watched"]), vt[[1, 1]])
  np.testing.assert_array_equal(_np(out["favorited"]), vt[[[0, 1], [1, 1]]])
  np.testing.assert_array_equal(_np(out["friends"]), ut[[1, 2]])
  # ragged ids with weights 0.5 (include_weights, :250): sum scales, mean does not
  feats = _fixture_inputs("ragged")
  wts = {k: v.with_values(torch.full((len(v.values),), 0.5)) for k, v in feats.items()}
  out_w = layer(feats, weights=wts)
  base = layer(feats)
  np.testing.assert_allclose(_np(out_w["watched"]), 0.5 * _np(base["watched"]), rtol=1e-6)
  np.testing.assert_allclose(_np(out_w["friends"]), _np(base["friends"]), rtol=1e-6)
  for k, v in _FIX.items():
    table = vt if k != "friends" else ut
    splits = np.concatenate([[0], np.cumsum(v["lengths"])])
    ref = o_emb.lookup_sparse(table, np.asarray(v["values"]), splits,
                              np.full(len(v["values"]), 0.5, np.float32),
                              "mean" if k == "friends" else "sum")
    np.testing.assert_allclose(_np(out_w[k]), ref, rtol=1e-6)
  # serving_config: look up a subset through the same tables (:891-900)
  serving = {"friends": FeatureConfig(table=user, name="friends")}
  got = layer({"friends": feats["friends"]}, serving_config=serving)
  np.testing.assert_array_equal(_np(got["friends"]), _np(base["friends"]))


def _random_case(rng, vocab, nrows, max_len):
  lens = rng.integers(0, max_len + 1, size=nrows)
  splits = np.concatenate([[0], np.cumsum(lens)])
  ids = rng.integers(0, vocab, size=splits[-1])
  w = rng.uniform(0.5, 2.0, size=splits[-1]).astype(np.float32)
  return ids, splits, w


@gpu
@pytest.mark.parametrize("scatter", ["rowscan", "sorted"])
def test_tpu_embedding_backward_shared_table(scatter, monkeypatch):
  """Two ragged features and a dense one on ONE table: table.grad is the sum of the three
  scatter-added IndexedSlices, each bit-exact against the oracle's occurrence-order sums."""
  from recommenders_amd.layers import embedding as emb
  if scatter == "sorted":
    monkeypatch.setattr(emb, "_ROWSCAN_MAX_WORK", 0)
  rng = np.random.default_rng(11)
  vocab, d, nrows = 300, 32, 256
  table0 = rng.normal(size=(vocab, d)).astype(np.float32)
  tc = TableConfig(vocabulary_size=vocab, dim=d, initializer=lambda s: table0, combiner="sqrtn",
                   name="t")
  cfg = [FeatureConfig(table=tc), FeatureConfig(table=tc), FeatureConfig(table=tc)]
  layer = TPUEmbedding(cfg)
  (ids_a, sp_a, w_a), (ids_b, sp_b, _) = _random_case(rng, vocab, nrows, 5), _random_case(
      rng, vocab, nrows, 3)
  ids_c = rng.integers(0, vocab, size=(nrows,))
  feats = [RaggedIds(_t(ids_a), sp_a), RaggedIds(_t(ids_b), sp_b), _t(ids_c)]
  wts = [RaggedIds(_t(w_a), sp_a), None, None]
  out = layer(feats, weights=wts)
  for o, (i, s, w) in zip(out[:2], ((ids_a, sp_a, w_a), (ids_b, sp_b, None))):
    np.testing.assert_allclose(_np(o), o_emb.lookup_sparse(table0, i, s, w, "sqrtn"),
                               rtol=2e-6, atol=1e-6)
  go = [rng.normal(size=(nrows, d)).astype(np.float32) for _ in range(3)]
  sum((o * _t(g)).sum() for o, g in zip(out, go)).backward()
  parts = [
      o_emb.scatter_add_grad(o_emb.lookup_sparse_grad_rows(go[0], sp_a, w_a, "sqrtn"), ids_a, vocab),
      o_emb.scatter_add_grad(o_emb.lookup_sparse_grad_rows(go[1], sp_b, None, "sqrtn"), ids_b, vocab),
      o_emb.scatter_add_grad(go[2], ids_c, vocab)]
  got = _np(layer.embedding_tables[tc].grad)
  # autograd accumulates the three dense gradients in reverse order of use: float32 sums of
  # three terms, order-dependent in the last bit
  ref = (parts[2] + parts[1]) + parts[0]
  np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-6)
  # one feature alone: no cross-feature sum; with the mean combiner (no sqrt) it is bit-exact
  tm = TableConfig(vocabulary_size=vocab, dim=d, initializer=lambda s: table0, combiner="mean",
                   name="m")
  single = TPUEmbedding([FeatureConfig(table=tm)])
  o = single([RaggedIds(_t(ids_a), sp_a)], weights=[RaggedIds(_t(w_a), sp_a)])[0]
  (o * _t(go[0])).sum().backward()
  np.testing.assert_array_equal(
      _np(single.embedding_tables[tm].grad),
      o_emb.scatter_add_grad(o_emb.lookup_sparse_grad_rows(go[0], sp_a, w_a, "mean"), ids_a, vocab))


@gpu
def test_tpu_embedding_sparse_adagrad_slices():
  """With optimizers.Adagrad the lookups hand (ids, grad_rows) slices to the fused sparse
  update: one combined IndexedSlices per table, duplicates summed before the update."""
  rng = np.random.default_rng(12)
  vocab, d, nrows = 500, 16, 128
  table0 = rng.uniform(-0.05, 0.05, size=(vocab, d)).astype(np.float32)
  tc = TableConfig(vocabulary_size=vocab, dim=d, initializer=lambda s: table0, combiner="mean",
                   name="t")
  layer = TPUEmbedding({"a": FeatureConfig(table=tc), "b": FeatureConfig(table=tc)})
  opt = tfrs.optimizers.Adagrad(layer.parameters(), learning_rate=0.1)
  ids_a, sp_a, w_a = _random_case(rng, vocab, nrows, 4)
  ids_b = rng.integers(0, vocab, size=(nrows,))
  out = layer({"a": RaggedIds(_t(ids_a), sp_a), "b": _t(ids_b)},
              weights={"a": RaggedIds(_t(w_a), sp_a), "b": None})
  ga, gb = (rng.normal(size=(nrows, d)).astype(np.float32) for _ in range(2))
  opt.zero_grad()
  ((out["a"] * _t(ga)).sum() + (out["b"] * _t(gb)).sum()).backward()
  p = layer.embedding_tables[tc]
  assert p.grad is None                             # no dense [vocab, d] gradient was built
  slices = list(p._tfrs_slices)
  opt.step()
  ids_all = np.concatenate([_np(s[0]).reshape(-1) for s in slices])
  rows_all = np.concatenate([_np(s[1]).reshape(-1, d) for s in slices])
  rows_a = o_emb.lookup_sparse_grad_rows(ga, sp_a, w_a, "mean")
  assert sorted(ids_all.tolist()) == sorted(np.concatenate([ids_a, ids_b]).tolist())
  t_ref, _ = o_emb.adagrad_sparse_update(table0, np.full_like(table0, 0.1), rows_all, ids_all,
                                         lr=0.1)
  np.testing.assert_allclose(_np(p), t_ref, rtol=1e-5, atol=1e-7)
  # and the slices themselves are the oracle's rows
  for s in slices:
    if s[0].numel() == ids_a.size and np.array_equal(_np(s[0]), ids_a):
      np.testing.assert_array_equal(_np(s[1]), rows_a)


@gpu
@pytest.mark.parametrize("kind", ["ragged", "sparse"])
def test_tpu_embedding_sequence_feature(kind):
  """max_sequence_length > 0: [B, L, D], zero padding, truncation at L; backward skips pads."""
  rng = np.random.default_rng(13)
  vocab, d, nrows, L = 50, 8, 64, 3
  table0 = rng.normal(size=(vocab, d)).astype(np.float32)
  tc = TableConfig(vocabulary_size=vocab, dim=d, initializer=lambda s: table0, name="t")
  layer = TPUEmbedding(FeatureConfig(table=tc, max_sequence_length=L))
  ids, splits, _ = _random_case(rng, vocab, nrows, 5)      # some rows longer than L, some empty
  if kind == "ragged":
    inp, pos = RaggedIds(_t(ids), splits), None
  else:
    rows = np.repeat(np.arange(nrows), np.diff(splits))
    pos = np.arange(splits[-1]) - splits[:-1][rows]
    inp = SparseIds(np.stack([rows, pos], 1), _t(ids), (nrows, 5))
  out = layer(inp)
  ref = o_emb.sequence_lookup(table0, ids, splits, L, pos)
  assert tuple(out.shape) == (nrows, L, d)
  np.testing.assert_array_equal(_np(out), ref)
  g = rng.normal(size=(nrows, L, d)).astype(np.float32)
  (out * _t(g)).sum().backward()
  grad_ref = np.zeros_like(table0)
  for b in range(nrows):
    for j in range(min(L, splits[b + 1] - splits[b])):
      grad_ref[ids[splits[b] + j]] += g[b, j]
  np.testing.assert_allclose(_np(layer.embedding_tables[tc].grad), grad_ref, rtol=1e-6, atol=1e-6)


# ===================================================================== GPU: UnifiedEmbedding
def _ue_dataset():
  """unified_embedding_test.py:30-64."""
  n = 10
  rng = np.random.default_rng(seed=42)
  vocabs = {
      "genre": ["romance", "drama", "fantasy", "action", "comedy", "horror"],
      "year": [str(y) for y in range(1950, 2023)],
      "history": [f"Movie {i}" for i in range(1000)],
  }
  data = {
      "genre": rng.choice(vocabs["genre"], size=n),
      "year": rng.choice(vocabs["year"], size=n),
      "city": rng.choice(vocabs["genre"], size=n),
      "num_watched": (100 * (1.0 - rng.power(4, size=n))).astype(int),
      "history": rng.choice(vocabs["history"], size=[n, 4]),
  }
  lens = rng.integers(1, 10, size=n)
  data["history_varlen"] = RaggedIds.from_nested(
      [rng.choice(vocabs["history"], size=k) for k in lens])
  return n, data


def _ue_reference(layer, cfg, name, feature, combiner="mean"):
  """Oracle composition: hash per chunk -> lookup in the chunk's table -> concat."""
  tables = layer.embedding_layer.embedding_tables
  parts = []
  for chunk in sorted(cfg.embedding_config[name]):
    fc = cfg.embedding_config[name][chunk]
    hp = cfg.hashing_config[name][chunk]
    table = _np(tables[fc.table])
    if isinstance(feature, RaggedIds):
      buckets = o_hash.hash_bucket_strong(np.asarray(feature.values), hp["num_bins"], hp["salt"])
      parts.append(o_emb.lookup_sparse(table, buckets, feature.row_splits.numpy(), None, combiner))
    else:
      buckets = o_hash.hash_bucket_strong(np.asarray(feature), hp["num_bins"], hp["salt"])
      parts.append(o_emb.gather(table, buckets))
  return np.concatenate(parts, axis=-1)


@gpu
@pytest.mark.parametrize("fuse", [True, False])
def test_unified_embedding_reference_cases(fuse, monkeypatch):
  """``fuse=True``: one hash+gather+concat kernel per dense feature; ``fuse=False``: Hashing ->
  lookup -> concat chunk by chunk.  Same values either way."""
  import functools
  from recommenders_amd.layers.feature_multiplexing import unified_embedding as ue_mod
  monkeypatch.setattr(ue_mod, "UnifiedEmbedding",
                      functools.partial(ue_mod.UnifiedEmbedding, fuse=fuse))
  UnifiedEmbedding = ue_mod.UnifiedEmbedding
  n, data = _ue_dataset()
  # test_single_feature (:66-76)
  cfg = UnifiedEmbeddingConfig(buckets_per_table=10, dim_per_table=8, num_tables=1,
                               name="single_ue_table")
  cfg.add_feature("genre", 2)
  layer = UnifiedEmbedding(cfg, optimizer=None)
  out = layer(data)
  assert len(out) == 1 and tuple(out[0].shape) == (n, 16)
  np.testing.assert_array_equal(_np(out[0]), _ue_reference(layer, cfg, "genre", data["genre"]))
  # test_multiple_features (:78-97) and test_feature_output_order (:127-147)
  for order, sizes in ((("genre", 1), ("year", 2), ("city", 3)), (8, 16, 24)), \
                      ((("year", 2), ("genre", 1), ("city", 3)), (16, 8, 24)):
    cfg = UnifiedEmbeddingConfig(buckets_per_table=10, dim_per_table=8, num_tables=3,
                                 name="multiple_ue_table")
    for name, chunks in order:
      cfg.add_feature(name, chunks)
    layer = UnifiedEmbedding(cfg, optimizer=None)
    outs = layer(data)
    assert [tuple(o.shape) for o in outs] == [(n, s) for s in sizes]
    for (name, _), o in zip(order, outs):
      np.testing.assert_array_equal(_np(o), _ue_reference(layer, cfg, name, data[name]))
    assert len(layer.embedding_layer.embedding_tables) == 3
  # test_dense_multivalent (:99-111): no combiner, rank preserved
  cfg = UnifiedEmbeddingConfig(buckets_per_table=10, dim_per_table=8, num_tables=3,
                               name="dense_multivalent_ue_table")
  cfg.add_feature("history", 3)
  layer = UnifiedEmbedding(cfg, optimizer=None)
  out = layer(data)[0]
  assert tuple(out.shape) == (n, 4, 24)
  np.testing.assert_array_equal(_np(out), _ue_reference(layer, cfg, "history", data["history"]))
  # test_sparse_multivalent (:113-125): the table combiner (TableConfig default: mean)
  cfg = UnifiedEmbeddingConfig(buckets_per_table=10, dim_per_table=8, num_tables=3,
                               name="sparse_multivalent_ue_table")
  cfg.add_feature("history_varlen", 3)
  layer = UnifiedEmbedding(cfg, optimizer=None)
  out = layer(data)[0]
  assert tuple(out.shape) == (n, 24)
  np.testing.assert_allclose(
      _np(out), _ue_reference(layer, cfg, "history_varlen", data["history_varlen"]),
      rtol=2e-6, atol=1e-6)
  # integer features hash through the id kernel
  cfg = UnifiedEmbeddingConfig(buckets_per_table=10, dim_per_table=8, num_tables=2, name="ints")
  cfg.add_feature("num_watched", 2)
  layer = UnifiedEmbedding(cfg, optimizer=None)
  out = layer({"num_watched": _t(data["num_watched"])})[0]
  np.testing.assert_array_equal(_np(out),
                                _ue_reference(layer, cfg, "num_watched", data["num_watched"]))
  with pytest.raises(KeyError):
    layer({"genre": data["genre"]})


@gpu
@pytest.mark.parametrize("dim,chunks", [(4, 3), (16, 17), (64, 2), (256, 2), (12, 2)])
def test_unified_fused_lookup_shapes(dim, chunks):
  """The fused kernel at every lane layout (D/4 = 1..64 lanes per row), more chunks than one
  launch descriptor holds (17 > 16), int32/int64/string values, against the oracle; dim 12
  is not 4 * 2^k and takes the per-chunk path."""
  rng = np.random.default_rng(dim)
  n = 1000
  cfg = UnifiedEmbeddingConfig(buckets_per_table=37, dim_per_table=dim, num_tables=3, name="u")
  cfg.add_feature("a", chunks)
  cfg.add_feature("b", 1)
  cfg.add_feature("s", 2)
  layer = UnifiedEmbedding(cfg, optimizer=None)
  feats = {"a": rng.integers(-50, 10**9, size=(n,)), "b": rng.integers(0, 99, size=(8, 5)).astype(np.int32),
           "s": np.asarray([f"item {i}" for i in rng.integers(0, 500, size=n)], dtype=object)}
  outs = layer({"a": _t(feats["a"]), "b": _t(feats["b"]), "s": feats["s"]})
  assert [tuple(o.shape) for o in outs] == [(n, chunks * dim), (8, 5, dim), (n, 2 * dim)]
  for name, o in zip(("a", "b", "s"), outs):
    np.testing.assert_array_equal(_np(o), _ue_reference(layer, cfg, name, feats[name]))
  empty = layer({"a": _t(np.zeros((0,), np.int64)), "b": _t(feats["b"]), "s": feats["s"]})[0]
  assert tuple(empty.shape) == (0, chunks * dim)
  with pytest.raises(ValueError):
    layer({"a": _t(np.zeros(3, np.float32)), "b": _t(feats["b"]), "s": feats["s"]})


@gpu
@pytest.mark.parametrize("sparse_opt", [False, True])
def test_unified_fused_backward(sparse_opt):
  """Backward of the fused lookup: every chunk's (buckets, grad columns) reach its table --
  as dense scatter-added gradients, or as IndexedSlices for the fused sparse Adagrad."""
  rng = np.random.default_rng(21)
  n, dim, bins = 512, 16, 23
  cfg = UnifiedEmbeddingConfig(buckets_per_table=bins, dim_per_table=dim, num_tables=2, name="u")
  cfg.add_feature("a", 3)                       # tables 0, 1, 0
  layer = UnifiedEmbedding(cfg, optimizer=None)
  t0, t1 = (layer.embedding_layer.embedding_tables[t] for t in cfg._table_configs)
  w0, w1 = _np(t0).copy(), _np(t1).copy()
  opt = tfrs.optimizers.Adagrad(layer.parameters(), learning_rate=0.1) if sparse_opt else None
  ids = rng.integers(0, 10**6, size=(n,))
  g = rng.normal(size=(n, 3 * dim)).astype(np.float32)
  out = layer({"a": _t(ids)})[0]
  (out * _t(g)).sum().backward()
  b = [o_hash.hash_bucket_strong(ids, bins, [0, c]) for c in range(3)]
  if not sparse_opt:
    ref0 = o_emb.scatter_add_grad(g[:, :dim], b[0], bins) + o_emb.scatter_add_grad(g[:, 2 * dim:], b[2], bins)
    np.testing.assert_allclose(_np(t0.grad), ref0, rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(_np(t1.grad), o_emb.scatter_add_grad(g[:, dim:2 * dim], b[1], bins))
    return
  assert t0.grad is None and len(t0._tfrs_slices) == 2 and len(t1._tfrs_slices) == 1
  opt.step()
  ref0, _ = o_emb.adagrad_sparse_update(w0, np.full_like(w0, 0.1),
                                        np.concatenate([g[:, :dim], g[:, 2 * dim:]]),
                                        np.concatenate([b[0], b[2]]), lr=0.1)
  ref1, _ = o_emb.adagrad_sparse_update(w1, np.full_like(w1, 0.1), g[:, dim:2 * dim], b[1], lr=0.1)
  np.testing.assert_allclose(_np(t0), ref0, rtol=1e-5, atol=1e-7)
  np.testing.assert_allclose(_np(t1), ref1, rtol=1e-5, atol=1e-7)


@gpu
@pytest.mark.parametrize("sparse_opt", [False, True])
def test_unified_embedding_all_dense_features_in_one_launch(sparse_opt):
  """Every dense integer feature of the layer goes out in ONE launch (tfrs_unified_embedding_fwd_multi;
  round 3: one launch per feature, launch-bound at the DCN-v2 shapes): 26 features x 3 chunks = 78 units
  (more than one launch descriptor holds: 64) over 5 shared tables, int64 and int32 id groups, a 2-D
  feature; values bit for bit the per-feature kernel's (fuse=False path) and the oracle's; the backward
  reaches every table as dense gradients or as IndexedSlices for the fused sparse Adagrad."""
  rng = np.random.default_rng(33)
  n, dim, bins, nf = 700, 16, 101, 26
  cfg = UnifiedEmbeddingConfig(buckets_per_table=bins, dim_per_table=dim, num_tables=5, name="u")
  for f in range(nf):
    cfg.add_feature(f"f{f}", 3)
  cfg.add_feature("two_d", 2)
  cfg.add_feature("i32", 1)
  cfg.add_feature("i32b", 2)
  layer = UnifiedEmbedding(cfg, optimizer=None)
  plain = UnifiedEmbedding(cfg, optimizer=None, fuse=False)
  plain.load_state_dict(layer.state_dict())
  feats = {f"f{f}": rng.integers(0, 10**9, size=(n,)) for f in range(nf)}
  feats["two_d"] = rng.integers(0, 10**6, size=(n // 7, 7))            # same number of ids, other shape
  feats["i32"] = rng.integers(0, 10**6, size=(n,)).astype(np.int32)
  feats["i32b"] = rng.integers(0, 10**6, size=(n,)).astype(np.int32)
  dev_feats = {k: _t(v) for k, v in feats.items()}
  groups = layer._multi_feature_groups(dev_feats)
  assert sorted(len(v) for v in groups.values()) == [2, nf + 1]         # int32 pair; 26 + the 2-D feature
  opt = tfrs.optimizers.Adagrad(layer.parameters(), learning_rate=0.1) if sparse_opt else None
  opt2 = tfrs.optimizers.Adagrad(plain.parameters(), learning_rate=0.1) if sparse_opt else None
  outs, outs2 = layer(dev_feats), plain(dev_feats)
  names = [f"f{f}" for f in range(nf)] + ["two_d", "i32", "i32b"]
  assert tuple(outs[nf].shape) == (n // 7, 7, 2 * dim)
  for name, o, o2 in zip(names, outs, outs2):
    np.testing.assert_array_equal(_np(o), _np(o2))
    np.testing.assert_array_equal(_np(o), _ue_reference(layer, cfg, name, feats[name]))
  ws = [_t(rng.normal(size=tuple(o.shape)).astype(np.float32)) for o in outs]
  sum((o * w).sum() for o, w in zip(outs, ws)).backward()
  sum((o * w).sum() for o, w in zip(outs2, ws)).backward()
  ta, tb = layer.embedding_layer.embedding_tables, plain.embedding_layer.embedding_tables
  if sparse_opt:
    opt.step()
    opt2.step()
    # (a table receives the slices / partial gradients of ~16 units; the two paths add them in different
    # orders: float32 sums of ~100 terms each, compared at a few ulps of the sum's magnitude)
    for t in cfg._table_configs:
      np.testing.assert_allclose(_np(ta[t]), _np(tb[t]), rtol=2e-5, atol=2e-6)
  else:
    for t in cfg._table_configs:
      np.testing.assert_allclose(_np(ta[t].grad), _np(tb[t].grad), rtol=2e-5, atol=2e-5)


@gpu
def test_unified_embedding_trains_shared_tables():
  """Gradients of all chunks that share a table accumulate into that table; state_dict
  round-trips (the reference's save/load test, :149-168, needs SavedModel)."""
  n, data = _ue_dataset()
  cfg = UnifiedEmbeddingConfig(buckets_per_table=10, dim_per_table=8, num_tables=2, name="ue")
  cfg.add_feature("year", 1)
  cfg.add_feature("city", 3)
  layer = UnifiedEmbedding(cfg, optimizer=None)
  outs = layer(data)
  sum((o * o).sum() for o in outs).backward()
  tables = layer.embedding_layer.embedding_tables
  assert len(tables) == 2 and all(p.grad is not None for p in tables.values())
  # table 0 serves year/0 and city/1; table 1 serves city/0 and city/2
  t0, t1 = (tables[t] for t in cfg._table_configs)
  g0 = np.zeros((10, 8), np.float32)
  for name, chunk_id, sl in (("city", 1, slice(8, 16)), ("year", 0, slice(0, 8))):
    chunk = f"ue_{name}_lookup_{chunk_id}"
    hp = cfg.hashing_config[name][chunk]
    b = o_hash.hash_bucket_strong(data[name], hp["num_bins"], hp["salt"])
    o = _np(outs[0 if name == "year" else 1])[:, sl]
    g0 = g0 + o_emb.scatter_add_grad(2 * o, b, 10)
  np.testing.assert_allclose(_np(t0.grad), g0, rtol=1e-5, atol=1e-6)
  clone = UnifiedEmbedding(cfg, optimizer=None)
  clone.load_state_dict(layer.state_dict())
  for a, b in zip(clone(data), outs):
    np.testing.assert_array_equal(_np(a), _np(b))
  assert t1.grad.abs().sum() > 0


# ================================================================== GPU: PartialTPUEmbedding
@gpu
@pytest.mark.parametrize("threshold", [10, None, 0])
def test_partial_embedding_lookup(threshold):
  """partial_tpu_embedding_test.py:57-137: scalar ids in, one [dim] vector per feature out."""
  cfg = _ref_feature_config([5, 20, 8, 9, 15], [2, 4, 6, 8, 10])
  layer = tfrs.experimental.layers.embedding.PartialTPUEmbedding(cfg, optimizer=None,
                                                                 size_threshold=threshold)
  inputs = {"0": 4, "1": 10, "2": 6, "3": 8, "4": 0}
  out = layer(inputs)
  assert set(out) == set(inputs)
  for key, val in out.items():
    assert tuple(val.shape) == (cfg[key].table.dim,)
    if key in layer.keras_embedding_layers:
      table = layer.keras_embedding_layers[key].embeddings
    else:
      table = layer.tpu_embedding.embedding_tables[cfg[key].table]
    np.testing.assert_array_equal(_np(val), _np(table)[inputs[key]])
  if threshold == 10:
    with pytest.raises(ValueError, match="dense tensor input"):
      layer({**inputs, "0": RaggedIds.from_nested([[1]])})
    # batched ids and a ragged feature on a large table
    big = layer({"0": torch.tensor([1, 2]), "1": RaggedIds.from_nested([[1, 2], [3]]),
                 "2": torch.tensor([0, 0]), "3": torch.tensor([3, 4]), "4": torch.tensor([5, 6])})
    t1 = _np(layer.tpu_embedding.embedding_tables[cfg["1"].table])
    np.testing.assert_allclose(_np(big["1"]), [(t1[1] + t1[2]) / 2, t1[3]], rtol=1e-6)
