"""Ranking widening: blocks.MLP, tasks.Ranking, experimental Ranking model.

CPU part: the oracle against the reference's known answers (tasks/ranking_test.py:28-62).
GPU part (``-m gpu``): the HIP-backed classes against the oracle."""

import math

import numpy as np
import pytest

from oracle import embedding as o_emb
from oracle import ranking as o_rank

torch = pytest.importorskip("torch")


def test_oracle_ranking_task_kat():
  """tasks/ranking_test.py:39-62: predictions [[1], [0.3]], labels [[1], [1]]."""
  for w in (None, [1.0, 1.0]):
    got = o_rank.ranking_task([[1.0], [1.0]], [[1.0], [0.3]], w)
    expected_loss = -(math.log(1) + math.log(0.3)) / 2.0
    np.testing.assert_allclose(got["loss"], expected_loss, rtol=1e-6, atol=1e-6)
    assert got["accuracy"] == 0.5 and got["label_mean"] == 1.0
    np.testing.assert_allclose(got["prediction_mean"], 0.65, rtol=1e-6)


def test_oracle_mlp_shapes_and_relu():
  x = np.array([[1.0, -2.0]], np.float32)
  out = o_rank.mlp(x, [np.array([[1.0, -1.0], [0.5, 0.5]], np.float32), np.array([[2.0], [3.0]], np.float32)],
                   [np.array([0.0, 0.1], np.float32), None], "relu", None)
  # layer 1: [0, -1.9] -> relu [0, 0]; layer 2: 0
  np.testing.assert_allclose(out, [[0.0]], atol=1e-7)


@pytest.mark.parametrize("interaction", ["dot", "cross"])
@pytest.mark.parametrize("concat_dense", [True, False])
def test_oracle_ranking_embedding_grads_match_central_differences(interaction, concat_dense):
  """The oracle's per-example backward of experimental/models/ranking.py:208-236 (used by the GPU tests at the
  BASELINE configs[3]/[4] shapes) against central differences of its own float64 forward, and its predictions
  against `ranking_model_forward` (three cross layers on one x0; DotInteraction defaults)."""
  rng = np.random.default_rng(0)
  n, F, D, nd, B = 4, 3, 4, 6, 64
  width = (D + (F + 1) * F // 2) if interaction == "dot" else (D + (F + 1) * D)
  if not concat_dense:
    width -= D
  bottom = ([rng.normal(size=(nd, 7)) * .5, rng.normal(size=(7, D)) * .5],
            [rng.normal(size=7) * .1, rng.normal(size=D) * .1], "relu", "relu")
  top = ([rng.normal(size=(width, 6)) * .5, rng.normal(size=(6, 1)) * .5],
         [rng.normal(size=6) * .1, rng.normal(size=1) * .1], "relu", "sigmoid")
  ck = [rng.normal(size=((F + 1) * D, (F + 1) * D)) * .2 for _ in range(3)]
  cb = [rng.normal(size=(F + 1) * D) * .1 for _ in range(3)]
  dense = rng.uniform(size=(n, nd))
  embs = [rng.normal(size=(n, D)) for _ in range(F)]
  y = rng.integers(0, 2, size=n)
  p, dx, d_dense = o_rank.ranking_model_embedding_grads(dense, embs, y, bottom, top, interaction, B,
                                                        concat_dense, ck, cb)
  ref = o_rank.ranking_model_forward(dense.astype(np.float32), [e.astype(np.float32) for e in embs], bottom,
                                     top, interaction, concat_dense, ck, cb)
  np.testing.assert_allclose(p, ref, rtol=1e-6, atol=1e-7)

  def loss64(es):
    dv, _, _ = o_rank._mlp_forward64(dense, *bottom)
    args = [np.asarray(e, float) for e in es] + [dv]
    if interaction == "dot":
      x = np.stack(args, 1)
      it = np.einsum("bfd,bgd->bfg", x, x)[:, np.tril(np.ones((F + 1, F + 1)), -1).astype(bool)]
    else:
      it = o_rank._cross_stack(np.concatenate(args, -1), ck, cb)[-1]
    out, _, _ = o_rank._mlp_forward64(np.concatenate([dv, it], 1) if concat_dense else it, *top)
    pc = np.clip(out.reshape(-1), 1e-7, 1 - 1e-7)
    return (-(y * np.log(pc) + (1 - y) * np.log(1 - pc))).sum() / B

  num, h = np.zeros_like(dx), 1e-6
  for f in range(F):
    for i in range(n):
      for k in range(D):
        e1 = [e.copy() for e in embs]
        e2 = [e.copy() for e in embs]
        e1[f][i, k] += h
        e2[f][i, k] -= h
        num[i, f, k] = (loss64(e1) - loss64(e2)) / (2 * h)
  assert np.abs(num - dx).max() <= 1e-8 * max(1.0, np.abs(dx).max() / 1e-2)
  assert d_dense.shape == (n, D)


# --------------------------------------------------------------------------- GPU
def _np(t):
  return t.detach().cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("use_weights", [False, True])
def test_ranking_task_kat_gpu(use_weights):
  import recommenders_amd as tfrs
  task = tfrs.tasks.Ranking(
      metrics=[tfrs.metrics.BinaryAccuracy(name="accuracy")],
      label_metrics=[tfrs.metrics.Mean(name="label_mean")],
      prediction_metrics=[tfrs.metrics.Mean(name="prediction_mean")],
      loss_metrics=[tfrs.metrics.Mean(name="loss_mean")])
  predictions = torch.tensor([[1.0], [0.3]], device="cuda")
  labels = torch.tensor([[1.0], [1.0]], device="cuda")
  w = torch.tensor([1.0, 1.0], device="cuda") if use_weights else None
  loss = task(predictions=predictions, labels=labels, sample_weight=w)
  expected_loss = -(math.log(1) + math.log(0.3)) / 2.0
  np.testing.assert_allclose(float(loss), expected_loss, rtol=1e-6, atol=1e-6)
  got = {m.name: float(m.result()) for m in task.metrics}
  np.testing.assert_allclose(got["accuracy"], 0.5)
  np.testing.assert_allclose(got["label_mean"], 1.0)
  np.testing.assert_allclose(got["prediction_mean"], 0.65, rtol=1e-6)
  np.testing.assert_allclose(got["loss_mean"], expected_loss, rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_mlp_matches_oracle_and_backprops():
  import recommenders_amd as tfrs
  rng = np.random.default_rng(3)
  x = rng.normal(size=(200, 13)).astype(np.float32)
  mlp = tfrs.layers.blocks.MLP(units=[40, 16, 1], final_activation="sigmoid")
  xt = torch.as_tensor(x).cuda().requires_grad_(True)
  out = mlp(xt)
  ks = [_np(l.kernel) for l in mlp._sublayers]
  bs = [_np(l.bias) for l in mlp._sublayers]
  np.testing.assert_allclose(_np(out), o_rank.mlp(x, ks, bs, "relu", "sigmoid"), rtol=2e-5, atol=1e-6)
  out.sum().backward()
  assert xt.grad is not None and all(l.kernel.grad is not None for l in mlp._sublayers)


@pytest.mark.gpu
@pytest.mark.parametrize("interaction", ["dot", "cross"])
@pytest.mark.parametrize("concat_dense", [True, False])
def test_ranking_model_forward_and_training(interaction, concat_dense):
  """Forward of experimental.models.Ranking against the oracle composition, then a few
  train steps with optimizers.Adagrad must lower the loss (what
  experimental/models/ranking_test.py asserts)."""
  import recommenders_amd as tfrs
  from recommenders_amd.experimental.models import ranking as rk
  rng = np.random.default_rng(11)
  B, D, num_dense = 256, 16, 8
  vocab = {"0": 40, "1": 70, "2": 25}
  emb_layer = rk.EmbeddingDict(vocab, D)
  bottom = tfrs.layers.blocks.MLP(units=[40, D], final_activation="relu")
  top = tfrs.layers.blocks.MLP(units=[40, 20, 1], final_activation="sigmoid")
  fi = tfrs.layers.feature_interaction.DotInteraction() if interaction == "dot" else rk.ConcatCross()
  model = rk.Ranking(emb_layer, bottom_stack=bottom, feature_interaction=fi, top_stack=top,
                     concat_dense=concat_dense)
  dense = rng.uniform(size=(B, num_dense)).astype(np.float32)
  sparse = {k: rng.integers(0, v, size=(B,)) for k, v in vocab.items()}
  labels = ((dense.mean(axis=1) + sum(sparse[k] for k in vocab) / sum(vocab.values())) / 2 + 0.5).astype(np.int64)
  feats = {"dense_features": torch.as_tensor(dense).cuda(),
           "sparse_features": {k: torch.as_tensor(v).cuda() for k, v in sparse.items()}}
  pred = model(feats)
  assert tuple(pred.shape) == (B,)
  embs = [o_emb.gather(_np(emb_layer.tables[k].embeddings), sparse[k]) for k in sorted(vocab)]
  bt = ([_np(l.kernel) for l in bottom._sublayers], [_np(l.bias) for l in bottom._sublayers], "relu", "relu")
  tp = ([_np(l.kernel) for l in top._sublayers], [_np(l.bias) for l in top._sublayers], "relu", "sigmoid")
  ck = _np(fi.cross.kernel) if interaction == "cross" else None
  cb = _np(fi.cross.bias) if interaction == "cross" else None
  ref = o_rank.ranking_model_forward(dense, embs, bt, tp, interaction, concat_dense, ck, cb)
  np.testing.assert_allclose(_np(pred), ref, rtol=5e-5, atol=2e-6)

  model.compile(optimizer=tfrs.optimizers.Adagrad(model.parameters(), learning_rate=0.05))
  batch = (feats, torch.as_tensor(labels).cuda())
  first = float(model.train_step(batch)["loss"])
  for _ in range(30):
    last = float(model.train_step(batch)["loss"])
  assert last < first, (first, last)
  # (EmbeddingDict keeps its tables as row ranges of one parameter: one gather, one sparse update)
  assert len(model.embedding_trainable_variables) == 1
  assert model.embedding_trainable_variables[0].shape[0] == sum(vocab.values())
  assert len(model.dense_trainable_variables) >= 10
  with pytest.raises(ValueError):
    model.compute_loss((feats,))
