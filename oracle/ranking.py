"""Oracle for the ranking widening (TEST INFRASTRUCTURE ONLY): ``layers/blocks.py:24-61``
(MLP of Keras Dense layers), ``tasks/ranking.py:77-119`` (loss + metric plumbing) and the
data flow of ``experimental/models/ranking.py:208-236``.

Pinned by the reference's known answers: ``tasks/ranking_test.py:28-62`` (loss
``-(ln 1 + ln 0.3)/2``, accuracy 0.5, label mean 1.0, prediction mean 0.65).  The model test
of the reference (``experimental/models/ranking_test.py``) only asserts that training lowers
the loss, so the forward composition below is PARITY UNPINNED beyond its parts (Dense /
DotInteraction / Cross, each pinned by its own golden vectors).

Keras semantics hard-coded here: ``Dense = act(x @ kernel[in,out] + bias)``;
``BinaryCrossentropy(from_logits=False)`` clips probabilities to [1e-7, 1-1e-7] and averages
over the last axis; reduction AUTO = sum over samples / number of samples.
"""

from typing import Dict, List, Optional, Sequence

import numpy as np

from oracle import feature_interaction as o_fi

_EPS = 1e-7

_ACT = {
    None: lambda x: x,
    "relu": lambda x: np.maximum(x, 0.0),
    "sigmoid": lambda x: 1.0 / (1.0 + np.exp(-x)),
    "tanh": np.tanh,
}


def mlp(x: np.ndarray, kernels: Sequence[np.ndarray], biases: Sequence[Optional[np.ndarray]],
        activation="relu", final_activation=None) -> np.ndarray:
  """blocks.py:54-59 (float64 accumulation, float32 result)."""
  h = np.asarray(x, dtype=np.float64)
  n = len(kernels)
  for li, (k, b) in enumerate(zip(kernels, biases)):
    h = h @ np.asarray(k, dtype=np.float64)
    if b is not None:
      h = h + np.asarray(b, dtype=np.float64)
    h = _ACT[final_activation if li == n - 1 else activation](h)
  return h.astype(np.float32)


def binary_crossentropy(y_true, y_pred, sample_weight=None, reduction="sum_over_batch_size"):
  y_pred = np.asarray(y_pred, dtype=np.float64)
  y_true = np.asarray(y_true, dtype=np.float64).reshape(y_pred.shape)
  p = np.clip(y_pred, _EPS, 1.0 - _EPS)
  bce = -(y_true * np.log(p) + (1.0 - y_true) * np.log(1.0 - p))
  per = bce.mean(axis=-1) if bce.ndim > 1 else bce
  if sample_weight is not None:
    w = np.asarray(sample_weight, dtype=np.float64)
    if w.ndim == per.ndim + 1 and w.shape[-1] == 1:
      w = w[..., 0]
    per = per * w
  if reduction == "none":
    return per.astype(np.float32)
  if reduction == "sum":
    return np.float32(per.sum())
  return np.float32(per.sum() / per.size)


def ranking_task(labels, predictions, sample_weight=None) -> Dict[str, float]:
  """tasks/ranking.py:92-115 with the metrics of ranking_test.py:30-35."""
  loss = float(binary_crossentropy(labels, predictions, sample_weight))
  p = np.asarray(predictions, dtype=np.float64).reshape(-1)
  y = np.asarray(labels, dtype=np.float64).reshape(-1)
  w = np.ones_like(p) if sample_weight is None else np.asarray(sample_weight, np.float64).reshape(-1)
  return {
      "loss": loss,
      "accuracy": float(np.average(((p > 0.5).astype(np.float64) == y), weights=w)),
      "label_mean": float(np.average(y, weights=w)),
      "prediction_mean": float(np.average(p, weights=w)),
      "loss_mean": loss,
  }


def _cross_stack(x0, kernels, biases):
  """``x_{l+1} = x0 * (x_l W_l + b_l) + x_l`` for every layer of the stack (dcn.py:151-186 applied
  ``len(kernels)`` times to the same ``x0``, as a DCN with several cross layers does), float64;
  returns every ``x_l`` (``xs[0] = x0``)."""
  xs = [np.asarray(x0, dtype=np.float64)]
  for w, b in zip(kernels, biases):
    z = xs[-1] @ np.asarray(w, dtype=np.float64)
    if b is not None:
      z = z + np.asarray(b, dtype=np.float64)
    xs.append(xs[0] * z + xs[-1])
  return xs


def _as_stack(cross_kernel, cross_bias):
  if isinstance(cross_kernel, (list, tuple)):
    biases = list(cross_bias) if cross_bias is not None else [None] * len(cross_kernel)
    return list(cross_kernel), biases
  return [cross_kernel], [cross_bias]


def ranking_model_forward(dense_features, sparse_embeddings: List[np.ndarray], bottom, top,
                          interaction: str, concat_dense: bool = True,
                          cross_kernel=None, cross_bias=None) -> np.ndarray:
  """experimental/models/ranking.py:208-236.  ``bottom`` / ``top`` = (kernels, biases,
  activation, final_activation); ``interaction`` = "dot" (DotInteraction defaults) or "cross"
  (Concatenate + full-rank Cross with the given kernel/bias; lists of kernels / biases = a stack of
  cross layers on the same ``x0``, the "3 Cross layers" of BASELINE configs[3])."""
  dense_vec = mlp(dense_features, *bottom)
  args = list(sparse_embeddings) + [dense_vec]
  if interaction == "dot":
    inter = o_fi.dot_interaction(args)
  else:
    x0 = np.concatenate(args, axis=-1)
    kernels, biases = _as_stack(cross_kernel, cross_bias)
    if len(kernels) == 1:
      inter = o_fi.cross(x0, None, kernel=kernels[0], bias=biases[0])
    else:
      inter = _cross_stack(x0, kernels, biases)[-1].astype(np.float32)
  feat = np.concatenate([dense_vec, inter], axis=1) if concat_dense else inter
  return mlp(feat, *top).reshape(-1)


def _mlp_forward64(x, kernels, biases, activation, final_activation):
  """Float64 forward that keeps every layer's input and pre-activation (for the backward below)."""
  h = np.asarray(x, dtype=np.float64)
  ins, pre = [], []
  n = len(kernels)
  for li, (k, b) in enumerate(zip(kernels, biases)):
    ins.append(h)
    z = h @ np.asarray(k, dtype=np.float64)
    if b is not None:
      z = z + np.asarray(b, dtype=np.float64)
    pre.append(z)
    h = _ACT[final_activation if li == n - 1 else activation](z)
  return h, ins, pre


def _act_grad(name, z):
  if name is None:
    return np.ones_like(z)
  if name == "relu":
    return (z > 0).astype(np.float64)
  if name == "sigmoid":
    s = 1.0 / (1.0 + np.exp(-z))
    return s * (1.0 - s)
  if name == "tanh":
    return 1.0 - np.tanh(z) ** 2
  raise ValueError(name)


def _mlp_backward64(dout, ins, pre, kernels, activation, final_activation):
  """Gradient wrt the MLP's input (``tape.gradient`` through blocks.py:54-59), float64."""
  g = dout
  n = len(kernels)
  for li in range(n - 1, -1, -1):
    g = g * _act_grad(final_activation if li == n - 1 else activation, pre[li])
    g = g @ np.asarray(kernels[li], dtype=np.float64).T
  return g


def ranking_model_embedding_grads(dense_features, sparse_embeddings: List[np.ndarray], labels, bottom, top,
                                  interaction: str, batch_size: int, concat_dense: bool = True,
                                  cross_kernel=None, cross_bias=None):
  """What ``tape.gradient(loss, embedding rows)`` returns for the given examples
  (``models/base.py:77`` under ``experimental/models/ranking.py:135-236``): the loss is the MEAN over the
  ``batch_size`` examples of the per-example binary cross-entropy (``:118-121`` reduction NONE, then
  ``:203-206``), and an example's prediction depends on no other example, so its gradient wrt its own
  embedding vectors can be restated from that example alone.  Returns ``(predictions [n] float32,
  d loss / d sparse_embeddings [n, F, D] float64, d loss / d bottom-stack output [n, D] float64)``.
  PARITY UNPINNED like the forward (the reference's model test asserts only a falling loss); the
  backward is checked against central differences of the forward in ``tests/test_ranking.py``."""
  f = len(sparse_embeddings)
  dense_vec, b_ins, b_pre = _mlp_forward64(dense_features, *bottom)
  embs = [np.asarray(e, dtype=np.float64) for e in sparse_embeddings]
  d = embs[0].shape[1]
  args = embs + [dense_vec]
  if interaction == "dot":
    x = np.stack(args, axis=1)                           # dot_interaction.py:74-104 in float64
    gram = np.einsum("bfd,bgd->bfg", x, x)
    mask = np.tril(np.ones((f + 1, f + 1)), -1).astype(bool)
    inter = gram[:, mask]
  else:
    kernels, biases = _as_stack(cross_kernel, cross_bias)
    xs = _cross_stack(np.concatenate(args, axis=-1), kernels, biases)
    inter = xs[-1]
  feat = np.concatenate([dense_vec, inter], axis=1) if concat_dense else inter
  out, t_ins, t_pre = _mlp_forward64(feat, *top)
  p = out.reshape(-1)
  y = np.asarray(labels, dtype=np.float64).reshape(-1)
  pc = np.clip(p, _EPS, 1.0 - _EPS)
  inside = (p > _EPS) & (p < 1.0 - _EPS)
  dp = np.where(inside, -(y / pc) + (1.0 - y) / (1.0 - pc), 0.0) / float(batch_size)
  dfeat = _mlp_backward64(dp.reshape(-1, 1), t_ins, t_pre, top[0], top[2], top[3])
  d_dense = dfeat[:, :d].copy() if concat_dense else np.zeros_like(dense_vec)
  dinter = dfeat[:, d:] if concat_dense else dfeat
  if interaction == "dot":
    g = np.zeros((p.shape[0], f + 1, f + 1))
    g[:, mask] = dinter
    dx = np.einsum("bfg,bgd->bfd", g, x) + np.einsum("bgf,bgd->bfd", g, x)
  else:
    x0 = xs[0]
    dx0 = np.zeros_like(x0)
    dy = dinter
    for li in range(len(kernels) - 1, -1, -1):
      w = np.asarray(kernels[li], dtype=np.float64)
      z = xs[li] @ w + (0.0 if biases[li] is None else np.asarray(biases[li], dtype=np.float64))
      dx0 += dy * z
      dy = (dy * x0) @ w.T + dy
    dx = (dx0 + dy).reshape(p.shape[0], f + 1, d)
  d_dense += dx[:, -1, :]
  return p.astype(np.float32), dx[:, :f, :], d_dense
