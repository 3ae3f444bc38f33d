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


def ranking_model_forward(dense_features, sparse_embeddings: List[np.ndarray], bottom, top,
                          interaction: str, concat_dense: bool = True,
                          cross_kernel=None, cross_bias=None) -> np.ndarray:
  """experimental/models/ranking.py:208-236.  ``bottom`` / ``top`` = (kernels, biases,
  activation, final_activation); ``interaction`` = "dot" (DotInteraction defaults) or "cross"
  (Concatenate + full-rank Cross with the given kernel/bias)."""
  dense_vec = mlp(dense_features, *bottom)
  args = list(sparse_embeddings) + [dense_vec]
  if interaction == "dot":
    inter = o_fi.dot_interaction(args)
  else:
    x0 = np.concatenate(args, axis=-1)
    inter = o_fi.cross(x0, None, kernel=cross_kernel, bias=cross_bias)
  feat = np.concatenate([dense_vec, inter], axis=1) if concat_dense else inter
  return mlp(feat, *top).reshape(-1)
