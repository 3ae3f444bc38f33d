"""Oracle for ``tfrs.metrics.FactorizedTopK`` (test infrastructure only).

Follows ``metrics/factorized_top_k.py:91-194``.  Pinned by
``metrics/factorized_top_k_test.py:39-86`` and ``:93-131`` through
``tests/golden/metric_*.json``.
"""

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from oracle import _clib

F32_MIN = np.finfo(np.float32).min


def in_top_k(targets: np.ndarray, predictions: np.ndarray, k: int) -> np.ndarray:
  """``tf.math.in_top_k``: true iff fewer than k entries are
  STRICTLY greater than the target's prediction; false for a non-finite target."""
  predictions = np.asarray(predictions, dtype=np.float32)
  t = predictions[np.arange(predictions.shape[0]), np.asarray(targets)]
  greater = (predictions > t[:, None]).sum(axis=1)
  return (greater < k) & np.isfinite(t)


def positive_scores(q: np.ndarray, c: np.ndarray) -> np.ndarray:
  """``reduce_sum(q * c, axis=1, keepdims=True)`` :133-134, as the same d-ordered
  fmaf chain the scoring uses, so the positive ties exactly with its own copy in
  the corpus."""
  q = np.ascontiguousarray(q, dtype=np.float32)
  c = np.ascontiguousarray(c, dtype=np.float32)
  out = np.empty((q.shape[0],), dtype=np.float32)
  _clib.lib().oracle_rowdot_f32(_clib.fptr(q), _clib.fptr(c),
                                _clib.i64(q.shape[0]), _clib.i64(q.shape[1]),
                                _clib.fptr(out))
  return out[:, None]


def update(retrieve: Callable[[np.ndarray, int], Tuple[np.ndarray, np.ndarray]],
           ks: Sequence[int], query_embeddings: np.ndarray,
           true_candidate_embeddings: np.ndarray,
           true_candidate_ids: Optional[np.ndarray] = None
           ) -> List[np.ndarray]:
  """One ``update_state`` (:91-194): returns, per k, the per-example 0/1 hit
  vector that the Keras ``Mean`` then averages (weighted by ``sample_weight``)."""
  pos = positive_scores(query_embeddings, true_candidate_embeddings)     # :133-134
  top_scores, retrieved_ids = retrieve(query_embeddings, max(ks))        # :136-137
  hits = []
  if true_candidate_ids is not None:                                     # :141-180
    ids = np.asarray(true_candidate_ids)
    if ids.ndim == 1:
      ids = ids[:, None]
    nan_pad = np.isnan(top_scores)
    top_scores = np.where(nan_pad, F32_MIN, top_scores)
    assert (top_scores[:, :-1] - top_scores[:, 1:] >= 0).all(), \
        "Top-K predictions must be sorted."
    match = ((ids == retrieved_ids) & ~nan_pad).astype(np.float32)
    for k in ks:
      hits.append(np.clip(match[:, :k].sum(axis=1), 0.0, 1.0))
  else:                                                                  # :181-192
    y_pred = np.concatenate([pos, top_scores], axis=1)
    targets = np.zeros((y_pred.shape[0],), dtype=np.int64)
    for k in ks:
      hits.append(in_top_k(targets, y_pred, k).astype(np.float32))
  return hits


def weighted_mean(values: np.ndarray, sample_weight: Optional[np.ndarray]) -> float:
  """``tf.keras.metrics.Mean.update_state(values, sample_weight)`` then
  ``result()`` for a single update: sum(v*w)/sum(w)."""
  v = np.asarray(values, dtype=np.float32).reshape(-1)
  if sample_weight is None:
    return float(v.mean()) if v.size else 0.0
  w = np.asarray(sample_weight, dtype=np.float32).reshape(-1)
  return float((v * w).sum() / w.sum())
