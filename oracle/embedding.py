"""Oracle for the embedding lookup path (test infrastructure only).

Semantics of ``tf.keras.layers.Embedding`` as called at ``README.md:62-66,77-78``
and of the CPU branch of ``TPUEmbedding.call``
(``layers/embedding/tpu_embedding_layer.py:913-919``, serving lookup :891-900),
restated from SURVEY.md App. A.5-A.7.

PARITY UNPINNED: the reference's tests assert only shapes / non-None for this
path (``tpu_embedding_layer_test.py:206-208``).  The known-answer vectors in
``tests/golden/embedding.json`` are derived from the fixture at
``tpu_embedding_layer_test.py:51-111`` by this restatement itself, not asserted
by the reference.
"""

from typing import Optional, Tuple

import numpy as np


def gather(table: np.ndarray, ids: np.ndarray) -> np.ndarray:
  """Dense ids: ``out[...] = table[ids[...]]``; rank preserved; out-of-range ids
  are an error (Keras Embedding on CPU)."""
  table = np.asarray(table, dtype=np.float32)
  ids = np.asarray(ids)
  if ids.size and (ids.min() < 0 or ids.max() >= table.shape[0]):
    raise IndexError("embedding id out of range")
  return table[ids]


def lookup_sparse(table: np.ndarray, ids: np.ndarray, row_splits: np.ndarray,
                  weights: Optional[np.ndarray] = None, combiner: str = "mean"
                  ) -> np.ndarray:
  """Ragged/sparse ids in CSR form: row b owns ``ids[row_splits[b]:row_splits[b+1]]``.
  ``sum``: sum w_j e_j; ``mean``: / sum w_j; ``sqrtn``: / sqrt(sum w_j^2); empty
  rows -> zeros.  Accumulation in id order, float32."""
  table = np.asarray(table, dtype=np.float32)
  ids = np.asarray(ids)
  row_splits = np.asarray(row_splits)
  nrows = row_splits.shape[0] - 1
  out = np.zeros((nrows, table.shape[1]), dtype=np.float32)
  for b in range(nrows):
    lo, hi = int(row_splits[b]), int(row_splits[b + 1])
    if hi == lo:
      continue
    w = (np.ones(hi - lo, dtype=np.float32) if weights is None
         else np.asarray(weights[lo:hi], dtype=np.float32))
    acc = np.zeros((table.shape[1],), dtype=np.float32)
    for j in range(hi - lo):
      acc = (acc + w[j] * table[ids[lo + j]]).astype(np.float32)
    if combiner == "sum":
      pass
    elif combiner == "mean":
      acc = acc / np.float32(w.sum(dtype=np.float32))
    elif combiner == "sqrtn":
      acc = acc / np.float32(np.sqrt((w * w).sum(dtype=np.float32)))
    else:
      raise ValueError(f"unknown combiner {combiner}")
    out[b] = acc
  return out


def lookup_sparse_grad_rows(grad_out: np.ndarray, row_splits: np.ndarray,
                            weights: Optional[np.ndarray] = None, combiner: str = "mean"
                            ) -> np.ndarray:
  """Backward of ``lookup_sparse`` wrt the looked-up rows (the values of the IndexedSlices
  gradient TensorFlow's tape produces for the table, ``models/base.py:77``): following the
  autodiff of ``sum_j w_j e_j / den`` -- divide the incoming gradient by ``den`` first, then
  scale by ``w_p`` -- ``grad_rows[p] = (grad_out[b] / den_b) * w_p`` in float32."""
  grad_out = np.asarray(grad_out, dtype=np.float32)
  row_splits = np.asarray(row_splits)
  nnz = int(row_splits[-1])
  rows = np.zeros((nnz, grad_out.shape[1]), dtype=np.float32)
  for b in range(row_splits.shape[0] - 1):
    lo, hi = int(row_splits[b]), int(row_splits[b + 1])
    if hi == lo:
      continue
    w = (np.ones(hi - lo, dtype=np.float32) if weights is None
         else np.asarray(weights[lo:hi], dtype=np.float32))
    if combiner == "sum":
      den = np.float32(1.0)
    elif combiner == "mean":
      den = np.float32(0.0)
      for x in w:
        den = np.float32(den + x)
    elif combiner == "sqrtn":
      sq = np.float32(0.0)
      for x in w:
        sq = np.float32(sq + x * x)
      den = np.float32(np.sqrt(sq))
    else:
      raise ValueError(f"unknown combiner {combiner}")
    g = (grad_out[b] / den).astype(np.float32)
    for j in range(hi - lo):
      rows[lo + j] = g * w[j]
  return rows


def sequence_lookup(table: np.ndarray, ids: np.ndarray, row_splits: np.ndarray,
                    max_sequence_length: int, positions: Optional[np.ndarray] = None
                    ) -> np.ndarray:
  """Sequence feature (``FeatureConfig.max_sequence_length > 0``) on the TPUEmbedding CPU
  branch (``tpu_embedding_layer.py:913-919``): no combiner; entry j of
  row b goes to ``out[b, j]`` (or ``out[b, positions[p]]`` for sparse inputs), entries at
  positions >= L are dropped, the rest of ``[B, L, D]`` is zero."""
  table = np.asarray(table, dtype=np.float32)
  row_splits = np.asarray(row_splits)
  nrows = row_splits.shape[0] - 1
  out = np.zeros((nrows, max_sequence_length, table.shape[1]), dtype=np.float32)
  for b in range(nrows):
    for p in range(int(row_splits[b]), int(row_splits[b + 1])):
      pos = p - int(row_splits[b]) if positions is None else int(positions[p])
      if pos < max_sequence_length:
        out[b, pos] = table[ids[p]]
  return out


def scatter_add_grad(grad_out: np.ndarray, ids: np.ndarray, vocab: int) -> np.ndarray:
  """Backward of ``gather``: dense ``[V, D]`` gradient, duplicates summed in
  occurrence order (float32), i.e. ``UnsortedSegmentSum``."""
  g = np.zeros((vocab, grad_out.shape[-1]), dtype=np.float32)
  flat_ids = np.asarray(ids).reshape(-1)
  flat_g = np.asarray(grad_out, dtype=np.float32).reshape(-1, grad_out.shape[-1])
  np.add.at(g, flat_ids, flat_g)
  return g


def adagrad_sparse_update(table: np.ndarray, accum: np.ndarray, grad_out: np.ndarray,
                          ids: np.ndarray, lr: float, eps: float = 1e-7, legacy: bool = False
                          ) -> Tuple[np.ndarray, np.ndarray]:
  """Keras Adagrad on deduplicated IndexedSlices (README.md:84; SURVEY.md App.
  A.7, tf-keras new-style formula): for each touched row, g = sum of duplicate
  grads; acc += g*g; row -= lr * g / sqrt(acc + eps).  ``legacy``: the optimizer_v2 /
  ``ResourceApplyAdagradV2`` form of TF <= 2.10, ``/ (sqrt(acc) + eps)`` (also
  ``torch.optim.Adagrad``'s, which tests/test_oracle_golden.py checks this function against).
  PARITY UNPINNED by the reference (``models/base_test.py`` asserts metric keys only);
  ``tools/tf_reference_vectors.py`` produces the TensorFlow-side vectors of both forms."""
  table = np.array(table, dtype=np.float32)
  accum = np.array(accum, dtype=np.float32)
  g = scatter_add_grad(grad_out, ids, table.shape[0])
  touched = np.unique(np.asarray(ids).reshape(-1))
  accum[touched] = accum[touched] + g[touched] * g[touched]
  den = (np.sqrt(accum[touched]) + np.float32(eps)) if legacy else np.sqrt(accum[touched] + np.float32(eps))
  table[touched] = table[touched] - np.float32(lr) * g[touched] / den
  return table, accum
