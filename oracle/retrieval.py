"""Oracle for ``tfrs.tasks.Retrieval`` and ``layers/loss.py`` (test infra only).

Follows ``tasks/retrieval.py:172-210`` and ``layers/loss.py:26-158``.  Pinned by
``tasks/retrieval_test.py:33-71,112-137,181-213,257-298`` and the property tests
of ``layers/loss_test.py:29-130`` through ``tests/golden/retrieval_*.json``.
"""

from typing import Optional, Tuple

import numpy as np

MAX_FLOAT = np.float32(np.finfo(np.float32).max / 100.0)   # loss.py:22
MIN_FLOAT = np.float32(np.finfo(np.float32).min / 100.0)   # loss.py:23, retrieval.py:26


def sampling_probability_correction(logits, p):
  """loss.py:153-158."""
  return logits - np.log(np.clip(np.asarray(p, dtype=np.float32), 1e-6, 1.0)
                         ).astype(np.float32)


def remove_accidental_hits(labels, logits, candidate_ids):
  """loss.py:117-147."""
  ids = np.asarray(candidate_ids)
  pos_idx = np.argmax(labels, axis=1)                        # :139
  pos_ids = ids[pos_idx]                                     # :140
  dup = (pos_ids[:, None] == ids[None, :]).astype(labels.dtype)   # :142-145
  dup = dup - labels                                         # :146
  return logits + dup * MIN_FLOAT                            # :147


def hard_negative_mining(logits, labels, num_hard_negatives: int):
  """loss.py:73-111.  ``top_k(sorted=False)`` order is unspecified in TF; this
  returns the selected columns in descending-score order (CE is permutation
  invariant)."""
  num_sampled = min(num_hard_negatives + 1, logits.shape[1])  # :91
  keyed = logits + labels * MAX_FLOAT                         # :104
  cols = np.argsort(-keyed, axis=1, kind="stable")[:, :num_sampled]
  return (np.take_along_axis(logits, cols, 1),
          np.take_along_axis(labels, cols, 1))


def scores(query_embeddings, candidate_embeddings):
  """retrieval.py:172-180 (3-D queries -> max over heads, :173-176)."""
  q = np.asarray(query_embeddings, dtype=np.float32)
  c = np.asarray(candidate_embeddings, dtype=np.float32)
  if q.ndim == 3:
    return np.einsum("qne,ce->qnc", q, c).max(axis=1)
  return q @ c.T


def softmax_ce_sum(labels, logits, sample_weight=None) -> np.float32:
  """Keras ``CategoricalCrossentropy(from_logits=True, reduction=SUM)``
  (retrieval.py:86-87): per-row -sum(y * log_softmax), times
  ``sample_weight``, summed."""
  z = logits.astype(np.float64)
  z = z - z.max(axis=1, keepdims=True)
  logp = z - np.log(np.exp(z).sum(axis=1, keepdims=True))
  per_row = -(labels.astype(np.float64) * logp).sum(axis=1)
  if sample_weight is not None:
    per_row = per_row * np.asarray(sample_weight, dtype=np.float64).reshape(-1)
  return np.float32(per_row.sum())


def logits_and_labels(query_embeddings, candidate_embeddings,
                      candidate_sampling_probability=None, candidate_ids=None,
                      score_mask=None, temperature=None, num_hard_negatives=None,
                      remove_accidental_hits_flag=False
                      ) -> Tuple[np.ndarray, np.ndarray]:
  """retrieval.py:172-208: the logits/labels the loss and batch metrics see."""
  s = scores(query_embeddings, candidate_embeddings)
  nq, nc = s.shape
  labels = np.eye(nq, nc, dtype=np.float32)                  # :185
  if temperature is not None:
    s = s / np.float32(temperature)                          # :187-188
  if candidate_sampling_probability is not None:
    s = sampling_probability_correction(s, candidate_sampling_probability)  # :190-192
  if remove_accidental_hits_flag:
    if candidate_ids is None:
      raise ValueError("When accidental hit removal is enabled, candidate ids "
                       "must be supplied.")                  # :195-199
    s = remove_accidental_hits(labels, s, candidate_ids)     # :200
  if score_mask is not None:
    s = np.where(np.asarray(score_mask, dtype=bool), s, MIN_FLOAT)   # :202-203
  if num_hard_negatives is not None:
    s, labels = hard_negative_mining(s, labels, num_hard_negatives)  # :205-208
  return s.astype(np.float32), labels


def loss(query_embeddings, candidate_embeddings, sample_weight=None, **kw) -> np.float32:
  """retrieval.py:210."""
  s, labels = logits_and_labels(query_embeddings, candidate_embeddings, **kw)
  return softmax_ce_sum(labels, s, sample_weight)


def loss_grads(query_embeddings, candidate_embeddings, sample_weight=None,
               temperature=None, candidate_sampling_probability=None,
               candidate_ids=None, score_mask=None,
               remove_accidental_hits_flag=False, return_yardsticks=False):
  """Analytic gradients of :210 wrt the two embedding matrices (what
  ``tape.gradient`` returns, models/base.py:77): G = w * (softmax(S) - I);
  dQ = G C / T, dC = G^T Q / T.  float64 accumulation."""
  s, labels = logits_and_labels(
      query_embeddings, candidate_embeddings, temperature=temperature,
      candidate_sampling_probability=candidate_sampling_probability,
      candidate_ids=candidate_ids, score_mask=score_mask,
      remove_accidental_hits_flag=remove_accidental_hits_flag)
  z = s.astype(np.float64)
  z = z - z.max(axis=1, keepdims=True)
  p = np.exp(z)
  p /= p.sum(axis=1, keepdims=True)
  g = p - labels
  # Yardstick of the gradients = first-order propagation of a unit relative rounding error through
  # the formula: the TERMS of dQ / dC are w p_ij c_j and -w y_ij c_j (p_ii - 1 cancels, so p and the
  # one-hot label enter separately), and p_ij = exp(S_ij - lse_i) itself carries the ABSOLUTE error of
  # its logit and of lse_i -- a dot product's error is relative to A_ij = sum_d |q_id| |c_jd| / |T|,
  # lse_i's is the p-weighted mean of those -- i.e. a relative error A_ij + sum_j p_ij A_ij.  Measured
  # in this unit the error of any f32-grade evaluation (TensorFlow's included) is a small multiple of
  # 2^-24 .. 2^-21 whatever the magnitude of the logits.
  qa = np.abs(np.asarray(query_embeddings, dtype=np.float64))
  ca = np.abs(np.asarray(candidate_embeddings, dtype=np.float64))
  cond = qa @ ca.T / (abs(float(temperature)) if temperature is not None else 1.0)
  cond = cond + (p * cond).sum(axis=1, keepdims=True)
  ga = p * (1.0 + cond) + labels
  if sample_weight is not None:
    g = g * np.asarray(sample_weight, dtype=np.float64).reshape(-1, 1)
    ga = ga * np.abs(np.asarray(sample_weight, dtype=np.float64)).reshape(-1, 1)
  if score_mask is not None:
    g = np.where(np.asarray(score_mask, dtype=bool), g, 0.0)
    ga = np.where(np.asarray(score_mask, dtype=bool), ga, 0.0)
  if temperature is not None:
    g = g / float(temperature)
    ga = ga / abs(float(temperature))
  q = np.asarray(query_embeddings, dtype=np.float64)
  c = np.asarray(candidate_embeddings, dtype=np.float64)
  if return_yardsticks:
    # the scale floating-point errors of the gradients are measured in (see above)
    return ((g @ c).astype(np.float32), (g.T @ q).astype(np.float32),
            ga @ np.abs(c), ga.T @ np.abs(q))
  return (g @ c).astype(np.float32), (g.T @ q).astype(np.float32)


def batch_top_k_categorical_accuracy(labels, logits, k: int) -> np.ndarray:
  """Keras ``TopKCategoricalAccuracy(k)`` as used for ``batch_metrics``
  (retrieval_test.py:46-49): in_top_k(argmax(labels), logits, k) per row."""
  from oracle.metrics import in_top_k
  return in_top_k(np.argmax(labels, axis=1), logits, k).astype(np.float32)
