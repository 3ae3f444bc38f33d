"""Oracle for the top-K retrieval layers (test infrastructure only).

Follows ``layers/factorized_top_k.py`` of the reference:
``_take_along_axis`` :57-80, ``_exclude`` :83-115, ``TopK._compute_score``
:320-333, ``TopK.query_with_exclusions`` :242-288, ``Streaming.call`` :404-509,
``BruteForce.call`` :586-607.  Pinned by ``layers/factorized_top_k_test.py:85-147``
(grid :31-66) through ``tests/golden/topk_grid.json``.
"""

from typing import Iterable, Optional, Sequence, Tuple, Union

import numpy as np

from oracle import _clib

BATCH_TOO_SMALL = "input must have at least k columns"


def scores(queries: np.ndarray, candidates: np.ndarray) -> np.ndarray:
  """``tf.matmul(q, c, transpose_b=True)`` (:333) as a d-ordered fmaf chain."""
  q = np.ascontiguousarray(queries, dtype=np.float32)
  c = np.ascontiguousarray(candidates, dtype=np.float32)
  assert q.ndim == 2 and c.ndim == 2 and q.shape[1] == c.shape[1]
  out = np.empty((q.shape[0], c.shape[0]), dtype=np.float32)
  _clib.lib().oracle_scores_f32(
      _clib.fptr(q), _clib.fptr(c), _clib.i64(q.shape[0]), _clib.i64(c.shape[0]),
      _clib.i64(q.shape[1]), _clib.fptr(out))
  return out


def scores_f64_emulated(queries: np.ndarray, candidates: np.ndarray) -> np.ndarray:
  """Independent NumPy statement of the same chain (products exact in f64, one
  rounding to f32 per step).  Only used to cross-check the C code on small
  inputs; can differ from a true fmaf by double rounding in rare cases."""
  q = np.asarray(queries, dtype=np.float32)
  c = np.asarray(candidates, dtype=np.float32)
  acc = np.zeros((q.shape[0], c.shape[0]), dtype=np.float32)
  for k in range(q.shape[1]):
    prod = q[:, k:k + 1].astype(np.float64) * c[:, k].astype(np.float64)[None, :]
    acc = (prod + acc.astype(np.float64)).astype(np.float32)
  return acc


def top_k(values: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
  """``tf.math.top_k(values, k, sorted=True)``: descending, ties -> lower column."""
  values = np.asarray(values, dtype=np.float32)
  if values.shape[1] < k:
    raise ValueError(BATCH_TOO_SMALL)
  # stable sort of the negated row keeps lower columns first among equals and
  # treats -0.0 == +0.0, as TopKV2 does.
  order = np.argsort(-values, axis=1, kind="stable")[:, :k]
  return np.take_along_axis(values, order, axis=1), order.astype(np.int64)


def take_along_axis(arr: np.ndarray, indices: np.ndarray) -> np.ndarray:
  """:57-80 (a partial ``numpy.take_along_axis`` over axis 1)."""
  return np.take_along_axis(np.asarray(arr), np.asarray(indices), axis=1)


def exclude(scores_: np.ndarray, identifiers: np.ndarray, exclude_: np.ndarray,
            k: int) -> Tuple[np.ndarray, np.ndarray]:
  """:83-115: mask excluded ids by -1e5, re-top-k, return the ORIGINAL scores."""
  scores_ = np.asarray(scores_, dtype=np.float32)
  identifiers = np.asarray(identifiers)
  exclude_ = np.asarray(exclude_)
  isin = (identifiers[:, :, None] == exclude_[:, None, :]).any(-1)       # :101-104
  adjusted = scores_ - isin.astype(np.float32) * np.float32(1.0e5)       # :107
  k = min(k, scores_.shape[1])                                           # :109
  _, idx = top_k(adjusted, k)                                            # :111
  return take_along_axis(scores_, idx), take_along_axis(identifiers, idx)


def brute_force(queries: np.ndarray, candidates: np.ndarray, k: int,
                identifiers: Optional[np.ndarray] = None
                ) -> Tuple[np.ndarray, np.ndarray]:
  """``BruteForce.call`` :586-607 (identifiers default to ``range(n)`` :544-545)."""
  q = np.ascontiguousarray(queries, dtype=np.float32)
  c = np.ascontiguousarray(candidates, dtype=np.float32)
  if c.shape[0] < k:
    raise ValueError(BATCH_TOO_SMALL)
  vals = np.empty((q.shape[0], k), dtype=np.float32)
  idx = np.empty((q.shape[0], k), dtype=np.int64)
  _clib.lib().oracle_bruteforce_topk(
      _clib.fptr(q), _clib.fptr(c), _clib.i64(q.shape[0]), _clib.i64(c.shape[0]),
      _clib.i64(q.shape[1]), _clib.i64(k), _clib.fptr(vals), _clib.iptr(idx))
  if identifiers is None:
    return vals, idx.astype(np.int32)
  return vals, np.asarray(identifiers)[idx]                              # :607


Batch = Union[np.ndarray, Tuple[np.ndarray, np.ndarray]]


def streaming(queries: np.ndarray, batches: Iterable[Batch], k: int,
              handle_incomplete_batches: bool = True
              ) -> Tuple[np.ndarray, np.ndarray]:
  """``Streaming.call`` :404-509.

  ``batches`` yields candidate blocks ``[nb, d]`` or ``(identifiers[nb],
  candidates[nb, d])`` tuples, in dataset order (tf.data ``map`` is
  order-deterministic).  Without identifiers the ids are an
  int32 row counter (:474-485).
  """
  q = np.ascontiguousarray(queries, dtype=np.float32)
  nq = q.shape[0]
  state_s = np.zeros((nq, 0), dtype=np.float32)                          # :491-494
  state_pos = np.zeros((nq, 0), dtype=np.int64)
  all_ids = []
  counter = 0
  have_ids = None
  for batch in batches:
    if isinstance(batch, tuple):
      ids, cand = batch
      have_ids = True
    else:
      ids, cand = None, batch
      have_ids = False
    cand = np.ascontiguousarray(cand, dtype=np.float32)
    nb = cand.shape[0]
    all_ids.append(np.arange(counter, counter + nb, dtype=np.int32)
                   if ids is None else np.asarray(ids))                  # :474-480
    s = scores(q, cand)                                                  # :429
    k_ = min(k, nb) if handle_incomplete_batches else k                  # :431-434
    xs, xi = top_k(s, k_)                                                # :436
    xi = xi + counter                                                    # :438 (position == gathered id order)
    # reduce step :459-472; ties resolve to the left-most concat column.
    ls, lx = state_s.shape[1], xs.shape[1]
    lo = min(k, ls + lx) if handle_incomplete_batches else k
    if lo > ls + lx:
      raise ValueError(BATCH_TOO_SMALL)
    out_s = np.empty((nq, lo), dtype=np.float32)
    out_i = np.empty((nq, lo), dtype=np.int64)
    _clib.lib().oracle_stream_fold(
        _clib.fptr(np.ascontiguousarray(state_s)), _clib.iptr(np.ascontiguousarray(state_pos)),
        _clib.i64(ls), _clib.fptr(np.ascontiguousarray(xs)),
        _clib.iptr(np.ascontiguousarray(xi)), _clib.i64(lx), _clib.i64(nq),
        _clib.i64(lo), _clib.fptr(out_s), _clib.iptr(out_i))
    state_s, state_pos = out_s, out_i
    counter += nb
  ids_flat = (np.concatenate(all_ids) if all_ids
              else np.zeros((0,), dtype=np.int32))
  if have_ids is None:
    return state_s, np.zeros((nq, 0), dtype=np.int32)
  return state_s, ids_flat[state_pos]


def query_with_exclusions(query_fn, queries: np.ndarray, exclusions: np.ndarray,
                          k: int) -> Tuple[np.ndarray, np.ndarray]:
  """:242-288: query ``k + E`` then ``_exclude``; ``query_fn(q, k)`` is one of
  ``brute_force`` / ``streaming`` partially applied."""
  adjusted_k = k + np.asarray(exclusions).shape[1]                       # :286
  x, y = query_fn(queries, adjusted_k)                                   # :287
  return exclude(x, y, exclusions, k)                                    # :288
