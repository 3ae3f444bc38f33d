"""Factorized retrieval top-K metrics on MI355X.

Mirror of ``tensorflow_recommenders/metrics/factorized_top_k.py``: ``Factorized`` :27-49,
``FactorizedTopK`` :52-194 -- same constructor, ``update_state`` arguments, metric
names (``factorized_top_k/top_{k}_categorical_accuracy``), ``result`` /
``reset_states``.  The corpus sweep uses the top-K layers; the per-example hit tests
(``in_top_k`` on ``concat([positive, top_k])`` :181-192, or the id match :141-180) run
in ``tfrs_rank_of_positive`` / ``tfrs_id_match_topk``.

Score-based updates over a raw candidate dataset (the quickstart's
``FactorizedTopK(candidates=movies.batch(128).map(item_model))``, README.md:69-71) skip the top-K
altogether: ``in_top_k`` only asks how many corpus rows score strictly above the positive, so the
sweep is ``tfrs_rank_count_accumulate`` per candidate block (ONE launch for a batched map of an
``Embedding`` tower: gather + scores + rank through the id indirection) followed by
``tfrs_topk_hits_update``, which folds the per-k weighted means into the metric state on the
device -- identical values, no sorted lists, no per-k torch reductions.
"""

import abc
import ctypes
from typing import Iterable, List, Optional, Sequence, Union

import numpy as np
import torch

from recommenders_amd import _lib
from recommenders_amd.layers import factorized_top_k as topk_layers


class Mean:
  """``tf.keras.metrics.Mean``: running weighted mean, state kept on the device."""

  def __init__(self, name: str = "mean"):
    self.name = name
    self._total = None
    self._count = None
    self._result_view = None      # written by a fused update kernel (FactorizedTopK)
    self._result_fresh = False

  def _bind(self, total: torch.Tensor, count: torch.Tensor, result: torch.Tensor) -> None:
    """State and result live in a caller-owned buffer (0-dim views): one kernel updates every
    ``Mean`` of a ``FactorizedTopK`` at once."""
    if self._total is not None:
      total.copy_(self._total)
      count.copy_(self._count)
    self._total, self._count, self._result_view = total, count, result
    self._result_fresh = False

  def update_state(self, values: torch.Tensor, sample_weight: Optional[torch.Tensor] = None):
    v = values.reshape(-1).to(torch.float32)
    if sample_weight is None:
      total, count = v.sum(), torch.full((), float(v.numel()), dtype=torch.float32, device=v.device)
    else:
      w = sample_weight.reshape(-1).to(v.device, torch.float32)
      total, count = (v * w).sum(), w.sum()
    # persistent state tensors updated IN PLACE: under HIP-graph replay (make_graphed_train_step)
    # the same storage accumulates across replays (an out-of-place `state = state + x` would
    # re-add to the captured warm-up value on every replay)
    if self._total is None:
      self._total, self._count = total.clone(), count.clone()
    else:
      self._total.add_(total)
      self._count.add_(count)
    self._result_fresh = False

  def result(self) -> torch.Tensor:
    if self._total is None:
      return torch.tensor(0.0)
    if self._result_fresh:
      # a FRESH tensor, as Keras returns: the view aliases the metric's shared device buffer, which the
      # next update_state overwrites and reset_states zeroes (logs kept per step / epoch must not change
      # afterwards).  Only under HIP-graph capture the view itself is handed out: the graphed step
      # documents that its returned tensors are overwritten by the next replay.
      if self._result_view.is_cuda and torch.cuda.is_current_stream_capturing():
        return self._result_view
      return self._result_view.clone()
    return torch.where(self._count > 0, self._total / self._count,
                       torch.zeros_like(self._total))

  def reset_states(self) -> None:
    if self._total is not None:       # keep the storage (a captured graph may write into it)
      self._total.zero_()
      self._count.zero_()
    if self._result_view is not None:
      self._result_view.zero_()

  reset_state = reset_states


class Factorized(torch.nn.Module, abc.ABC):
  """Computes metrics across top K candidates surfaced by a retrieval model (:27-49)."""

  @abc.abstractmethod
  def update_state(self, query_embeddings, true_candidate_embeddings,
                   true_candidate_ids=None):
    raise NotImplementedError()

  @property
  def metrics(self) -> List[Mean]:
    return []

  def reset_states(self) -> None:
    for metric in self.metrics:
      metric.reset_states()

  def result(self) -> List[torch.Tensor]:
    return [metric.result() for metric in self.metrics]


class FactorizedTopK(Factorized):
  """Top-K categorical accuracy over the whole candidate corpus (:52-194)."""

  def __init__(self, candidates: Union[topk_layers.TopK, Iterable],
               ks: Sequence[int] = (1, 5, 10, 50, 100), name: str = "factorized_top_k") -> None:
    super().__init__()
    self.name = name
    self._dataset = None
    if not isinstance(candidates, topk_layers.TopK):                   # :77-81
      self._dataset = candidates      # raw dataset: score-based updates sweep it with rank counts
      candidates = topk_layers.Streaming(k=max(ks)).index_from_dataset(candidates)
    self._ks = list(ks)
    self._candidates = candidates
    self._top_k_metrics = [
        Mean(name=f"{self.name}/top_{x}_categorical_accuracy") for x in ks]   # :85-89
    self._fused_state = None          # (state[2 * nks], results[nks]) on the device
    self._counts = None               # {(nq, device): uint32 [nq] rank counts}, zero between updates

  @property
  def metrics(self) -> List[Mean]:
    return self._top_k_metrics

  def reset_states(self) -> None:
    """Two launches when the metrics live in the fused state buffer (instead of three per k)."""
    if self._fused_state is not None and all(m._result_view is not None for m in self._top_k_metrics):
      self._fused_state[0].zero_()
      self._fused_state[1].zero_()
      return
    super().reset_states()

  reset_state = reset_states

  def update_state(self, query_embeddings, true_candidate_embeddings,
                   true_candidate_ids=None, sample_weight=None):
    if true_candidate_ids is None and not self._candidates.is_exact():   # :125-131
      raise ValueError(
          f"The candidate generation layer ({self._candidates}) does not return "
          "exact results. To perform evaluation using that layer, you must "
          "supply `true_candidate_ids`, which will be checked against "
          "the candidate ids returned from the candidate generation layer.")
    q = topk_layers._as_f32_matrix(query_embeddings, "query_embeddings")
    c = topk_layers._as_f32_matrix(true_candidate_embeddings, "true_candidate_embeddings")
    nq = q.shape[0]
    kmax = max(self._ks)
    lib = _lib.load()
    ks_arr = (ctypes.c_int32 * len(self._ks))(*self._ks)
    if sample_weight is not None and not isinstance(sample_weight, torch.Tensor):
      sample_weight = torch.as_tensor(np.asarray(sample_weight))
    if (true_candidate_ids is None and self._dataset is not None
        and q.shape[1] <= topk_layers.MAX_FUSED_DIM and len(self._ks) <= 16 and nq > 0):
      self._update_by_rank_counts(q, c, sample_weight)                  # :181-192 without the top-K
      return None
    hits = torch.empty((len(self._ks), nq), dtype=torch.float32, device=q.device)

    if true_candidate_ids is not None:                                  # :141-180 id based
      top_scores, rows = self._candidates._query_rows(q, kmax)
      table = self._candidates._identifier_table()
      true_codes = table.codes_of_values(
          true_candidate_ids.reshape(-1) if hasattr(true_candidate_ids, "reshape")
          else np.asarray(true_candidate_ids).reshape(-1))
      got_codes = table.codes_of_rows(rows).to(torch.int32).contiguous()
      # scores from the exact layers are finite and sorted; the NaN padding / sortedness
      # assertion of :146-161 concerns approximate (ScaNN) layers only.
      _lib.check(lib.tfrs_id_match_topk(
          _lib.ptr(got_codes), _lib.ptr(true_codes.reshape(-1).contiguous()), nq,
          got_codes.shape[1], ks_arr, len(self._ks), _lib.ptr(hits), _lib.current_stream()))
    else:                                                               # :181-192 score based
      top_scores, _ = self._candidates._query_rows(q, kmax)
      top_scores = top_scores.contiguous()
      _lib.check(lib.tfrs_rank_of_positive(
          _lib.ptr(q), _lib.ptr(c), nq, q.shape[1], _lib.ptr(top_scores),
          top_scores.shape[1], ks_arr, len(self._ks), _lib.ptr(hits), _lib.current_stream()))

    for i, metric in enumerate(self._top_k_metrics):
      metric.update_state(hits[i], sample_weight)
    return None

  # -- score-based update over a raw dataset: rank counts instead of sorted lists -----------------
  def _bound_state(self, device):
    nks = len(self._ks)
    if self._fused_state is None or self._fused_state[0].device != device:
      state = torch.zeros((2 * nks,), dtype=torch.float32, device=device)
      results = torch.zeros((nks,), dtype=torch.float32, device=device)
      for i, metric in enumerate(self._top_k_metrics):
        metric._bind(state[i], state[nks + i], results[i])
      self._fused_state = (state, results)
    return self._fused_state

  def _update_by_rank_counts(self, q, c, sample_weight) -> None:
    """``in_top_k(0, concat([pos, top_k]), k)`` is ``#{scores > pos} < k`` for k <= max(ks)
    (:181-192; the retrieved list holds the max(ks) best scores): count, do not sort."""
    from recommenders_amd import data as tfrs_data
    lib = _lib.load()
    nq, d = q.shape
    if c.shape != q.shape:
      raise ValueError("true_candidate_embeddings must have the shape of query_embeddings")
    state, results = self._bound_state(q.device)
    # one scratch buffer per (batch size, device), kept for the life of the metric: a captured step
    # (Model.fit replays one HIP graph per batch shape) holds the pointer of the buffer it was captured
    # with, so a buffer must never be dropped -- and its memory handed to someone else -- when a batch of
    # another size comes by (the ragged last batch of an epoch)
    if self._counts is None:
      self._counts = {}
    key = (nq, q.device)
    if key not in self._counts:
      self._counts[key] = torch.zeros((nq,), dtype=torch.int32, device=q.device)
    counts = self._counts[key]
    stream = _lib.current_stream()
    w = None
    if sample_weight is not None:
      w = sample_weight.reshape(-1).to(q.device, torch.float32).contiguous()
      if w.numel() != nq:
        raise ValueError("sample_weight must have one entry per query")
    ks_arr = (ctypes.c_int32 * len(self._ks))(*self._ks)
    try:
      rows = self._dataset.as_embedding_rows() if isinstance(self._dataset, tfrs_data.Dataset) else None
      if rows is not None and rows[0].shape[1] == d and rows[1].numel() > 0:
        # candidates ARE table[ids]: gather + scores + rank of the positive, ONE launch
        table, ids = rows
        ids = ids.contiguous()
        _lib.check(lib.tfrs_rank_count_accumulate(
            _lib.ptr(q), _lib.ptr(c), nq, d, _lib.ptr(table), _lib.ptr(ids),
            1 if ids.dtype == torch.int64 else 0, ids.numel(), table.shape[0], _lib.ptr(counts), 1,
            stream))
      else:
        first = 1
        for element in self._dataset:
          block = element[1] if isinstance(element, (tuple, list)) else element
          block = topk_layers._as_f32_matrix(block, "candidates")
          if block.shape[1] != d:
            raise ValueError(f"Candidate dimension {block.shape[1]} does not match queries ({d}).")
          _lib.check(lib.tfrs_rank_count_accumulate(
              _lib.ptr(q), _lib.ptr(c), nq, d, _lib.ptr(block), None, 0, block.shape[0],
              block.shape[0], _lib.ptr(counts), first, stream))
          first = 0
      _lib.check(lib.tfrs_topk_hits_update(
          _lib.ptr(counts), nq, ks_arr, len(self._ks), _lib.ptr(w), _lib.ptr(state),
          _lib.ptr(results), None, stream))
    except Exception:
      self._counts.pop(key, None)       # a half-swept count buffer must not be reused
      raise
    for metric in self._top_k_metrics:
      metric._result_fresh = True
