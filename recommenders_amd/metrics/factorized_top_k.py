"""Factorized retrieval top-K metrics on MI355X.

Mirror of ``tensorflow_recommenders/metrics/factorized_top_k.py``: ``Factorized`` :27-49,
``FactorizedTopK`` :52-194 -- same constructor, ``update_state`` arguments, metric
names (``factorized_top_k/top_{k}_categorical_accuracy``), ``result`` /
``reset_states``.  The corpus sweep uses the top-K layers; the per-example hit tests
(``in_top_k`` on ``concat([positive, top_k])`` :181-192, or the id match :141-180) run
in ``tfrs_rank_of_positive`` / ``tfrs_id_match_topk``.
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

  def result(self) -> torch.Tensor:
    if self._total is None:
      return torch.tensor(0.0)
    return torch.where(self._count > 0, self._total / self._count,
                       torch.zeros_like(self._total))

  def reset_states(self) -> None:
    if self._total is not None:       # keep the storage (a captured graph may write into it)
      self._total.zero_()
      self._count.zero_()

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
    if not isinstance(candidates, topk_layers.TopK):                   # :77-81
      candidates = topk_layers.Streaming(k=max(ks)).index_from_dataset(candidates)
    self._ks = list(ks)
    self._candidates = candidates
    self._top_k_metrics = [
        Mean(name=f"{self.name}/top_{x}_categorical_accuracy") for x in ks]   # :85-89

  @property
  def metrics(self) -> List[Mean]:
    return self._top_k_metrics

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
    hits = torch.empty((len(self._ks), nq), dtype=torch.float32, device=q.device)
    if sample_weight is not None and not isinstance(sample_weight, torch.Tensor):
      sample_weight = torch.as_tensor(np.asarray(sample_weight))

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
