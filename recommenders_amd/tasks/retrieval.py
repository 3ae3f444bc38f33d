"""The factorized retrieval task on MI355X.

Mirror of ``tensorflow_recommenders/tasks/retrieval.py:29-235``: same constructor and
``call`` arguments.  The default loss -- in-batch sampled softmax, i.e. Keras
``CategoricalCrossentropy(from_logits=True, reduction=SUM)`` on ``labels = eye`` (:86-87,
:185, :210) -- runs as a fused f32-MFMA kernel pair that never materialises the
``[num_queries, num_candidates]`` logits (``tfrs_inbatch_softmax_ce_fwd/_bwd``), with
temperature (:187), sampling-probability correction (:190), accidental-hit removal
(:194-200) and ``score_mask`` (:202) folded into the logit function.

``batch_metrics`` (:228-232) of unadjusted logits are updated from rank counts
(``TopKCategoricalAccuracy.update_from_embeddings``) and ``num_hard_negatives`` (:205-208) over
plain dot-product logits from the fused top-K search (``hard_negative_softmax_loss``): neither
builds the ``[num_queries, num_candidates]`` matrix.  What still needs the explicit matrix -- a
user-supplied ``loss`` (its contract IS the matrix), multi-head (3-D) queries (:173-176), and
batch metrics / hard negatives combined with a logit adjustment that can reorder a row
(temperature for the metrics, sampling correction, accidental-hit removal, score mask) --
computes it with the HIP dense GEMM and then follows the reference's op sequence on that tensor.
"""

import os
from typing import Callable, List, Optional, Sequence, Union

import numpy as np
import torch

from recommenders_amd import _lib
from recommenders_amd import _streams
from recommenders_amd.layers import loss as loss_layers
from recommenders_amd.layers.feature_interaction.dcn import _DenseFn
from recommenders_amd.metrics import factorized_top_k as tfrs_metrics
from recommenders_amd.tasks import base

MIN_FLOAT = float(np.finfo(np.float32).min / 100.0)   # retrieval.py:26


class _InBatchSoftmaxFn(torch.autograd.Function):
  """loss = sum_b w_b (logsumexp_c S_bc - S_bb), gradients wrt both embedding matrices."""

  @staticmethod
  def forward(ctx, q, c, sample_weight, inv_t, log_corr, cand_ids, score_mask):
    lib = _lib.load()
    q = q.contiguous()
    c = c.contiguous()
    nq, d = q.shape
    nc = c.shape[0]
    ws = torch.empty((lib.tfrs_inbatch_softmax_workspace_bytes(nq, nc, d),),
                     dtype=torch.uint8, device=q.device)
    loss = torch.empty((), dtype=torch.float32, device=q.device)
    lse = torch.empty((nq,), dtype=torch.float32, device=q.device)
    pos = torch.empty((nq,), dtype=torch.float32, device=q.device)
    _lib.check(lib.tfrs_inbatch_softmax_ce_fwd(
        _lib.ptr(q), _lib.ptr(c), nq, nc, d, _lib.ptr(sample_weight), float(inv_t),
        _lib.ptr(log_corr), _lib.ptr(cand_ids), _lib.ptr(score_mask), _lib.ptr(loss),
        _lib.ptr(lse), _lib.ptr(pos), _lib.ptr(ws), ws.numel(), _lib.current_stream()))
    # the workspace keeps the forward's operand images for the backward (split-fp16 path)
    ctx.save_for_backward(q, c, sample_weight, log_corr, cand_ids, score_mask, lse, ws)
    ctx.inv_t = float(inv_t)
    return loss

  @staticmethod
  def backward(ctx, gloss):
    q, c, sample_weight, log_corr, cand_ids, score_mask, lse, ws = ctx.saved_tensors
    lib = _lib.load()
    nq, d = q.shape
    nc = c.shape[0]
    dq = torch.empty_like(q)
    dc = torch.empty_like(c)
    g = gloss.to(torch.float32).contiguous()
    _lib.check(lib.tfrs_inbatch_softmax_ce_bwd(
        _lib.ptr(q), _lib.ptr(c), nq, nc, d, _lib.ptr(sample_weight), ctx.inv_t,
        _lib.ptr(log_corr), _lib.ptr(cand_ids), _lib.ptr(score_mask), _lib.ptr(lse),
        _lib.ptr(g), _lib.ptr(dq), _lib.ptr(dc), _lib.ptr(ws), ws.numel(),
        0 if os.environ.get("TFRS_SOFTMAX_NO_REUSE") else 1,
        _lib.current_stream()))
    return dq, dc, None, None, None, None, None


def in_batch_softmax_loss(query_embeddings: torch.Tensor, candidate_embeddings: torch.Tensor,
                          sample_weight: Optional[torch.Tensor] = None,
                          temperature: Optional[float] = None,
                          candidate_sampling_probability: Optional[torch.Tensor] = None,
                          candidate_ids: Optional[torch.Tensor] = None,
                          score_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
  """Functional form of the fused default loss (retrieval.py:172-210)."""
  dev = query_embeddings.device
  inv_t = 1.0 if temperature is None else 1.0 / float(temperature)
  w = None if sample_weight is None else sample_weight.reshape(-1).to(dev, torch.float32).contiguous()
  corr = None
  if candidate_sampling_probability is not None:                      # loss.py:157-158
    corr = torch.log(torch.clamp(candidate_sampling_probability.to(dev, torch.float32),
                                 1e-6, 1.0)).contiguous()
  ids = None if candidate_ids is None else candidate_ids.reshape(-1).to(dev).long().contiguous()
  mask = None if score_mask is None else score_mask.to(dev).to(torch.uint8).contiguous()
  return _InBatchSoftmaxFn.apply(query_embeddings.to(torch.float32),
                                 candidate_embeddings.to(torch.float32), w, inv_t, corr, ids, mask)


def hard_negative_softmax_loss(query_embeddings: torch.Tensor, candidate_embeddings: torch.Tensor,
                               num_hard_negatives: int, sample_weight: Optional[torch.Tensor] = None,
                               temperature: Optional[float] = None) -> torch.Tensor:
  """``HardNegativeMining`` + softmax cross-entropy (layers/loss.py:61-111, tasks/retrieval.py:205-210)
  WITHOUT the ``[B, C]`` logits matrix, for logits that are plain (temperature-scaled) dot products: the
  reference keeps, per row, the positive and the ``num_hard_negatives`` highest negatives
  (``top_k(logits + labels * MAX_FLOAT, k + 1, sorted=False)``); here the fused top-K search of the batch's
  candidates (``top_k_of_block``: exact f32 scores, no matrix) names those rows, the lookup kernel gathers
  them (its backward is the deterministic scatter-add) and the cross-entropy runs on ``[B, k + 1]``
  logits.  Boundary ties between equal negatives pick different but equally scored rows: same loss."""
  from recommenders_amd.layers import embedding as emb_layers
  from recommenders_amd.layers import factorized_top_k as ftk
  q = query_embeddings.to(torch.float32)
  c = candidate_embeddings.to(torch.float32)
  nq, nc = q.shape[0], c.shape[0]
  num_sampled = min(int(num_hard_negatives) + 1, nc)                    # loss.py:91
  n_neg = num_sampled - 1
  pos = (q * c[:nq]).sum(dim=1, keepdim=True)                            # [B, 1]
  if n_neg > 0:
    with torch.no_grad():
      # the positive may or may not be among the best n_neg + 1 rows: search one more and drop it
      _, rows = ftk.top_k_of_block(q.detach(), c.detach(), min(n_neg + 1, nc))
      is_pos = (rows == torch.arange(nq, device=rows.device, dtype=rows.dtype)[:, None])
      order = torch.sort(is_pos.to(torch.int8), dim=1, stable=True).indices[:, :n_neg]   # negatives first, by score
      neg_rows = torch.gather(rows, 1, order).long()
    neg = (emb_layers._GatherFn.apply(c, neg_rows) * q[:, None, :]).sum(dim=2)            # [B, n_neg]
    logits = torch.cat([pos, neg], dim=1)
  else:
    logits = pos
  if temperature is not None:
    logits = logits / float(temperature)                                 # :187-188
  per_row = torch.logsumexp(logits, dim=1) - logits[:, 0]                # CE with the label on column 0
  if sample_weight is not None:
    per_row = per_row * sample_weight.reshape(-1).to(per_row.device, torch.float32)
  return per_row.sum()                                                   # Reduction.SUM (:86-87)


class _CrossReplicaConcatFn(torch.autograd.Function):
  """all-gather along dim 0 with the caller's block first; backward = sum over ranks of the
  gradients of the caller's block (a reduce-scatter)."""

  @staticmethod
  def forward(ctx, values, group):
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    values = values.contiguous()
    n = values.shape[0]
    gathered = torch.empty((world * n,) + tuple(values.shape[1:]), dtype=values.dtype,
                           device=values.device)
    dist.all_gather_into_tensor(gathered, values, group=group)   # blocks in rank order
    ctx.group, ctx.world, ctx.rank = group, world, rank
    return torch.roll(gathered, -rank * n, dims=0)

  @staticmethod
  def backward(ctx, grad):
    import torch.distributed as dist
    world, rank = ctx.world, ctx.rank
    g = torch.roll(grad.reshape((world, -1) + tuple(grad.shape[1:])), rank, dims=0).contiguous()
    out = torch.empty_like(g[0])
    if dist.get_backend(ctx.group) == "gloo":          # gloo has no reduce_scatter
      dist.all_reduce(g, group=ctx.group)
      out.copy_(g[rank])
    else:
      dist.reduce_scatter_tensor(out, g, group=ctx.group)
    return out, None


def cross_replica_concat(values: torch.Tensor, group=None) -> torch.Tensor:
  """``_cross_replica_concat`` of the reference (``tasks/retrieval.py:238-321``) on
  ``torch.distributed``: every rank contributes ``values`` ``[n, ...]`` and receives the
  concatenation of all ranks' blocks ``[world * n, ...]``, rotated so that ITS OWN block comes
  first (rank i sees blocks i, i+1, ..., world-1, 0, ..., i-1) -- so ``labels = eye(n, world*n)``
  still pairs query ``b`` with candidate ``b``.  One RCCL all-gather over xGMI forward; the
  gradient of the caller's block is the sum of every rank's gradient for it (one
  reduce-scatter), i.e. the gradient of the sum of the per-rank losses.  Without an initialised
  process group (or with a single rank) it returns ``values`` unchanged."""
  import torch.distributed as dist
  if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
    return values
  return _CrossReplicaConcatFn.apply(values, group)


class TopKCategoricalAccuracy:
  """``tf.keras.metrics.TopKCategoricalAccuracy(k)`` for ``batch_metrics``: weighted
  mean of ``in_top_k(argmax(labels), logits, k)`` (ties count for the target)."""

  def __init__(self, k: int = 5, name: str = "top_k_categorical_accuracy"):
    self.k = k
    self.name = name
    self._mean = tfrs_metrics.Mean(name)

  def update_state(self, labels, logits, sample_weight=None):
    target = torch.gather(logits, 1, torch.argmax(labels, dim=1, keepdim=True))
    greater = (logits > target).sum(dim=1)
    hit = ((greater < self.k) & torch.isfinite(target.squeeze(1))).to(torch.float32)
    self._mean.update_state(hit, sample_weight)

  def update_from_embeddings(self, q: torch.Tensor, c: torch.Tensor, sample_weight=None) -> None:
    """The same update WITHOUT the ``[B, C]`` logits matrix, for unadjusted logits ``q @ c.T``
    (labels = eye): ``in_top_k`` only asks how many logits of a row are strictly greater than the
    positive's, so the batch's candidates are swept by the rank-count kernel
    (``tfrs_rank_count_accumulate``: f32 MFMA, the d-ordered fma chain -- the positive ties exactly
    with its own column) and ``tfrs_topk_hits_update`` turns the counts into hit indicators."""
    import ctypes
    lib = _lib.load()
    nq, d = q.shape
    q = q.detach().to(torch.float32).contiguous()
    c = c.detach().to(torch.float32).contiguous()
    counts = torch.zeros((nq,), dtype=torch.int32, device=q.device)
    hits = torch.empty((1, nq), dtype=torch.float32, device=q.device)
    scratch = torch.zeros((3,), dtype=torch.float32, device=q.device)     # state[2], result[1] (unused)
    stream = _lib.current_stream()
    _lib.check(lib.tfrs_rank_count_accumulate(
        _lib.ptr(q), _lib.ptr(c), nq, d, _lib.ptr(c), None, 0, c.shape[0], c.shape[0], _lib.ptr(counts), 1,
        stream))
    ks = (ctypes.c_int32 * 1)(int(self.k))
    _lib.check(lib.tfrs_topk_hits_update(_lib.ptr(counts), nq, ks, 1, None, _lib.ptr(scratch),
                                         _lib.ptr(scratch[2:]), _lib.ptr(hits), stream))
    self._mean.update_state(hits[0], sample_weight)

  def result(self):
    return self._mean.result()

  def reset_states(self):
    self._mean.reset_states()


class _LogitsCeFn(torch.autograd.Function):
  """``sum_i w_i * (-sum_j y_ij log_softmax(S_i)_j)`` on an explicit ``[B, C]`` logits matrix
  (``tfrs_logits_ce_fwd`` / ``_bwd``): Keras ``CategoricalCrossentropy(from_logits=True, reduction=SUM)``
  (reference ``tasks/retrieval.py:86-87``) for the paths that have to build the matrix."""

  @staticmethod
  def forward(ctx, scores, labels, sample_weight):
    scores = scores.contiguous().to(torch.float32)
    labels = labels.contiguous().to(torch.float32)
    nq, nc = scores.shape
    w = None
    if sample_weight is not None:
      # one weight per row; a scalar / one-element weight broadcasts as in Keras -- anything else is an error here,
      # not an out-of-bounds read in the kernel
      w = torch.as_tensor(sample_weight).reshape(-1).to(scores.device, torch.float32)
      if w.numel() == 1:
        w = w.expand(nq)
      elif w.numel() != nq:
        raise ValueError(f"sample_weight has {w.numel()} elements for {nq} rows of logits")
      w = w.contiguous()
    rows = torch.empty((3, max(nq, 1)), dtype=torch.float32, device=scores.device)   # row loss, lse, sum of labels
    _lib.check(_lib.load().tfrs_logits_ce_fwd(_lib.ptr(scores), _lib.ptr(labels), nq, nc, _lib.ptr(w),
                                              _lib.ptr(rows[0]), _lib.ptr(rows[1]), _lib.ptr(rows[2]),
                                              _lib.current_stream()))
    ctx.save_for_backward(scores, labels, rows)
    ctx.weight = w
    return rows[0, :nq].sum()

  @staticmethod
  def backward(ctx, grad):
    scores, labels, rows = ctx.saved_tensors
    nq, nc = scores.shape
    ds = torch.empty_like(scores)
    g = grad.reshape(1).to(torch.float32).contiguous()
    _lib.check(_lib.load().tfrs_logits_ce_bwd(_lib.ptr(scores), _lib.ptr(labels), nq, nc, _lib.ptr(ctx.weight),
                                              _lib.ptr(rows[1]), _lib.ptr(rows[2]), _lib.ptr(g), _lib.ptr(ds),
                                              _lib.current_stream()))
    return ds, None, None


def logits_softmax_ce_sum(scores: torch.Tensor, labels: torch.Tensor,
                          sample_weight: Optional[torch.Tensor] = None) -> torch.Tensor:
  return _LogitsCeFn.apply(scores, labels, sample_weight)


class Retrieval(torch.nn.Module, base.Task):
  """A factorized retrieval task (reference :29-235)."""

  def __init__(self, loss: Optional[Callable] = None,
               metrics: Optional[Union[Sequence[tfrs_metrics.Factorized],
                                       tfrs_metrics.Factorized]] = None,
               batch_metrics: Optional[List] = None, loss_metrics: Optional[List] = None,
               temperature: Optional[float] = None, num_hard_negatives: Optional[int] = None,
               remove_accidental_hits: bool = False, name: Optional[str] = None,
               cross_replica_negatives: bool = False, process_group=None) -> None:
    """``cross_replica_negatives`` (not in the reference's signature, which defines
    ``_cross_replica_concat`` :238-321 but never calls it): under ``torch.distributed`` every
    rank scores its queries against the candidates of ALL ranks -- ``world`` times more
    in-batch negatives for one all-gather of ``[B, D]`` per rank.  Candidate ids and
    sampling probabilities are gathered the same way; a ``score_mask`` must already have the
    gathered width."""
    super().__init__()
    self._cross_replica_negatives = cross_replica_negatives
    self._process_group = process_group
    self.name = name or "retrieval"
    self._loss = loss            # None -> fused in-batch softmax (:86-87)
    if metrics is None:
      metrics = []
    if not isinstance(metrics, Sequence):
      metrics = [metrics]
    self._factorized_metrics = list(metrics)
    self._batch_metrics = batch_metrics or []
    self._loss_metrics = loss_metrics or []
    self._temperature = temperature
    self._num_hard_negatives = num_hard_negatives
    self._remove_accidental_hits = remove_accidental_hits

  @property
  def factorized_metrics(self):
    return self._factorized_metrics                                     # :101-106

  @factorized_metrics.setter
  def factorized_metrics(self, value) -> None:                          # :108-119
    if not isinstance(value, Sequence):
      value = []                  # (the reference's behaviour: a bare metric object clears the list)
    self._factorized_metrics = list(value)

  def __setattr__(self, name, value):
    # torch.nn.Module.__setattr__ registers a Module VALUE as a sub-module before any property setter is
    # consulted: `task.factorized_metrics = FactorizedTopK(...)` would add a child named like the property
    # and leave the list the task updates untouched.  Route the name to the setter, as Keras does.
    if name == "factorized_metrics":
      type(self).factorized_metrics.fset(self, value)
      return
    super().__setattr__(name, value)

  @property
  def metrics(self):
    """All metric objects, like Keras' ``layer.metrics``."""
    out = []
    for m in self._factorized_metrics:
      out.extend(m.metrics)
    return out + list(self._batch_metrics) + list(self._loss_metrics)

  def _logits_and_labels(self, q, c, candidate_sampling_probability, candidate_ids,
                         score_mask):
    """Explicit logits following :172-208 (only for paths that need the matrix)."""
    if q.dim() == 3:                                                    # :172-176 maxsim
      nq, heads, d = q.shape
      flat = _DenseFn.apply(q.reshape(nq * heads, d).contiguous(), c.t().contiguous(), None)
      scores = flat.reshape(nq, heads, -1).max(dim=1).values
    else:
      scores = _DenseFn.apply(q.contiguous(), c.t().contiguous(), None)  # :178-180
    nq, nc = scores.shape
    labels = torch.eye(nq, nc, dtype=torch.float32, device=scores.device)   # :185
    if self._temperature is not None:
      scores = scores / self._temperature                               # :187-188
    if candidate_sampling_probability is not None:
      scores = loss_layers.SamplingProbablityCorrection()(
          scores, candidate_sampling_probability.to(scores.device))     # :190-192
    if self._remove_accidental_hits:
      scores = loss_layers.RemoveAccidentalHits()(labels, scores,
                                                  candidate_ids.to(scores.device))  # :200
    if score_mask is not None:
      scores = torch.where(score_mask.to(scores.device).bool(), scores,
                           torch.full_like(scores, MIN_FLOAT))          # :202-203
    if self._num_hard_negatives is not None:
      scores, labels = loss_layers.HardNegativeMining(self._num_hard_negatives)(
          scores, labels)                                               # :205-208
    return scores, labels

  def forward(self, query_embeddings: torch.Tensor, candidate_embeddings: torch.Tensor,
              sample_weight: Optional[torch.Tensor] = None,
              candidate_sampling_probability: Optional[torch.Tensor] = None,
              candidate_ids: Optional[torch.Tensor] = None, compute_metrics: bool = True,
              compute_batch_metrics: bool = True,
              score_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    if self._remove_accidental_hits and candidate_ids is None:          # :194-199
      raise ValueError("When accidental hit removal is enabled, candidate ids "
                       "must be supplied.")
    q, c = query_embeddings, candidate_embeddings
    if self._cross_replica_negatives:
      c = cross_replica_concat(c, self._process_group)
      if candidate_ids is not None:
        candidate_ids = cross_replica_concat(torch.as_tensor(candidate_ids).to(c.device),
                                             self._process_group)
      if candidate_sampling_probability is not None:
        candidate_sampling_probability = cross_replica_concat(
            torch.as_tensor(candidate_sampling_probability).to(c.device), self._process_group)
    if sample_weight is not None and not isinstance(sample_weight, torch.Tensor):
      sample_weight = torch.as_tensor(np.asarray(sample_weight, dtype=np.float32))
    if sample_weight is not None:
      sample_weight = sample_weight.to(q.device)

    # (captured step: the metric branch below forks HERE, in front of the loss kernels -- _streams.mark)
    fork_point = (_streams.mark(q.device) if compute_metrics and q.dim() == 2 and self._factorized_metrics else None)

    # the fused kernels hold an embedding row in registers: dims above 128 (outside their
    # envelope; the reference accepts any dim) take the explicit-logits path below
    wide = q.dim() == 2 and q.shape[-1] > 128
    # batch metrics of unadjusted logits need a rank count per row, not the matrix (missing item 3 of
    # round 3's review): only logit adjustments that can reorder a row -- a temperature (its division
    # can merge near-ties), the sampling correction, accidental-hit removal, a score mask, hard-negative
    # mining -- or a foreign metric class keep them on the explicit-logits path
    fused_batch_metrics = (
        compute_batch_metrics and len(self._batch_metrics) > 0 and q.dim() == 2 and not wide
        and self._loss is None and self._num_hard_negatives is None and self._temperature is None
        and candidate_sampling_probability is None and not self._remove_accidental_hits
        and score_mask is None
        and all(type(m) is TopKCategoricalAccuracy for m in self._batch_metrics))
    # hard-negative mining over plain (temperature-scaled) dot products: the top-K search names the rows
    fused_hard_negatives = (
        self._num_hard_negatives is not None and self._loss is None and q.dim() == 2 and not wide
        and candidate_sampling_probability is None and not self._remove_accidental_hits
        and score_mask is None and int(self._num_hard_negatives) + 2 <= 1024
        and not (compute_batch_metrics and len(self._batch_metrics) > 0))
    need_matrix = (q.dim() == 3 or self._loss is not None
                   or (self._num_hard_negatives is not None and not fused_hard_negatives) or wide
                   or (compute_batch_metrics and len(self._batch_metrics) > 0 and not fused_batch_metrics))
    scores = labels = None
    if need_matrix:
      scores, labels = self._logits_and_labels(
          q, c, candidate_sampling_probability, candidate_ids, score_mask)

    if self._loss is None and q.dim() == 2 and self._num_hard_negatives is None and not wide:
      loss = in_batch_softmax_loss(                                     # fused :172-210
          q, c, sample_weight, self._temperature, candidate_sampling_probability,
          candidate_ids if self._remove_accidental_hits else None, score_mask)
    elif fused_hard_negatives:
      loss = hard_negative_softmax_loss(q, c, self._num_hard_negatives, sample_weight, self._temperature)
    elif self._loss is None:
      # CategoricalCrossentropy(from_logits, SUM) on the explicit logits (:86-87, :210): row-wise HIP kernels
      loss = logits_softmax_ce_sum(scores, labels, sample_weight)
    else:
      loss = self._loss(y_true=labels, y_pred=scores, sample_weight=sample_weight)

    for metric in self._loss_metrics:                                   # :213-214
      metric.update_state(loss.detach())

    if compute_metrics and q.dim() == 2:                                # :216-226
      # (a parallel branch of a captured step: the update needs the embeddings only, not the loss -- _streams.py)
      with torch.no_grad(), _streams.forked(q.device, after=fork_point):
        for metric in self._factorized_metrics:
          metric.update_state(q.detach(), c.detach()[:q.shape[0]],
                              true_candidate_ids=candidate_ids, sample_weight=sample_weight)

    if compute_batch_metrics:                                           # :228-232
      with torch.no_grad():
        for metric in self._batch_metrics:
          if fused_batch_metrics:
            metric.update_from_embeddings(q, c, sample_weight=sample_weight)
          else:
            metric.update_state(labels, scores.detach(), sample_weight=sample_weight)

    return loss

  call = forward
