"""Keras losses the ranking task uses (tf.keras.losses semantics, restated).

``BinaryCrossentropy(from_logits=False)``: probabilities are clipped to
``[1e-7, 1 - 1e-7]``, the per-sample loss is the mean over the last axis of
``-(y log p + (1 - y) log(1 - p))``; ``reduction``: ``"sum_over_batch_size"`` (Keras AUTO:
weighted sum / number of samples), ``"sum"`` or ``"none"`` (per-sample vector, as
``experimental/models/ranking.py:116-118`` asks for).  ``MeanSquaredError`` likewise.
"""

from typing import Optional

import torch

_EPS = 1e-7


class _Loss:
  def __init__(self, reduction: str = "sum_over_batch_size", name: Optional[str] = None):
    if reduction in ("auto", "AUTO"):
      reduction = "sum_over_batch_size"
    if reduction not in ("sum_over_batch_size", "sum", "none"):
      raise ValueError(f"Unknown reduction: {reduction!r}")
    self.reduction = reduction
    self.name = name

  def _per_sample(self, y_true: torch.Tensor, y_pred: torch.Tensor) -> torch.Tensor:
    raise NotImplementedError

  def __call__(self, y_true, y_pred, sample_weight=None) -> torch.Tensor:
    y_pred = y_pred.to(torch.float32)
    y_true = y_true.to(y_pred.device, torch.float32).reshape(y_pred.shape)
    per = self._per_sample(y_true, y_pred)                      # [batch, ...] last axis reduced
    if sample_weight is not None:
      w = sample_weight.to(per.device, torch.float32)
      if w.dim() == per.dim() + 1 and w.shape[-1] == 1:        # Keras squeezes a trailing 1
        w = w.squeeze(-1)
      per = per * w
    if self.reduction == "none":
      return per
    if self.reduction == "sum":
      return per.sum()
    return per.sum() / per.numel()


class BinaryCrossentropy(_Loss):
  def __init__(self, from_logits: bool = False, reduction: str = "sum_over_batch_size",
               name: Optional[str] = "binary_crossentropy"):
    super().__init__(reduction, name)
    self.from_logits = from_logits

  def _per_sample(self, y_true, y_pred):
    if self.from_logits:
      bce = torch.nn.functional.binary_cross_entropy_with_logits(y_pred, y_true, reduction="none")
    else:
      p = torch.clamp(y_pred, _EPS, 1.0 - _EPS)
      bce = -(y_true * torch.log(p) + (1.0 - y_true) * torch.log(1.0 - p))
    return bce.mean(dim=-1) if bce.dim() > 1 else bce


class MeanSquaredError(_Loss):
  def __init__(self, reduction: str = "sum_over_batch_size", name: Optional[str] = "mean_squared_error"):
    super().__init__(reduction, name)

  def _per_sample(self, y_true, y_pred):
    se = (y_pred - y_true) ** 2
    return se.mean(dim=-1) if se.dim() > 1 else se
