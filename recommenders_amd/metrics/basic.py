"""Small Keras-style metrics used by the ranking task (state on the device, ``update_state`` /
``result`` / ``reset_states`` like ``tf.keras.metrics``)."""

from typing import Optional

import torch

from recommenders_amd.metrics.factorized_top_k import Mean


class BinaryAccuracy(Mean):
  """``tf.keras.metrics.BinaryAccuracy(threshold=0.5)``: mean of ``(y_pred > threshold) == y_true``."""

  def __init__(self, name: str = "binary_accuracy", threshold: float = 0.5):
    super().__init__(name)
    self.threshold = threshold

  def update_state(self, y_true, y_pred, sample_weight=None):
    y_pred = y_pred.to(torch.float32)
    y_true = y_true.to(y_pred.device, torch.float32).reshape(y_pred.shape)
    hit = ((y_pred > self.threshold).to(torch.float32) == y_true).to(torch.float32)
    if hit.dim() > 1:
      hit = hit.mean(dim=-1)
    super().update_state(hit, sample_weight)


class RootMeanSquaredError(Mean):
  """``tf.keras.metrics.RootMeanSquaredError``."""

  def __init__(self, name: str = "root_mean_squared_error"):
    super().__init__(name)

  def update_state(self, y_true, y_pred, sample_weight=None):
    y_pred = y_pred.to(torch.float32)
    y_true = y_true.to(y_pred.device, torch.float32).reshape(y_pred.shape)
    se = (y_pred - y_true) ** 2
    if se.dim() > 1:
      se = se.mean(dim=-1)
    super().update_state(se, sample_weight)

  def result(self) -> torch.Tensor:
    return torch.sqrt(super().result())


class AUC:
  """``tf.keras.metrics.AUC(num_thresholds=200, curve="ROC")``: confusion counts at evenly spaced
  thresholds, trapezoidal area under the (fpr, tpr) curve."""

  def __init__(self, name: str = "auc", num_thresholds: int = 200):
    self.name = name
    n = num_thresholds
    eps = 1e-7
    self._thresholds = torch.tensor([0.0 - eps] + [i / (n - 1) for i in range(1, n - 1)] + [1.0 + eps])
    self.reset_states()

  def reset_states(self) -> None:
    if getattr(self, "_tp", None) is not None:   # keep the storage (graph replay writes into it)
      for t in (self._tp, self._fp, self._tn, self._fn):
        t.zero_()
    else:
      self._tp = self._fp = self._tn = self._fn = None

  reset_state = reset_states

  def update_state(self, y_true, y_pred, sample_weight: Optional[torch.Tensor] = None):
    p = y_pred.reshape(-1).to(torch.float32)
    y = y_true.reshape(-1).to(p.device, torch.float32) > 0.5
    w = (torch.ones_like(p) if sample_weight is None
         else sample_weight.reshape(-1).to(p.device, torch.float32))
    thr = self._thresholds.to(p.device)
    pred_pos = p[None, :] > thr[:, None]                      # [T, N]
    tp = (w * (pred_pos & y[None, :])).sum(dim=1)
    fp = (w * (pred_pos & ~y[None, :])).sum(dim=1)
    fn = (w * (~pred_pos & y[None, :])).sum(dim=1)
    tn = (w * (~pred_pos & ~y[None, :])).sum(dim=1)
    if self._tp is None:
      self._tp, self._fp, self._tn, self._fn = tp.clone(), fp.clone(), tn.clone(), fn.clone()
    else:                                  # in place: the state survives HIP-graph replays
      self._tp.add_(tp)
      self._fp.add_(fp)
      self._tn.add_(tn)
      self._fn.add_(fn)

  def result(self) -> torch.Tensor:
    if self._tp is None:
      return torch.tensor(0.0)
    tpr = self._tp / torch.clamp(self._tp + self._fn, min=1e-12)
    fpr = self._fp / torch.clamp(self._fp + self._tn, min=1e-12)
    return torch.sum((fpr[:-1] - fpr[1:]) * (tpr[:-1] + tpr[1:]) / 2.0)
