"""Base model: the ``compute_loss`` / ``train_step`` / ``test_step`` contract.

Mirror of ``tensorflow_recommenders/models/base.py:21-104`` on ``torch.nn.Module``:
subclasses implement ``compute_loss(inputs, training=False)``; ``train_step`` runs
forward, adds regularisation losses, back-propagates and applies the optimizer, and
returns the metrics dict with the reference's keys (``loss``,
``regularization_loss``, ``total_loss`` plus ``{metric.name: result}``, :80-85).
``compile(optimizer=...)`` / ``fit`` / ``evaluate`` are thin loops around those steps.
"""

from typing import Any, Dict, Iterable, List, Optional

import torch


class Model(torch.nn.Module):

  def __init__(self):
    super().__init__()
    self.optimizer: Optional[torch.optim.Optimizer] = None

  def compute_loss(self, inputs, training: bool = False) -> torch.Tensor:   # :49-62
    raise NotImplementedError("Implementers must implement the `compute_loss` method.")

  def compile(self, optimizer: Optional[torch.optim.Optimizer] = None, **kwargs) -> None:
    self.optimizer = optimizer

  # Keras-like hooks -------------------------------------------------------------------
  @property
  def losses(self) -> List[torch.Tensor]:
    """Regularisation losses of all sub-layers (Keras ``model.losses``)."""
    out = []
    for module in self.modules():
      fn = getattr(module, "regularization_losses", None)
      if callable(fn) and module is not self:
        out.extend(fn())
    return out

  @property
  def metrics(self) -> List[Any]:
    seen, out = set(), []
    for module in self.modules():
      if module is self:
        continue
      ms = getattr(module, "metrics", None)
      if ms is None or callable(ms):
        continue
      for m in ms:
        if id(m) not in seen:
          seen.add(id(m))
          out.append(m)
    return out

  def _regularization_loss(self, like: torch.Tensor) -> torch.Tensor:
    losses = self.losses
    if not losses:
      return torch.zeros((), dtype=torch.float32, device=like.device)
    return torch.stack([l.sum() for l in losses]).sum()                # :71-73

  def _metrics_dict(self, loss, reg, total) -> Dict[str, Any]:
    out = {metric.name: metric.result() for metric in self.metrics}   # :80
    out["loss"] = loss.detach()
    out["regularization_loss"] = reg.detach()
    out["total_loss"] = total.detach()
    return out

  def train_step(self, inputs) -> Dict[str, Any]:                      # :64-85
    if self.optimizer is None:
      raise RuntimeError("Call `compile(optimizer=...)` before training.")
    self.train()
    self.optimizer.zero_grad(set_to_none=True)
    loss = self.compute_loss(inputs, training=True)
    reg = self._regularization_loss(loss)
    total = loss + reg
    total.backward()                                                   # :77
    self.optimizer.step()                                              # :78
    return self._metrics_dict(loss, reg, total)

  def test_step(self, inputs) -> Dict[str, Any]:                       # :87-104
    self.eval()
    with torch.no_grad():
      loss = self.compute_loss(inputs, training=False)
      reg = self._regularization_loss(loss)
      total = loss + reg
    return self._metrics_dict(loss, reg, total)

  def fit(self, dataset: Iterable, epochs: int = 1) -> Dict[str, List[Any]]:
    history: Dict[str, List[Any]] = {}
    for _ in range(epochs):
      for m in self.metrics:
        m.reset_states()
      logs = {}
      for batch in dataset:
        logs = self.train_step(batch)
      for k, v in logs.items():
        history.setdefault(k, []).append(float(v))
    return history

  def evaluate(self, dataset: Iterable, return_dict: bool = True):
    for m in self.metrics:
      m.reset_states()
    logs = {}
    for batch in dataset:
      logs = self.test_step(batch)
    logs = {k: float(v) for k, v in logs.items()}
    return logs if return_dict else list(logs.values())
