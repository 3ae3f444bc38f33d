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

  def _regularization_loss(self, like: torch.Tensor) -> Optional[torch.Tensor]:
    """Sum of the layers' regularisation losses (:71-73); ``None`` when there are none (the
    metrics dict then reports a cached zero and no add / fill kernels are launched)."""
    losses = self.losses
    if not losses:
      return None
    return torch.stack([l.sum() for l in losses]).sum()

  def _constant(self, value: float, like: torch.Tensor) -> torch.Tensor:
    cache = self.__dict__.setdefault("_const_cache", {})
    key = (value, like.device)
    if key not in cache:
      cache[key] = torch.full((), value, dtype=torch.float32, device=like.device)
    return cache[key]

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
    total = loss if reg is None else loss + reg
    total.backward(gradient=self._constant(1.0, total))               # :77
    self.optimizer.step()                                              # :78
    return self._metrics_dict(loss, self._constant(0.0, loss) if reg is None else reg, total)

  def make_graphed_train_step(self, example_inputs, warmup: int = 3):
    """``train_step`` captured once in a HIP graph and replayed.

    A two-tower step on MovieLens-sized batches is ~15 short kernels (gathers, the fused
    in-batch softmax forward/backward, row-scan scatter + sparse Adagrad): on MI355X the
    launch + autograd bookkeeping of the eager step costs 3-4x the kernels themselves.  The
    capture runs the very same ``train_step`` (same kernels, same order, same arithmetic) on
    static input buffers; ``step(inputs)`` copies the batch into them and replays.

    Contract: every batch must have the shapes/dtypes of ``example_inputs``; the step must be
    free of host synchronisation (``compute_metrics=False`` or tensor-only metrics; no
    ``validate_ids``); the returned dict holds static tensors that the next replay overwrites.
    Parameters and optimizer state are left exactly as they were before the call (the warm-up
    iterations needed for capture are rolled back) -- except with an optimizer whose state is
    created lazily and that has no ``reset_state_`` (plain ``torch.optim``): then the warm-up
    iterations stay applied as ordinary training steps on ``example_inputs``."""
    if self.optimizer is None:
      raise RuntimeError("Call `compile(optimizer=...)` before training.")

    def map_tensors(x, fn):
      if isinstance(x, torch.Tensor):
        return fn(x)
      if isinstance(x, dict):
        return {k: map_tensors(v, fn) for k, v in x.items()}
      if isinstance(x, (list, tuple)):
        return type(x)(map_tensors(v, fn) for v in x)
      return x

    def copy_into(dst, src):
      if isinstance(dst, torch.Tensor):
        if dst.shape != src.shape or dst.dtype != src.dtype:
          raise ValueError(f"graphed train step was captured for {tuple(dst.shape)} "
                           f"{dst.dtype}; got {tuple(src.shape)} {src.dtype}")
        dst.copy_(src, non_blocking=True)
      elif isinstance(dst, dict):
        for k in dst:
          copy_into(dst[k], src[k])
      elif isinstance(dst, (list, tuple)):
        for d, s_ in zip(dst, src):
          copy_into(d, s_)

    static_inputs = map_tensors(example_inputs, lambda t: t.detach().clone())
    params = [p for p in self.parameters()]
    saved_params = [p.detach().clone() for p in params]
    # optimizer state that already exists is snapshotted; lazily created state is re-initialised
    # through the optimizer's own ``reset_state_`` (recommenders_amd.optimizers) after capture
    had_state = [(t, t.detach().clone()) for st in self.optimizer.state.values()
                 for t in st.values() if isinstance(t, torch.Tensor)]
    can_roll_back = bool(had_state) or callable(getattr(self.optimizer, "reset_state_", None))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      for _ in range(max(warmup, 1)):
        self.train_step(static_inputs)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
      logs = self.train_step(static_inputs)
    if can_roll_back:
      with torch.no_grad():
        for p, v in zip(params, saved_params):
          p.copy_(v)
        if had_state:
          for t, v in had_state:
            t.copy_(v)
        else:
          self.optimizer.reset_state_()

    def flat_tensors(x, out):
      if isinstance(x, torch.Tensor):
        out.append(x)
      elif isinstance(x, dict):
        for k in sorted(x, key=str):     # pair tensors by key, not by insertion order
          flat_tensors(x[k], out)
      elif isinstance(x, (list, tuple)):
        for v in x:
          flat_tensors(v, out)
      return out

    static_flat = flat_tensors(static_inputs, [])

    def step(inputs):
      src_flat = flat_tensors(inputs, [])
      same_layout = len(src_flat) == len(static_flat) and all(
          s_.shape == d.shape and s_.dtype == d.dtype and s_.device == d.device
          for s_, d in zip(src_flat, static_flat))
      if same_layout and len(static_flat) > 1:
        torch._foreach_copy_(static_flat, src_flat)      # one fused copy kernel for the batch
      else:
        copy_into(static_inputs, inputs)                 # (also raises on a shape mismatch)
      graph.replay()
      return logs

    step.graph = graph
    step.static_inputs = static_inputs
    return step

  def test_step(self, inputs) -> Dict[str, Any]:                       # :87-104
    self.eval()
    with torch.no_grad():
      loss = self.compute_loss(inputs, training=False)
      reg = self._regularization_loss(loss)
      total = loss if reg is None else loss + reg
    return self._metrics_dict(loss, self._constant(0.0, loss) if reg is None else reg, total)

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
