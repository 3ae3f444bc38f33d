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

from recommenders_amd import _streams


class Model(torch.nn.Module):

  def __init__(self):
    super().__init__()
    self.optimizer: Optional[torch.optim.Optimizer] = None

  def compute_loss(self, inputs, training: bool = False) -> torch.Tensor:   # :49-62
    raise NotImplementedError("Implementers must implement the `compute_loss` method.")

  def compile(self, optimizer: Optional[torch.optim.Optimizer] = None, process_group=None,
              sync_gradients: Optional[bool] = None, bucket_bytes: int = 64 << 20,
              **kwargs) -> None:
    """``sync_gradients`` (default: on whenever ``torch.distributed`` is initialised with more
    than one rank) makes ``train_step`` data-parallel the way the reference is under a
    ``tf.distribute`` strategy (``experimental/models/ranking.py:199-201``: "the default
    gradients allreduce performs sum"): after ``backward`` the gradients of all ranks are
    SUMMED before the optimizer runs, so replicas stay identical.  Dense gradients travel in
    flat buckets of ``bucket_bytes`` as reduce-scatter + all-gather (RCCL over xGMI; per-link
    bound rings like few large messages); the ``(ids, rows)`` slices of embedding lookups are
    all-gathered in rank order and handed to the fused sparse Adagrad as one IndexedSlices."""
    self.optimizer = optimizer
    self._process_group = process_group
    self._sync_gradients = sync_gradients
    self._bucket_bytes = int(bucket_bytes)
    self.invalidate_captured_steps()     # captured steps belong to the configuration they were captured under

  # data-parallel gradient exchange ------------------------------------------------------
  def _sync_world(self) -> int:
    import torch.distributed as dist
    if getattr(self, "_sync_gradients", None) is False:
      return 1
    if not (dist.is_available() and dist.is_initialized()):
      return 1
    return dist.get_world_size(getattr(self, "_process_group", None))

  def _all_reduce_gradients(self) -> None:
    """Sum of every rank's gradients, in place (see ``compile``)."""
    import torch.distributed as dist
    world = self._sync_world()
    if world == 1:
      return
    group = getattr(self, "_process_group", None)
    params = [p for p in self.parameters()
              if p.requires_grad and not getattr(p, "_tfrs_row_sharded", False)]
    # dense gradients: flat buckets in parameter order.  The bucket layout must be identical on
    # every rank, so it is built from ALL dense trainable parameters, not from those that happen
    # to have a gradient on this rank: a layer one rank did not use this step contributes zeros
    # (a rank-dependent bucket size makes reduce_scatter hang or mix parameters).
    dense = [p for p in params if not getattr(p, "_tfrs_sparse_grad", False)]
    for p in dense:
      if p.grad is None:
        p.grad = torch.zeros_like(p)
    bucket: List[torch.Tensor] = []
    size = 0

    def flush():
      nonlocal bucket, size
      if not bucket:
        return
      grads = [p.grad for p in bucket]
      n = sum(g.numel() for g in grads)
      padded = -(-n // world) * world
      flat = torch.zeros((padded,), dtype=torch.float32, device=grads[0].device)
      torch._foreach_copy_(list(flat[:n].split([g.numel() for g in grads])),
                           [g.reshape(-1).to(torch.float32) for g in grads])
      shard = torch.empty((padded // world,), dtype=torch.float32, device=flat.device)
      dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, group=group)
      dist.all_gather_into_tensor(flat, shard, group=group)
      for g, piece in zip(grads, flat[:n].split([g.numel() for g in grads])):
        g.copy_(piece.view_as(g))
      bucket, size = [], 0

    for p in dense:
      bucket.append(p)
      size += p.grad.numel() * 4
      if size >= self._bucket_bytes:
        flush()
    flush()
    # embedding slices: every rank contributes (ids, rows); concatenated in rank order so that
    # all replicas apply the same IndexedSlices in the same order (bit-identical tables)
    for p in params:
      slices = getattr(p, "_tfrs_slices", None)
      if slices is None or not getattr(p, "_tfrs_sparse_grad", False):
        continue
      d = p.shape[1]
      if slices:
        ids = torch.cat([s[0].reshape(-1).long() for s in slices])
        rows = torch.cat([s[1].reshape(-1, d) for s in slices])
      else:
        ids = torch.empty((0,), dtype=torch.int64, device=p.device)
        rows = torch.empty((0, d), dtype=torch.float32, device=p.device)
      n = torch.tensor([ids.numel()], dtype=torch.int64, device=p.device)
      counts = torch.empty((world,), dtype=torch.int64, device=p.device)
      dist.all_gather_into_tensor(counts, n, group=group)
      counts = [int(c) for c in counts.tolist()]
      m = max(counts)
      if m == 0:
        continue
      ids_pad = torch.full((m,), -1, dtype=torch.int64, device=p.device)
      rows_pad = torch.zeros((m, d), dtype=torch.float32, device=p.device)
      ids_pad[:ids.numel()] = ids
      rows_pad[:ids.numel()] = rows
      all_ids = torch.empty((world * m,), dtype=torch.int64, device=p.device)
      all_rows = torch.empty((world * m, d), dtype=torch.float32, device=p.device)
      dist.all_gather_into_tensor(all_ids, ids_pad, group=group)
      dist.all_gather_into_tensor(all_rows, rows_pad, group=group)
      if all(c == m for c in counts):
        merged = (all_ids, all_rows)
      else:
        keep = torch.cat([torch.arange(r * m, r * m + c, device=p.device) for r, c in enumerate(counts)])
        merged = (all_ids[keep], all_rows[keep])
      slices.clear()
      slices.append(merged)

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

  def _reset_metrics(self) -> None:
    """Start of an epoch (Keras resets every metric): metric CONTAINERS that keep the state of their
    metrics in one buffer (``FactorizedTopK``: fused update kernel) reset it in two launches; everything
    else is reset metric by metric."""
    done = set()
    for module in self.modules():
      # (a task keeps its FactorizedTopK objects in a plain list: they are not registered sub-modules)
      for holder in [module] + list(getattr(module, "_factorized_metrics", None) or []):
        if holder is self or getattr(holder, "_fused_state", None) is None:
          continue
        fn = getattr(holder, "reset_states", None)
        if callable(fn) and id(holder) not in done:
          fn()
          done.add(id(holder))
          done.update(id(m) for m in holder.metrics)
    for m in self.metrics:
      if id(m) not in done:
        m.reset_states()

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
    # (under capture the metric update of the task runs as a parallel branch of the graph, _streams.py; it reads the
    # embedding tables, so it is joined in front of the optimizer step that rewrites them)
    with _streams.scope() as branches:
      loss = self.compute_loss(inputs, training=True)
      reg = self._regularization_loss(loss)
      total = loss if reg is None else loss + reg
      total.backward(gradient=self._constant(1.0, total))               # :77
      self._all_reduce_gradients()     # the strategy's gradient all-reduce (sum); no-op on 1 rank
      branches.join()
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
    Parameters, optimizer state and metric state are left exactly as they were before the call
    (the warm-up iterations needed for capture are rolled back, also when the capture fails) --
    except with an optimizer whose state is created lazily and that has no ``reset_state_``
    (plain ``torch.optim``): then the warm-up iterations stay applied as ordinary training steps
    on ``example_inputs``.  ``fit`` uses this by default (see there)."""
    if self.optimizer is None:
      raise RuntimeError("Call `compile(optimizer=...)` before training.")
    if self._sync_world() > 1:
      raise RuntimeError("make_graphed_train_step does not capture the gradient exchange; use "
                         "train_step under data parallelism (or compile(sync_gradients=False)).")
    return self._make_graphed_step(self.train_step, example_inputs, warmup, training=True)

  def make_graphed_test_step(self, example_inputs, warmup: int = 1):
    """``test_step`` captured in a HIP graph (same contract as ``make_graphed_train_step``;
    nothing but the metric state is written, and that is rolled back after the capture)."""
    return self._make_graphed_step(self.test_step, example_inputs, warmup, training=False)

  def _metric_state_tensors(self) -> List[torch.Tensor]:
    """Every device tensor the metrics keep between updates (running totals, fused state buffers,
    rank-count scratch), deduplicated by storage + offset: what a capture's warm-up steps change
    besides parameters and optimizer state."""
    seen, out = set(), []
    stack = list(self.metrics)
    for module in self.modules():
      if module is not self and callable(getattr(module, "update_state", None)):
        stack.append(module)
    for obj in stack:
      for value in vars(obj).values():
        items = (value if isinstance(value, (tuple, list)) else
                 tuple(value.values()) if isinstance(value, dict) else (value,))
        for t in items:
          if isinstance(t, torch.Tensor) and not isinstance(t, torch.nn.Parameter):
            key = (t.data_ptr(), tuple(t.shape), t.dtype)
            if t.numel() and key not in seen:
              seen.add(key)
              out.append(t)
    return out

  def _make_graphed_step(self, step_fn, example_inputs, warmup: int, training: bool):
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
          raise ValueError(f"graphed step was captured for {tuple(dst.shape)} "
                           f"{dst.dtype}; got {tuple(src.shape)} {src.dtype}")
        dst.copy_(src, non_blocking=True)
      elif isinstance(dst, dict):
        for k in dst:
          copy_into(dst[k], src[k])
      elif isinstance(dst, (list, tuple)):
        for d, s_ in zip(dst, src):
          copy_into(d, s_)

    device = next(self.parameters()).device
    static_inputs = map_tensors(example_inputs, lambda t: t.detach().to(device).clone())
    params = [p for p in self.parameters()]
    saved_params = [p.detach().clone() for p in params] if training else []
    # optimizer state that already exists is snapshotted; lazily created state is re-initialised
    # through the optimizer's own ``reset_state_`` (recommenders_amd.optimizers) after capture
    had_state = []
    if training:
      had_state = [(t, t.detach().clone()) for st in self.optimizer.state.values()
                   for t in st.values() if isinstance(t, torch.Tensor)]
    can_roll_back = bool(had_state) or callable(getattr(self.optimizer, "reset_state_", None))
    metric_before = [(t, t.detach().clone()) for t in self._metric_state_tensors()]
    known = {(t.data_ptr(), tuple(t.shape), t.dtype) for t, _ in metric_before}
    # module buffers (BatchNorm running statistics, step counters) and the device's random stream are
    # advanced by the warm-up steps as well: both are put back
    buffers_before = [(b, b.detach().clone()) for b in self.buffers()]
    rng_before = torch.cuda.get_rng_state(device) if device.type == "cuda" else None

    def roll_back():
      with torch.no_grad():
        for b, v in buffers_before:
          b.copy_(v)
        if rng_before is not None:
          torch.cuda.set_rng_state(rng_before, device)
        if training and can_roll_back:
          for p, v in zip(params, saved_params):
            p.copy_(v)
          if had_state:
            for t, v in had_state:
              t.copy_(v)
          else:
            self.optimizer.reset_state_()
        for t, v in metric_before:
          t.copy_(v)
        # metric state created BY the warm-up (lazily allocated totals): back to zero, in place --
        # the captured graph accumulates into this very storage
        for t in self._metric_state_tensors():
          if (t.data_ptr(), tuple(t.shape), t.dtype) not in known:
            t.zero_()

    graph = torch.cuda.CUDAGraph()
    try:
      side = torch.cuda.Stream()
      side.wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(side):
        for _ in range(max(warmup, 1)):
          step_fn(static_inputs)
      torch.cuda.current_stream().wait_stream(side)
      # (a high-priority capture stream was measured: the replayed step takes 0.32 ms instead of 0.118 -- kept default)
      with torch.cuda.graph(graph):
        logs = step_fn(static_inputs)
    finally:
      roll_back()

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
    bump = getattr(self.optimizer, "bump_table_versions", None) if training else None
    import ctypes
    from recommenders_amd import _lib
    n_static = len(static_flat)
    fast_copy = (1 <= n_static <= 16 and all(t.is_cuda and t.is_contiguous() for t in static_flat))
    dst_ptrs = (ctypes.c_void_p * max(n_static, 1))(*[t.data_ptr() for t in static_flat])
    dst_bytes = (ctypes.c_int64 * max(n_static, 1))(*[t.numel() * t.element_size() for t in static_flat])

    def step(inputs):
      src_flat = flat_tensors(inputs, [])
      same_layout = len(src_flat) == len(static_flat) and all(
          s_.shape == d.shape and s_.dtype == d.dtype and s_.device == d.device
          for s_, d in zip(src_flat, static_flat))
      if same_layout and fast_copy and all(s_.is_contiguous() for s_ in src_flat):
        # the batch into the static buffers in ONE library launch (torch._foreach_copy_: 6 us for two 32 KB id vectors)
        n = len(src_flat)
        _lib.check(_lib.load().tfrs_copy_multi(n, dst_ptrs, (ctypes.c_void_p * n)(*[s_.data_ptr() for s_ in src_flat]),
                                               dst_bytes, _lib.current_stream()))
      elif same_layout and len(static_flat) > 1:
        torch._foreach_copy_(static_flat, src_flat)      # one fused copy kernel for the batch
      else:
        copy_into(static_inputs, inputs)                 # (also raises on a shape mismatch)
      graph.replay()
      if bump is not None:
        bump()       # the replay wrote the tables through raw pointers and ran no host code
      return logs

    step.graph = graph
    step.static_inputs = static_inputs
    return step

  def test_step(self, inputs) -> Dict[str, Any]:                       # :87-104
    self.eval()
    with torch.no_grad(), _streams.scope():
      loss = self.compute_loss(inputs, training=False)
      reg = self._regularization_loss(loss)
      total = loss if reg is None else loss + reg
    return self._metrics_dict(loss, self._constant(0.0, loss) if reg is None else reg, total)

  # -- fit / evaluate: a captured step is the default, as Keras compiles train_step ------------
  @staticmethod
  def _batch_key(batch):
    """Shape key of a batch (tensors by position / dict key); ``None`` when it holds anything
    but tensors (then the step stays eager)."""
    out = []

    def walk(x, path):
      if isinstance(x, torch.Tensor):
        out.append((path, tuple(x.shape), x.dtype))
        return True
      if isinstance(x, dict):
        return all(walk(x[k], path + (str(k),)) for k in sorted(x, key=str))
      if isinstance(x, (list, tuple)):
        return all(walk(v, path + (i,)) for i, v in enumerate(x))
      return False

    return tuple(out) if walk(batch, ()) and out else None

  def _graph_steps_allowed(self, graph: Optional[bool], training: bool) -> bool:
    """``graph=None`` (default): replay captured steps when that is known to be safe -- a ROCm device,
    no gradient exchange (a collective inside a capture is not supported here), and for training an
    optimizer whose step has no host-side state that changes from step to step: this package's
    ``Adagrad``, or a ``torch.optim`` optimizer built with ``capturable=True`` (a host-side step
    counter or learning-rate schedule would be frozen into the graph at capture time).
    ``graph=True`` forces it, ``graph=False`` / ``TFRS_FIT_GRAPH=0`` keeps every step eager."""
    import os
    if graph is False or os.environ.get("TFRS_FIT_GRAPH", "1") == "0":
      return False
    if not torch.cuda.is_available():
      return False
    first = next(self.parameters(), None)
    if first is None or not first.is_cuda:
      return False
    if training and self._sync_world() > 1:
      return False
    if graph is True or not training:
      return True
    from recommenders_amd import optimizers as own
    opt = self.optimizer
    if isinstance(opt, own.Adagrad):
      return True
    return bool(opt.param_groups) and all(g.get("capturable", False) for g in opt.param_groups)

  def invalidate_captured_steps(self) -> None:
    """Drops every step `fit` / `evaluate` captured (and the private memory pools of their graphs).
    Called by `compile`; call it yourself after changing host-side state a captured step read at capture
    time and that `_capture_fingerprint` cannot see (e.g. a Python flag your `compute_loss` branches on)."""
    for name in ("_fit_graphs", "_eval_graphs"):
      self.__dict__.pop(name, None)
    self.__dict__.pop("_capture_fp", None)

  def _capture_fingerprint(self):
    """What a captured step froze at capture time, as far as it can be seen from here: the optimizer
    object and its hyper-parameters (this package's Adagrad passes `lr` / `eps` as kernel arguments), the
    metric objects the step updates (a `task.factorized_metrics = ...` reassignment), the sub-modules.
    Objects are kept by reference (compared with `is`), so a recycled `id()` cannot alias."""
    hyper = []
    if self.optimizer is not None:
      for group in self.optimizer.param_groups:
        hyper.append(tuple(sorted((k, v) for k, v in group.items()
                                  if k != "params" and isinstance(v, (int, float, bool, str, tuple, type(None))))))
    holders = []
    for module in self.modules():
      holders.extend(getattr(module, "_factorized_metrics", None) or [])
    return (self.optimizer, tuple(hyper), tuple(self.metrics), tuple(holders), tuple(self.modules()))

  def _check_capture_fingerprint(self) -> None:
    """Start of every `fit` epoch / `evaluate` call: captured steps are dropped when
    anything in `_capture_fingerprint` changed since they were captured."""
    fp = self._capture_fingerprint()
    old = self.__dict__.get("_capture_fp")

    def same(a, b):
      if isinstance(a, tuple) and isinstance(b, tuple):
        return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
      if isinstance(a, (int, float, bool, str, type(None))) or isinstance(b, (int, float, bool, str, type(None))):
        return type(a) is type(b) and a == b
      return a is b

    if old is not None and not same(old, fp):
      self.invalidate_captured_steps()
    self.__dict__["_capture_fp"] = fp

  @staticmethod
  def _max_captured_shapes() -> int:
    """Every captured shape owns a graph with a private memory pool: ragged / sequence batches would
    grow without bound.  Shapes beyond this many stay eager (TFRS_FIT_GRAPH_MAX_SHAPES, default 8)."""
    import os
    return int(os.environ.get("TFRS_FIT_GRAPH_MAX_SHAPES", "8"))

  def _run_epoch(self, dataset: Iterable, eager_step, make_graphed, cache: dict, allowed: bool):
    """One pass over ``dataset``.  A batch shape seen for the SECOND time is captured
    (``make_graphed``) and replayed from then on; first sightings -- among them the ragged last
    batch of an epoch, 19 x 4096 + 2176 at the MovieLens shapes -- run the eager step.  A step that
    cannot be captured (host synchronisation inside it: ``validate_ids``, host-side identifiers)
    is remembered as eager-only; the failed capture's warm-up is rolled back."""
    logs = {}
    for batch in dataset:
      key = self._batch_key(batch) if allowed else None
      entry = cache.get(key) if key is not None else "eager"
      if entry is None:
        cache[key] = "seen"
        logs = eager_step(batch)
      elif entry == "seen":
        if sum(callable(v) for v in cache.values()) >= self._max_captured_shapes():
          logs = eager_step(batch)       # the cap is reached: this shape stays eager
          continue
        try:
          cache[key] = make_graphed(batch, warmup=1)   # (the shape already ran once, eagerly)
          logs = cache[key](batch)
        except (RuntimeError, ValueError) as e:
          cache[key] = "eager"
          cache.setdefault("_errors", []).append(repr(e))
          torch.cuda.synchronize()
          logs = eager_step(batch)
      elif entry == "eager":
        logs = eager_step(batch)
      else:
        logs = entry(batch)
    return logs

  def fit(self, dataset: Iterable, epochs: int = 1, graph: Optional[bool] = None) -> Dict[str, List[Any]]:
    """Keras ``Model.fit`` over a re-iterable of batches (``models/base.py:64-85`` runs under
    ``Model.fit``, which COMPILES ``train_step`` by default; ``README.md:84-98``).  Here the
    compiled form is a HIP graph per batch shape (``_run_epoch``): same kernels, order and
    arithmetic as the eager ``train_step``, bit-identical parameters and logs
    (``tests/test_ops_gpu.py::test_fit_replays_captured_steps_and_matches_eager``)."""
    history: Dict[str, List[Any]] = {}
    if self.optimizer is None:
      raise RuntimeError("Call `compile(optimizer=...)` before training.")
    allowed = self._graph_steps_allowed(graph, training=True)
    pending = []
    for _ in range(epochs):
      self._check_capture_fingerprint()  # a changed lr / optimizer / metric object drops the captured steps
      cache = self.__dict__.setdefault("_fit_graphs", {})
      self._reset_metrics()
      logs = self._run_epoch(dataset, self.train_step, self.make_graphed_train_step, cache, allowed)
      # The epoch's log values stay on the device until the last epoch has been issued (one stacked copy per epoch: the
      # logs of a replayed step are static tensors that the next epoch overwrites): reading them back here would drain
      # the stream once per epoch -- ~0.1 ms of a 2.2 ms epoch at the MovieLens shapes -- although nothing on the host
      # depends on them before `fit` returns.
      pending.append(self._logs_snapshot(logs))
    for keys, dev_keys, stacked, host_vals in pending:
      vals = dict(host_vals)
      if stacked is not None:
        vals.update(zip(dev_keys, stacked.tolist()))
      for k in keys:
        history.setdefault(k, []).append(vals[k])
    return history

  @staticmethod
  def _logs_snapshot(logs: Dict[str, Any]):
    """(keys, device keys, their values stacked into ONE new device tensor, host values): no synchronisation."""
    keys = list(logs)
    dev = [k for k in keys if isinstance(logs[k], torch.Tensor) and logs[k].is_cuda]
    stacked = torch.stack([logs[k].detach().to(torch.float32).reshape(()) for k in dev]) if dev else None
    host = {k: float(logs[k]) for k in keys if k not in dev}
    return keys, dev, stacked, host

  @staticmethod
  def _logs_to_floats(logs: Dict[str, Any]) -> Dict[str, float]:
    """The epoch's log values as Python floats with ONE device synchronisation (a `float(v)` per
    entry is a stream synchronisation + copy each: eight per epoch for the quickstart model)."""
    keys = list(logs)
    dev = [k for k in keys if isinstance(logs[k], torch.Tensor) and logs[k].is_cuda]
    out = {k: None for k in keys}
    if dev:
      vals = torch.stack([logs[k].detach().to(torch.float32).reshape(()) for k in dev]).tolist()
      out.update(zip(dev, vals))
    for k in keys:
      if out[k] is None:
        out[k] = float(logs[k])
    return out

  def evaluate(self, dataset: Iterable, return_dict: bool = True, graph: Optional[bool] = None):
    self._reset_metrics()
    allowed = self._graph_steps_allowed(graph, training=False)
    self._check_capture_fingerprint()
    cache = self.__dict__.setdefault("_eval_graphs", {})
    logs = self._run_epoch(dataset, self.test_step, self.make_graphed_test_step, cache, allowed)
    logs = self._logs_to_floats(logs)
    return logs if return_dict else list(logs.values())
