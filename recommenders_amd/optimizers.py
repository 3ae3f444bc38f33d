"""Optimizers for the training step of ``tfrs.Model`` on MI355X.

``Adagrad`` restates ``tf.keras.optimizers.Adagrad`` as the reference's quickstart uses it
(``README.md:84``: ``tfrs.Model.compile(optimizer=tf.keras.optimizers.Adagrad(0.5))``, applied
in ``models/base.py:77-78``):

    accumulator starts at ``initial_accumulator_value`` (0.1); per step
    ``acc += g * g ;  var -= learning_rate * g / sqrt(acc + epsilon)``   (epsilon = 1e-7)

and, like TensorFlow's handling of ``IndexedSlices`` gradients of embedding lookups, only
touches the rows that were looked up: duplicate ids are summed first, then the update is
applied once per unique row.  For parameters of :class:`recommenders_amd.layers.embedding.Embedding`
the gradient never exists as a dense ``[vocab, dim]`` tensor -- the lookup's backward hands
``(ids, grad_rows)`` slices to the optimizer, which runs the fused sort + segmented
scatter-add + Adagrad kernel (``tfrs_embedding_scatter_add_bwd`` with ``adagrad=1``).
Dense parameters (Cross kernels, MLPs) take the same formula element-wise.

The reference's tests do not pin Adagrad numerics ("parity unpinned"); the
formula above is the one of ``tf.keras.optimizers.Adagrad`` since TF 2.11 and of tf-keras
(``variable.assign_sub(lr * grad / sqrt(accumulator + epsilon))``); the test oracle restates the same
formula.  TF <= 2.10 (the reference's release script pins TF 2.9.0, ``tools/build_scripts/release.sh:6``)
runs the optimizer_v2 kernel ``ResourceApplyAdagradV2``, which divides by ``sqrt(acc) + epsilon``:
``Adagrad(..., legacy=True)`` selects that form -- it is also ``torch.optim.Adagrad``'s, which
``tests/test_ops_gpu.py`` cross-checks it against as an independent implementation;
``tools/tf_reference_vectors.py`` writes the TensorFlow-side vectors of both forms for a maintainer with
TensorFlow installed (``tests/test_tf_vectors.py`` consumes them when present).
"""

from typing import Iterable
import weakref

import torch

from recommenders_amd.layers import embedding as emb


def _adagrad_dense_multi(items, lr: float, eps: float, mode: int) -> None:
  import ctypes
  from recommenders_amd import _lib
  n = len(items)
  vp, i64a = ctypes.c_void_p * n, ctypes.c_int64 * n
  _lib.check(_lib.load().tfrs_adagrad_dense_multi(
      n, vp(*[p.data_ptr() for p, _, _ in items]), vp(*[a.data_ptr() for _, a, _ in items]),
      vp(*[g.data_ptr() for _, _, g in items]), i64a(*[p.numel() for p, _, _ in items]), float(lr), float(eps), int(mode),
      _lib.current_stream()))
  for p, _, _ in items:      # (written through raw pointers)
    torch.autograd.graph.increment_version(p)


class Adagrad(torch.optim.Optimizer):
  """``tf.keras.optimizers.Adagrad(learning_rate, initial_accumulator_value, epsilon)``."""

  def __init__(self, params: Iterable, learning_rate: float = 0.001,
               initial_accumulator_value: float = 0.1, epsilon: float = 1e-7, legacy: bool = False):
    if initial_accumulator_value < 0.0:
      raise ValueError("initial_accumulator_value must be non-negative")
    defaults = dict(learning_rate=float(learning_rate),
                    initial_accumulator_value=float(initial_accumulator_value),
                    epsilon=float(epsilon), legacy=bool(legacy))
    super().__init__(params, defaults)
    for group in self.param_groups:
      for p in group["params"]:
        if getattr(p, "_tfrs_embedding", False):
          p._tfrs_sparse_grad = True        # the lookup's backward now emits slices
          p._tfrs_slices = []
          # the LATEST optimizer built on a table owns its sparse-gradient mode: an older one
          # that is closed / collected afterwards must not switch it off
          p._tfrs_sparse_owner = weakref.ref(self)

  def _owns(self, p) -> bool:
    owner = getattr(p, "_tfrs_sparse_owner", None)
    return owner is not None and owner() is self

  def close(self) -> None:
    """Hands the embedding tables back to dense gradients: after ``close()`` (also called when the
    optimizer is garbage-collected) a lookup's backward produces an ordinary ``.grad`` again, so
    the tables can be trained by another optimizer.  Only tables this optimizer still owns are
    released: a newer ``Adagrad`` built on the same parameters keeps its sparse mode and its
    pending slices."""
    for group in self.param_groups:
      for p in group["params"]:
        if getattr(p, "_tfrs_sparse_grad", False) and self._owns(p):
          p._tfrs_sparse_grad = False
          p._tfrs_slices = []
          p._tfrs_sparse_owner = None

  def __del__(self):
    try:
      self.close()
    except Exception:   # interpreter shutdown
      pass

  def _accumulator(self, p: torch.Tensor, init: float) -> torch.Tensor:
    state = self.state[p]
    if "accumulator" not in state:
      state["accumulator"] = torch.full_like(p, init)
    return state["accumulator"]

  @torch.no_grad()
  def reset_state_(self) -> None:
    """Accumulators back to ``initial_accumulator_value`` in place (same storage)."""
    for group in self.param_groups:
      for p in group["params"]:
        acc = self.state[p].get("accumulator") if p in self.state else None
        if acc is not None:
          acc.fill_(group["initial_accumulator_value"])

  def bump_table_versions(self) -> None:
    """Version counters of every table this optimizer updates through raw pointers.  ``step()`` does it
    for the tables it touched; a HIP-graph REPLAY of a captured step runs no host code, so
    ``Model.make_graphed_train_step`` calls this after every replay -- anything keyed on the counters
    (Streaming's packed-block cache over detached views of a trained table) then sees the change."""
    for group in self.param_groups:
      for p in group["params"]:
        if getattr(p, "_tfrs_sparse_grad", False):
          torch.autograd.graph.increment_version(p)

  def zero_grad(self, set_to_none: bool = True) -> None:
    super().zero_grad(set_to_none=set_to_none)
    for group in self.param_groups:
      for p in group["params"]:
        if getattr(p, "_tfrs_sparse_grad", False):
          p._tfrs_slices.clear()

  @torch.no_grad()
  def step(self, closure=None):
    loss = None
    if closure is not None:
      with torch.enable_grad():
        loss = closure()
    for group in self.param_groups:
      lr, eps, legacy = group["learning_rate"], group["epsilon"], group.get("legacy", False)
      sparse = []     # (table, accumulator, grad rows, ids) of every looked-up table of the group
      touched = []
      for p in group["params"]:
        acc = self._accumulator(p, group["initial_accumulator_value"])
        slices = getattr(p, "_tfrs_slices", None)
        if slices:
          if len(slices) == 1:
            ids, rows = slices[0]
          else:   # the same table looked up several times: one combined IndexedSlices
            ids = torch.cat([s[0].reshape(-1) for s in slices])
            rows = torch.cat([s[1].reshape(-1, p.shape[1]) for s in slices])
          sparse.append((p.data, acc, rows, ids))
          touched.append(p)
          slices.clear()
      if sparse:
        emb.adagrad_sparse_update_multi_(sparse, lr, eps, legacy)   # small tables: one launch for all
        # the kernels wrote through raw pointers: bump the version counters so that anything
        # keyed on them (Streaming's packed-block cache over views of a table) sees the change
        for table in touched:
          torch.autograd.graph.increment_version(table)
      dense = []      # (parameter, accumulator, gradient) of every dense parameter of the group on a GPU
      for p in group["params"]:
        acc = self._accumulator(p, group["initial_accumulator_value"])
        if p.grad is not None:
          g = p.grad
          if (p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and not g.is_sparse
              and p.is_contiguous() and acc.is_contiguous()):
            dense.append((p, acc, g.contiguous()))
            continue
          acc.addcmul_(g, g)
          p.addcdiv_(g, torch.sqrt(acc) + eps if legacy else torch.sqrt(acc + eps), value=-lr)
      # every dense parameter of the group in one launch per 32 tensors (``tfrs_adagrad_dense_multi``): the four torch
      # kernels per tensor above were 72 launches of a DCN-v2 step
      for lo in range(0, len(dense), 32):
        _adagrad_dense_multi(dense[lo:lo + 32], lr, eps, 2 if legacy else 1)
    return loss
