"""Embedding lookup on MI355X.

``Embedding`` mirrors ``tf.keras.layers.Embedding`` as the reference uses it
(``README.md:62-66,77-78``): ``Embedding(input_dim, output_dim)``, called on integer
ids of any shape, returning ``ids.shape + [output_dim]`` float32; default initialiser
uniform(-0.05, 0.05).  ``embedding_lookup_sparse`` is the combiner lookup of the
``TPUEmbedding`` CPU branch (``layers/embedding/tpu_embedding_layer.py:913-919``) on
CSR-form ragged ids with ``sum`` / ``mean`` / ``sqrtn`` combiners.

Forward = HBM-bound gather kernel; backward = deterministic sort + segmented
scatter-add (no atomics), optionally fused with the row-wise Adagrad update.
"""

from typing import Optional

import torch

from recommenders_amd import _lib

_COMBINERS = {"sum": 0, "mean": 1, "sqrtn": 2}


def _check_device(t: torch.Tensor) -> None:
  if not t.is_cuda:
    raise RuntimeError("recommenders_amd ops need tensors on the GPU; there is no CPU fallback.")


def gather_rows(table: torch.Tensor, ids: torch.Tensor, validate: bool = False) -> torch.Tensor:
  """out[...] = table[ids[...]] through ``tfrs_embedding_gather_fwd``."""
  _check_device(table)
  table = table.contiguous()
  if ids.dtype not in (torch.int32, torch.int64):
    ids = ids.long()
  flat = ids.to(table.device).reshape(-1).contiguous()
  out = torch.empty((flat.numel(), table.shape[1]), dtype=torch.float32, device=table.device)
  err = torch.zeros((1,), dtype=torch.int32, device=table.device) if validate else None
  _lib.check(_lib.load().tfrs_embedding_gather_fwd(
      _lib.ptr(table), table.shape[0], table.shape[1], _lib.ptr(flat),
      1 if flat.dtype == torch.int64 else 0, flat.numel(), _lib.ptr(out), _lib.ptr(err),
      _lib.current_stream()))
  if validate and int(err.item()):
    raise IndexError("embedding id out of range [0, %d)" % table.shape[0])
  return out.reshape(tuple(ids.shape) + (table.shape[1],))


# vocab * n below which the sort-free row-scan kernel (one wave per table row) is used
_ROWSCAN_MAX_WORK = 1 << 26


def _use_rowscan(vocab: int, n: int, d: int) -> bool:
  return d <= 256 and vocab * max(n, 1) <= _ROWSCAN_MAX_WORK


def scatter_add_rows(grad_out: torch.Tensor, ids: torch.Tensor, vocab: int) -> torch.Tensor:
  """Dense ``[vocab, d]`` gradient of ``gather_rows`` (duplicates summed in occurrence
  order; bit-reproducible)."""
  d = grad_out.shape[-1]
  g = grad_out.reshape(-1, d).contiguous()
  if ids.dtype not in (torch.int32, torch.int64):
    ids = ids.long()
  if _use_rowscan(vocab, ids.numel(), d):
    flat = ids.reshape(-1).contiguous()
    table_grad = torch.empty((vocab, d), dtype=torch.float32, device=g.device)
    _lib.check(_lib.load().tfrs_embedding_scatter_add_rowscan(
        _lib.ptr(g), _lib.ptr(flat), 1 if flat.dtype == torch.int64 else 0, flat.numel(), d,
        vocab, _lib.ptr(table_grad), None, 0.0, 0.0, 0, _lib.current_stream()))
    return table_grad
  flat = ids.reshape(-1).contiguous()
  table_grad = torch.zeros((vocab, d), dtype=torch.float32, device=g.device)
  _scatter_unsorted(g, flat, vocab, table_grad, None, 0.0, 0.0, 0)
  return table_grad


def _scatter_unsorted(g, flat, vocab, dst, accum, lr, eps, adagrad) -> None:
  """(id, position) radix sort + segmented scatter-add / fused Adagrad in the library
  (``tfrs_embedding_scatter_add_unsorted``); ids outside ``[0, vocab)`` are ignored."""
  lib = _lib.load()
  n = flat.numel()
  ws = torch.empty((lib.tfrs_embedding_scatter_add_workspace_bytes(n),), dtype=torch.uint8,
                   device=g.device)
  _lib.check(lib.tfrs_embedding_scatter_add_unsorted(
      _lib.ptr(g), _lib.ptr(flat), 1 if flat.dtype == torch.int64 else 0, n, g.shape[-1], vocab,
      _lib.ptr(dst), _lib.ptr(accum), float(lr), float(eps), adagrad, _lib.ptr(ws), ws.numel(),
      _lib.current_stream()))


def adagrad_sparse_update_(table: torch.Tensor, accum: torch.Tensor, grad_out: torch.Tensor,
                           ids: torch.Tensor, lr: float, eps: float = 1e-7, legacy: bool = False) -> None:
  """In-place fused scatter-add + Keras Adagrad on the touched rows only
  (``models/base.py:77-78`` with ``Adagrad``, ``README.md:84``):
  g = sum of duplicate grads; acc += g*g; row -= lr * g / sqrt(acc + eps)
  (``legacy``: ``/ (sqrt(acc) + eps)``, the optimizer_v2 form of TF <= 2.10 and ``torch.optim.Adagrad``)."""
  mode = 2 if legacy else 1
  d = grad_out.shape[-1]
  g = grad_out.reshape(-1, d).contiguous()
  if ids.dtype not in (torch.int32, torch.int64):
    ids = ids.long()
  if _use_rowscan(table.shape[0], ids.numel(), d):
    flat = ids.reshape(-1).contiguous()
    _lib.check(_lib.load().tfrs_embedding_scatter_add_rowscan(
        _lib.ptr(g), _lib.ptr(flat), 1 if flat.dtype == torch.int64 else 0, flat.numel(), d,
        table.shape[0], _lib.ptr(table), _lib.ptr(accum), float(lr), float(eps), mode,
        _lib.current_stream()))
    return
  _scatter_unsorted(g, ids.reshape(-1).contiguous(), table.shape[0], table, accum, lr, eps, mode)


def adagrad_sparse_update_multi_(updates, lr: float, eps: float = 1e-7, legacy: bool = False) -> None:
  """``adagrad_sparse_update_`` for several tables of one optimizer step; ``updates`` is a list of
  ``(table, accum, grad_rows, ids)``.  The small tables (row-scan path) of the step go out in
  ONE launch (``tfrs_embedding_scatter_add_rowscan_multi``): each table's update is a chain of
  dependent latencies, so separate launches pay the chain once per table."""
  small, rest = [], []
  for table, accum, grad_out, ids in updates:
    d = grad_out.shape[-1]
    if ids.dtype not in (torch.int32, torch.int64):
      ids = ids.long()
    (small if _use_rowscan(table.shape[0], ids.numel(), d) else rest).append(
        (table, accum, grad_out.reshape(-1, d).contiguous(), ids.reshape(-1).contiguous()))
  for table, accum, g, ids in rest:
    adagrad_sparse_update_(table, accum, g, ids, lr, eps, legacy)
  for lo in range(0, len(small), 8):
    grp = small[lo:lo + 8]
    if len(grp) == 1:
      table, accum, g, ids = grp[0]
      adagrad_sparse_update_(table, accum, g, ids, lr, eps, legacy)
      continue
    import ctypes
    n = len(grp)
    vp, i64a, ia = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
    _lib.check(_lib.load().tfrs_embedding_scatter_add_rowscan_multi(
        n, vp(*[g.data_ptr() for _, _, g, _ in grp]), vp(*[i.data_ptr() for _, _, _, i in grp]),
        ia(*[1 if i.dtype == torch.int64 else 0 for _, _, _, i in grp]),
        i64a(*[i.numel() for _, _, _, i in grp]), ia(*[g.shape[-1] for _, _, g, _ in grp]),
        i64a(*[t.shape[0] for t, _, _, _ in grp]), vp(*[t.data_ptr() for t, _, _, _ in grp]),
        vp(*[a.data_ptr() for _, a, _, _ in grp]), float(lr), float(eps), 2 if legacy else 1,
        _lib.current_stream()))


def _emit_table_grad(ctx, grad_out):
  """Backward of a lookup.  Default: the dense ``[vocab, d]`` gradient.  When the table is
  owned by ``recommenders_amd.optimizers.Adagrad`` the ``(ids, grad_rows)`` slices are handed
  to the optimizer instead (TensorFlow's ``IndexedSlices``, models/base.py:77-78) and no dense
  gradient is ever built."""
  (ids,) = ctx.saved_tensors
  table = ctx.table_ref
  if getattr(table, "_tfrs_sparse_grad", False):
    table._tfrs_slices.append((ids, grad_out.contiguous()))
    return None
  return scatter_add_rows(grad_out.contiguous(), ids, ctx.vocab)


class _GatherFn(torch.autograd.Function):

  @staticmethod
  def forward(ctx, table, ids):
    ctx.save_for_backward(ids)
    ctx.vocab = table.shape[0]
    ctx.table_ref = table
    return gather_rows(table, ids)

  @staticmethod
  def backward(ctx, grad_out):
    return _emit_table_grad(ctx, grad_out), None


def embedding_lookup_sparse(table: torch.Tensor, ids: torch.Tensor, row_splits: torch.Tensor,
                            weights: Optional[torch.Tensor] = None, combiner: str = "mean",
                            validate: bool = False) -> torch.Tensor:
  """Per-row combiner over ragged ids in CSR form (row b owns
  ``ids[row_splits[b]:row_splits[b+1]]``); empty rows give zeros."""
  if combiner not in _COMBINERS:
    raise ValueError(f"combiner must be one of {sorted(_COMBINERS)}; got {combiner!r}")
  _check_device(table)
  table = table.contiguous()
  ids = ids.to(table.device).long().contiguous()
  row_splits = row_splits.to(table.device).long().contiguous()
  nrows = row_splits.numel() - 1
  w = None if weights is None else weights.to(table.device, torch.float32).contiguous()
  out = torch.empty((nrows, table.shape[1]), dtype=torch.float32, device=table.device)
  err = torch.zeros((1,), dtype=torch.int32, device=table.device) if validate else None
  _lib.check(_lib.load().tfrs_embedding_segment_reduce_fwd(
      _lib.ptr(table), table.shape[0], table.shape[1], _lib.ptr(ids), _lib.ptr(row_splits), 1,
      _lib.ptr(w), nrows, _COMBINERS[combiner], _lib.ptr(out), _lib.ptr(err),
      _lib.current_stream()))
  if validate and int(err.item()):
    raise IndexError("embedding id out of range [0, %d)" % table.shape[0])
  return out


class Embedding(torch.nn.Module):
  """``tf.keras.layers.Embedding(input_dim, output_dim)`` on HBM-resident tables."""

  def __init__(self, input_dim: int, output_dim: int, validate_ids: bool = False,
               device: Optional[torch.device] = None):
    super().__init__()
    self.input_dim = input_dim
    self.output_dim = output_dim
    self.validate_ids = validate_ids
    dev = device if device is not None else (
        torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu"))
    w = torch.empty((input_dim, output_dim), dtype=torch.float32, device=dev)
    w.uniform_(-0.05, 0.05)  # Keras "uniform" initialiser
    self.embeddings = torch.nn.Parameter(w)
    self.embeddings._tfrs_embedding = True   # lets optimizers.Adagrad ask for sliced gradients

  def forward(self, ids: torch.Tensor) -> torch.Tensor:
    if not isinstance(ids, torch.Tensor):
      ids = torch.as_tensor(ids)
    ids = ids.to(self.embeddings.device)
    if self.validate_ids:
      return _ValidatedGather.apply(self.embeddings, ids)
    return _GatherFn.apply(self.embeddings, ids)


class _ValidatedGather(_GatherFn):

  @staticmethod
  def forward(ctx, table, ids):
    ctx.save_for_backward(ids)
    ctx.vocab = table.shape[0]
    ctx.table_ref = table
    return gather_rows(table, ids, validate=True)


# the feature/table-configured front-end lives beside the raw lookups, as in the reference's
# layers/embedding package (layers/embedding/__init__.py:17)
from recommenders_amd.layers.tpu_embedding_layer import (  # noqa: E402,F401
    FeatureConfig, RaggedIds, SparseIds, TableConfig, TPUEmbedding)
