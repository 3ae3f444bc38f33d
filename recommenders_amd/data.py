"""A minimal stand-in for the ``tf.data.Dataset`` pipelines the reference's retrieval path is fed
with (``README.md:42-52,69-71``; ``layers/factorized_top_k.py:384-390`` "index_from_dataset";
``metrics/factorized_top_k.py:59-81`` "candidates: ... a dataset of candidate embeddings"):

    movies = Dataset.from_tensor_slices(movie_ids)
    task = tfrs.tasks.Retrieval(metrics=tfrs.metrics.FactorizedTopK(
        candidates=movies.batch(128).map(self.item_model)))

``from_tensor_slices(x).batch(n).map(fn)`` is a RE-ITERABLE of ``fn(x[lo:lo + n])`` blocks in order
(``tf.data`` semantics), evaluated lazily on every pass -- so a candidate
tower that is being trained is re-embedded with its current weights each time, as in the
reference.  The pipeline keeps its structure visible (`source`, `batch_size`, `map_fn`) so that
consumers can recognise the one shape that has a fused kernel: a batched map whose function is an
``Embedding`` layer over integer ids is "rows ``table[ids]``" and the score-based
``FactorizedTopK`` sweeps it in a single launch through the id indirection of
``tfrs_rank_count_accumulate`` -- same values as 14 gathers + 14 block updates, bit for bit.
Anything else is just iterated.  Not a data loader: no shuffling, prefetching or file formats
(out of scope).
"""

from typing import Any, Callable, Iterator, Optional

import torch


class Dataset:
  """``from_tensor_slices(x)[.batch(n)][.map(fn)]``; elements are tensors (or tuples of them)."""

  def __init__(self, source: Any, batch_size: Optional[int] = None,
               map_fn: Optional[Callable] = None) -> None:
    self.source = source
    self.batch_size = batch_size
    self.map_fn = map_fn

  @staticmethod
  def from_tensor_slices(tensors: Any) -> "Dataset":
    return Dataset(tensors)

  def batch(self, batch_size: int) -> "Dataset":
    if self.batch_size is not None or self.map_fn is not None:
      raise NotImplementedError("Dataset: only from_tensor_slices(x).batch(n).map(fn) pipelines")
    if int(batch_size) < 1:
      raise ValueError("batch_size must be positive")
    return Dataset(self.source, int(batch_size), None)

  def map(self, map_fn: Callable, num_parallel_calls: Any = None) -> "Dataset":
    if self.map_fn is not None:
      prev, nxt = self.map_fn, map_fn
      return Dataset(self.source, self.batch_size, lambda *a: nxt(prev(*a)))
    return Dataset(self.source, self.batch_size, map_fn)

  def _rows(self) -> int:
    first = self.source[0] if isinstance(self.source, (tuple, list)) else self.source
    return int(first.shape[0])

  def __len__(self) -> int:
    n = self._rows()
    return n if self.batch_size is None else -(-n // self.batch_size)

  def __iter__(self) -> Iterator:
    n = self._rows()
    step = self.batch_size or 1
    for lo in range(0, n, step):
      if isinstance(self.source, (tuple, list)):
        element = tuple(t[lo:lo + step] if self.batch_size else t[lo] for t in self.source)
      else:
        element = self.source[lo:lo + step] if self.batch_size else self.source[lo]
      if self.map_fn is not None:
        with torch.no_grad():
          element = self.map_fn(*element) if isinstance(element, tuple) else self.map_fn(element)
      yield element

  # -- structure queries used by the fused consumers -------------------------------------------
  def as_embedding_rows(self):
    """``(table, ids)`` when the elements of this pipeline are exactly ``table[ids]`` in order --
    a (batched) map of an ``Embedding`` layer over one integer id tensor on the GPU -- else
    ``None``."""
    from recommenders_amd.layers.embedding import Embedding
    fn, ids = self.map_fn, self.source
    # exactly the base layer: the fused consumer reads `fn.embeddings` and bypasses `fn.forward`, so a
    # subclass with its own forward (normalisation, scaling), forward (pre-)hooks or id validation
    # (IndexError on bad ids) must be iterated like any other map function
    if type(fn) is not Embedding or self.batch_size is None:
      return None
    if fn.validate_ids or fn._forward_hooks or fn._forward_pre_hooks:
      return None
    if not (isinstance(ids, torch.Tensor) and ids.is_cuda and ids.dim() == 1
            and ids.dtype in (torch.int32, torch.int64)):
      return None
    table = fn.embeddings
    if not (table.is_cuda and table.dtype == torch.float32 and table.is_contiguous()):
      return None
    return table, ids
