"""Top-K retrieval layers on MI355X.

Host-side mirror of ``tensorflow_recommenders/layers/factorized_top_k.py``
(``TopK`` :140-333, ``Streaming`` :336-512, ``BruteForce`` :515-610): same class
names, constructor/call arguments and error behaviour.  The arithmetic
(``tf.matmul`` + ``tf.math.top_k`` + the Streaming reduce) runs in
``libtfrs_hip.so`` -- an f32-MFMA scan with the top-K selection fused behind it.

Deviations forced by the host framework (documented in DESIGN.md):
  * tensors are ``torch.Tensor`` on a CUDA(ROCm) device; NumPy inputs are uploaded;
  * a ``tf.data.Dataset`` of candidates becomes any re-iterable of candidate blocks
    ``[nb, d]`` or ``(identifiers[nb], candidates[nb, d])`` tuples;
  * identifiers of non-numeric dtype (e.g. strings) stay on the host as NumPy arrays:
    the device returns row numbers and the final ``identifiers[idx]`` gather
    (:607, :438) is done host-side; numeric identifiers are gathered on the device;
  * ``ScaNN`` (approximate search in an external C++ library) is not provided.
"""

import abc
import ctypes
import os
from typing import Any, Callable, Dict, Iterable, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from recommenders_amd import _lib

Tensor = torch.Tensor
ArrayLike = Union[torch.Tensor, np.ndarray, Sequence]

BATCH_TOO_SMALL_MESSAGE = (
    "Tried to retrieve k={k} top items, but the candidate "
    "dataset batch size is too small. This may be because "
    "your candidate batch size is too small or the last "
    "batch of your dataset is too small. "
    "To resolve this, increase your batch size, set the "
    "drop_remainder argument to True when batching your "
    "candidates, or set the handle_incomplete_batches "
    "argument to True in the constructor. ")

NOT_INDEXED_MESSAGE = ("The `index` method must be called first to "
                       "create the retrieval index.")


def _device() -> torch.device:
  if not torch.cuda.is_available():
    raise RuntimeError(
        "recommenders_amd needs a ROCm GPU (MI355X): no device is visible and "
        "there is no CPU fallback.")
  return torch.device("cuda", torch.cuda.current_device())


def _as_f32_matrix(x: ArrayLike, what: str) -> Tensor:
  """Float32, contiguous, 2-D, on the GPU."""
  if not isinstance(x, torch.Tensor):
    x = torch.as_tensor(np.asarray(x))
  if x.dim() != 2:
    raise ValueError(f"The {what} tensor must be 2D (got {tuple(x.shape)}).")
  return x.to(device=_device(), dtype=torch.float32).contiguous()


def _workspace(nbytes: int) -> Tensor:
  return torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=_device())


_RAW_DIMS = (8, 16, 32, 64, 128)   # dims whose row-major rows the grouped Streaming path reads in place
_RAW_MAX_BLOCKS = 192              # blocks per tfrs_streaming_topk_update_blocks call (kRawMaxBlocks)
MAX_FUSED_DIM = 128    # TFRS_MAX_DIM: embedding dims the fused scan kernels keep in registers
MAX_FUSED_K = 1024     # TFRS_MAX_K: results per query the selection kernels hold in one pass
_WIDE_BLOCK = 32768    # candidate rows per materialised score block on the wide-dim path


def compute_scores(queries: Tensor, candidates: Tensor) -> Tensor:
  """``tf.matmul(queries, candidates, transpose_b=True)`` (``TopK._compute_score`` :320-333)
  through ``tfrs_compute_scores``: the candidate matrix is read in place (no transposed copy)."""
  from recommenders_amd.layers.feature_interaction import dcn
  q, c = queries.contiguous(), candidates.contiguous()
  nq, d = q.shape
  nc = c.shape[0]
  out = torch.empty((nq, nc), dtype=torch.float32, device=q.device)
  lib = _lib.load()
  f16 = 1 if dcn._use_f16_gemm(nq, nc, d) else 0
  ws = dcn._gemm_workspace(lib.tfrs_gemm_f16_workspace_bytes(nq, nc, d), q.device) if f16 else None
  _lib.check(lib.tfrs_compute_scores(_lib.ptr(q), _lib.ptr(c), nq, nc, d, _lib.ptr(out), f16,
                                     _lib.ptr(ws), ws.numel() if ws is not None else 0,
                                     _lib.current_stream()))
  return out


def _wide_topk_update(q: Tensor, block: Tensor, base_row: int, k: int, state_scores: Tensor,
                      state_rows: Tensor, state_len: int) -> int:
  """One candidate block of the wide-dim path (d > 128): materialised scores of at most
  ``_WIDE_BLOCK`` rows at a time, folded into the running state (:440-472)."""
  lib = _lib.load()
  new_len = ctypes.c_int32(state_len)
  for lo in range(0, block.shape[0], _WIDE_BLOCK):
    part = block[lo:lo + _WIDE_BLOCK]
    scores = compute_scores(q, part)
    _lib.check(lib.tfrs_topk_update_from_scores(
        _lib.ptr(scores), q.shape[0], part.shape[0], part.shape[0], base_row + lo, k,
        _lib.ptr(state_scores), _lib.ptr(state_rows), state_len, ctypes.byref(new_len),
        _lib.current_stream()))
    state_len = int(new_len.value)
  return state_len


def top_k_of_block(queries: Tensor, block: Tensor, k: int) -> Tuple[Tensor, Tensor]:
  """Exact top-``k`` (scores, row numbers) of ``queries @ block.T`` for ONE resident candidate block, with
  no index object, no host synchronisation and no ``[nq, n]`` matrix: the block is searched in place
  (``tfrs_streaming_topk_update_blocks``) when its layout allows it, else through the per-block entry
  point.  Used by ``tasks.Retrieval`` for hard-negative mining; capturable in a HIP graph."""
  q = queries.contiguous()
  block = block.contiguous()
  nq, d = q.shape
  n = block.shape[0]
  if not (1 <= k <= min(n, MAX_FUSED_K)) or d > MAX_FUSED_DIM:
    raise ValueError(f"top_k_of_block: k={k} / dim={d} outside the fused kernels' envelope")
  lib = _lib.load()
  scores = torch.zeros((nq, k), dtype=torch.float32, device=q.device)
  rows = torch.zeros((nq, k), dtype=torch.int32, device=q.device)
  new_len = ctypes.c_int32(0)
  if d in _RAW_DIMS and block.data_ptr() % 16 == 0:
    ws = _workspace(lib.tfrs_streaming_topk_blocks_workspace_bytes(nq, n, d, k))
    ptrs = (ctypes.c_void_p * 1)(block.data_ptr())
    counts = (ctypes.c_int64 * 1)(n)
    _lib.check(lib.tfrs_streaming_topk_update_blocks(
        _lib.ptr(q), nq, d, ptrs, counts, 1, 0, 0, k, _lib.ptr(scores), _lib.ptr(rows), 0,
        ctypes.byref(new_len), _lib.ptr(ws), ws.numel(), _lib.current_stream()))
  else:
    ws = _workspace(lib.tfrs_streaming_topk_workspace_bytes(nq, n, d, k))
    _lib.check(lib.tfrs_streaming_topk_update(
        _lib.ptr(q), nq, d, _lib.ptr(block), n, 0, k, _lib.ptr(scores), _lib.ptr(rows), 0,
        ctypes.byref(new_len), _lib.ptr(ws), ws.numel(), _lib.current_stream()))
  return scores, rows


def _check_candidates_with_identifiers(candidates: Iterable) -> None:
  """Precondition of the dataset used for indexing (reference :118-137), checked on
  the first element: either blocks, or 2-tuples with equal leading dimensions."""
  for first in candidates:
    if isinstance(first, (tuple, list)):
      if len(first) != 2:
        raise ValueError(
            "The dataset must yield candidate embeddings or "
            "tuples of (candidate identifiers, candidate embeddings). "
            f"Got a {len(first)}-tuple instead.")
      ids, cand = first
      if len(ids) != len(cand):
        raise ValueError(
            "Candidates and identifiers have to have the same batch dimension. "
            f"Got {len(cand)} and {len(ids)}.")
    break


class _Identifiers:
  """Identifier table: maps device row numbers to user identifiers (:607, :438) and
  user identifiers to comparable int32 codes for exclusions (:101-104)."""

  def __init__(self, values: Optional[ArrayLike], n: int):
    self.n = n
    self.host: Optional[np.ndarray] = None     # non-numeric identifiers
    self.device: Optional[Tensor] = None       # numeric identifiers
    self._code_of: Optional[Dict[Any, int]] = None
    self._codes_dev: Optional[Tensor] = None
    if values is None:
      return                                   # identifiers = arange(n), int32 (:544-545)
    if isinstance(values, torch.Tensor):
      self.device = values.to(_device())
    else:
      arr = np.asarray(values)
      if arr.dtype.kind in "iufb":
        self.device = torch.as_tensor(arr).to(_device())
      else:
        self.host = arr

  @property
  def is_range(self) -> bool:
    return self.host is None and self.device is None

  def gather(self, idx: Tensor):
    """identifiers[idx] for an int32 index tensor."""
    if self.is_range:
      return idx
    if self.device is not None:
      return self.device[idx.long()]
    return self.host[idx.cpu().numpy()]

  def _build_codes(self) -> None:
    if self._code_of is not None:
      return
    vals = self.host if self.host is not None else self.device.cpu().numpy()
    uniq, inverse = np.unique(vals, return_inverse=True)
    self._uniq_host = uniq
    self._uniq_dev = (torch.as_tensor(uniq).to(_device())
                      if self.device is not None else None)
    self._code_of = {v.item() if hasattr(v, "item") else v: i for i, v in enumerate(uniq)}
    self._codes_dev = torch.as_tensor(inverse.astype(np.int32)).to(_device())

  def codes_of_rows(self, idx: Tensor) -> Tensor:
    """int32 code (rank among the distinct identifiers) of each retrieved row."""
    if self.is_range:
      return idx
    self._build_codes()
    return self._codes_dev[idx.long()].contiguous()

  def codes_of_values(self, values: ArrayLike) -> Tensor:
    """int32 codes of user-supplied identifiers (-1 = not in the index)."""
    if isinstance(values, torch.Tensor):
      values = values.cpu().numpy()
    arr = np.asarray(values)
    if self.is_range:
      codes = np.where((arr >= 0) & (arr < self.n), arr, -1).astype(np.int32)
    else:
      self._build_codes()
      flat = [self._code_of.get(v.item() if hasattr(v, "item") else v, -1)
              for v in arr.reshape(-1)]
      codes = np.asarray(flat, dtype=np.int32).reshape(arr.shape)
    return torch.as_tensor(codes).to(_device()).contiguous()

  def values_of_codes(self, codes: Tensor):
    if self.is_range:
      return codes
    if self._uniq_dev is not None:
      return self._uniq_dev[codes.long()]
    return self._uniq_host[codes.cpu().numpy()]


def _exclude(scores: Tensor, row_idx: Tensor, identifiers: _Identifiers,
             exclude: ArrayLike, k: int):
  """``_exclude`` (:83-115) through ``tfrs_topk_exclude``: candidates whose
  identifier is in the query's exclusion row are pushed down by 1e5, the top-k is
  re-taken, and the ORIGINAL scores / identifiers of the winners are returned.
  Identifiers are compared through int32 codes (rank among distinct identifiers)."""
  nq, kin = scores.shape
  codes = identifiers.codes_of_rows(row_idx).to(torch.int32).contiguous()
  excl = identifiers.codes_of_values(exclude)
  if excl.dim() != 2 or excl.shape[0] != nq:
    raise ValueError(
        f"exclusions must be [num_queries, num_to_exclude]; got {tuple(excl.shape)}")
  kout = min(k, kin)
  out_scores = torch.empty((nq, kout), dtype=torch.float32, device=scores.device)
  out_codes = torch.empty((nq, kout), dtype=torch.int32, device=scores.device)
  scores = scores.contiguous()
  _lib.check(_lib.load().tfrs_topk_exclude(
      _lib.ptr(scores), _lib.ptr(codes), nq, kin, _lib.ptr(excl), excl.shape[1], k,
      _lib.ptr(out_scores), _lib.ptr(out_codes), _lib.current_stream()))
  return out_scores, identifiers.values_of_codes(out_codes)


class TopK(torch.nn.Module, abc.ABC):
  """Interface for top K layers (reference :140-333).

  Implementers provide ``index`` (build the retrieval index from a candidate
  matrix) and ``call`` (top K candidates for a batch of queries).
  """

  def __init__(self, k: int, **kwargs) -> None:
    name = kwargs.pop("name", None)
    super().__init__()
    self.name = name if name is not None else type(self).__name__.lower()
    self._k = k

  @abc.abstractmethod
  def index(self, candidates: ArrayLike, identifiers: Optional[ArrayLike] = None) -> "TopK":
    """Builds the retrieval index; an existing index is dropped (:158-177)."""
    raise NotImplementedError()

  def index_from_dataset(self, candidates: Iterable) -> "TopK":
    """Builds the index from an iterable of candidate blocks or (identifier block,
    candidate block) pairs (:179-215)."""
    _check_candidates_with_identifiers(candidates)
    blocks, ids = [], []
    has_ids = None
    for element in candidates:
      if isinstance(element, (tuple, list)):
        i, c = element
        has_ids = True
        ids.append(i.cpu().numpy() if isinstance(i, torch.Tensor) else np.asarray(i))
      else:
        c = element
        has_ids = False
      blocks.append(_as_f32_matrix(c, "candidates"))
    if not blocks:
      raise ValueError("The candidate dataset is empty.")
    cand = torch.cat(blocks, dim=0)
    return self.index(cand, np.concatenate(ids, axis=0) if has_ids else None)

  @abc.abstractmethod
  def call(self, queries, k: Optional[int] = None):
    """Returns (top scores [B, k], top identifiers [B, k]) (:217-240)."""
    raise NotImplementedError()

  def forward(self, queries, k: Optional[int] = None):
    return self.call(queries, k=k)

  def query_with_exclusions(self, queries, exclusions: ArrayLike, k: Optional[int] = None):
    """Top-k with per-query excluded identifiers (:242-288): query ``k + E``, then
    ``_exclude``."""
    k = k if k is not None else self._k
    num_excl = (exclusions.shape[1] if hasattr(exclusions, "shape")
                else np.asarray(exclusions).shape[1])
    adjusted_k = k + num_excl                                         # :286
    scores, rows = self._query_rows(queries, adjusted_k)               # :287
    return _exclude(scores, rows, self._identifier_table(), exclusions, k)   # :288

  @abc.abstractmethod
  def is_exact(self) -> bool:
    raise NotImplementedError()

  # -- implementation hooks -------------------------------------------------------------
  @abc.abstractmethod
  def _query_rows(self, queries, k: int) -> Tuple[Tensor, Tensor]:
    """(scores, int32 row numbers) before the identifier gather."""

  @abc.abstractmethod
  def _identifier_table(self) -> _Identifiers:
    pass

  def _embed(self, queries) -> Tensor:
    query_model = getattr(self, "query_model", None)
    if query_model is not None:
      queries = query_model(queries)
    return _as_f32_matrix(queries, "queries")


_APPEND_CHUNK_ROWS = 1 << 20   # rows gathered before an index append (BruteForce.index_from_dataset)


class _IndexHandle:
  """RAII wrapper of ``tfrs_index_t``."""

  def __init__(self):
    self._lib = _lib.load()
    h = ctypes.c_void_p()
    _lib.check(self._lib.tfrs_index_create(ctypes.byref(h)))
    self.handle = h

  def __del__(self):
    h, self.handle = getattr(self, "handle", None), None
    if h:
      try:
        self._lib.tfrs_index_destroy(h)
      except Exception:  # interpreter shutdown
        pass


class _Duplicates:
  """Bookkeeping of a de-duplicated index: ``start[u] .. start[u + 1]`` delimits, in ``rows``, the
  ascending original row numbers that distinct row ``u`` stands for; ``distinct_of_row[r]`` is the
  distinct row of original row ``r`` (to rebuild the corpus)."""

  def __init__(self, start: Tensor, rows: Tensor, distinct_of_row: Tensor, max_multiplicity: int):
    self.start, self.rows, self.distinct_of_row = start, rows, distinct_of_row
    self.count = int(start.numel() - 1)
    self.max_multiplicity = max_multiplicity


def _find_duplicates(cand: Tensor, min_multiplicity: int = 16, min_fraction: float = 0.1):
  """Groups bit-identical rows of ``cand`` (64-bit row hash from ``tfrs_row_hash64``, stable sort,
  exact comparison of hash neighbours -- a hash collision only costs a missed merge).  Returns
  ``(canonical_rows, _Duplicates)`` when de-duplication pays -- some row occurs at least
  ``min_multiplicity`` times or at least ``min_fraction`` of the rows are copies -- else ``None``.
  Index-time work: one pass over the rows, one sort of n 64-bit keys."""
  n, d = cand.shape
  if n < 2:
    return None
  lib = _lib.load()
  h = torch.empty((n,), dtype=torch.int64, device=cand.device)
  _lib.check(lib.tfrs_row_hash64(_lib.ptr(cand), n, d, _lib.ptr(h), _lib.current_stream()))
  hs, perm = torch.sort(h, stable=True)            # equal hashes: ascending original row
  same_hash = hs[1:] == hs[:-1]
  if not bool(same_hash.any()):
    return None
  # exact comparison, only where neighbouring hashes agree
  pos = torch.nonzero(same_hash).reshape(-1) + 1
  eq = torch.zeros((n,), dtype=torch.bool, device=cand.device)
  for lo in range(0, pos.numel(), 1 << 20):        # bounded temporaries
    p = pos[lo:lo + (1 << 20)]
    a, b = cand.index_select(0, perm[p]), cand.index_select(0, perm[p - 1])
    eq[p] = (a.view(torch.int32) == b.view(torch.int32)).all(dim=1)
  new_group = ~eq                                   # position 0 starts a group
  group = torch.cumsum(new_group.to(torch.int64), 0) - 1          # group of every sorted position
  n_groups = int(group[-1].item()) + 1
  sizes = torch.bincount(group, minlength=n_groups)
  max_mult = int(sizes.max().item())
  if n_groups == n or (max_mult < min_multiplicity and n - n_groups < min_fraction * n):
    return None
  # canonical (lowest) original row of every group; distinct rows are numbered by ascending
  # canonical row so that the distinct index keeps the corpus order
  first_pos = torch.nonzero(new_group).reshape(-1)
  canon = perm[first_pos]
  canon_sorted, by_canon = torch.sort(canon)
  rank_of_group = torch.empty_like(by_canon)
  rank_of_group[by_canon] = torch.arange(n_groups, device=cand.device)
  distinct_of_sorted = rank_of_group[group]
  distinct_of_row = torch.empty((n,), dtype=torch.int64, device=cand.device)
  distinct_of_row[perm] = distinct_of_sorted
  # rows grouped by distinct row, ascending within a group (stable sort of 0..n-1 by distinct row)
  _, rows = torch.sort(distinct_of_row, stable=True)
  counts = torch.bincount(distinct_of_row, minlength=n_groups)
  start = torch.zeros((n_groups + 1,), dtype=torch.int64, device=cand.device)
  start[1:] = torch.cumsum(counts, 0)
  return canon_sorted, _Duplicates(start.contiguous(), rows.to(torch.int32).contiguous(),
                                   distinct_of_row.to(torch.int32).contiguous(), max_mult)


class _DeferredFinite:
  """Deferred finiteness record for the searches that have no index handle (``Streaming`` over blocks read in
  place): ``note`` ORs "some element is NaN / Inf" into the host-visible flag word of an otherwise empty library
  handle -- ONE tiny launch, no synchronisation (``tfrs_index_note_nonfinite``; the first version of this record was
  ten torch kernels, 50 us of a 1.4 ms single-query call); ``check`` (at the next call) raises if a completed call set
  it -- the same deferred contract as ``BruteForce``'s flag word."""

  def __init__(self) -> None:
    self._lib = _lib.load()
    self._handle = ctypes.c_void_p()
    _lib.check(self._lib.tfrs_index_create(ctypes.byref(self._handle)))

  def __del__(self):
    try:
      if self._handle:
        self._lib.tfrs_index_destroy(self._handle)
        self._handle = None
    except Exception:   # interpreter shutdown
      pass

  def note(self, *tensors: Tensor) -> None:
    if torch.cuda.is_current_stream_capturing():
      return                     # (a replayed graph runs no host code: nothing could read the flag)
    flat = [t.contiguous() for t in tensors if t.numel() > 0]
    for i in range(0, len(flat), 2):
      x, y = flat[i], (flat[i + 1] if i + 1 < len(flat) else None)
      _lib.check(self._lib.tfrs_index_note_nonfinite(
          self._handle, _lib.ptr(x), x.numel(), _lib.ptr(y), y.numel() if y is not None else 0, 2,
          _lib.current_stream()))

  def check(self, what: str) -> None:
    out = ctypes.c_int32(0)
    _lib.check(self._lib.tfrs_index_nonfinite(self._handle, 2, ctypes.byref(out)))
    if out.value & 2:
      raise ValueError(f"{what}: the queries or the best scores of an earlier call contained NaN or Inf; queries and "
                       "candidate blocks must be finite (include/tfrs_hip.h).")


def _raise_if_nonfinite_candidates(handle) -> None:
  """After the packer has run and the stream was synchronised: bit 0 of the handle's flag word."""
  out = ctypes.c_int32(0)
  _lib.check(handle._lib.tfrs_index_nonfinite(handle.handle, 0, ctypes.byref(out)))
  if out.value & 1:
    raise ValueError("The candidates contain NaN or Inf: the fp16-prefiltered search needs finite candidate rows "
                     "(its error bound is built from row norms; include/tfrs_hip.h).  Clean the embeddings -- a "
                     "diverged training run is the usual source -- before indexing them.")


class BruteForce(TopK):
  """Brute force retrieval (reference :515-610): exact top-K of ``q @ candidates^T``.

  ``index`` copies the candidates into a layer-owned, MFMA-friendly packed corpus in
  HBM (:559-584); ``call`` is one fused scan, the ``[B, N]`` score matrix is never
  materialised.

  ``dedup`` (default ``"auto"``; not in the reference): ``tf.math.top_k`` breaks ties by the lower
  index (:605), so on a corpus with many EXACT copies of a row (default / cold-start embeddings,
  popularity-weighted duplicates) every copy of a top-K row is a candidate for the K-th place, no
  score threshold separates them and the filtered scans degrade to their exact-recompute path
  (68x slower at BASELINE configs[1] shapes on a Zipf-duplicated corpus).  ``index`` therefore
  looks for bit-identical rows and, when some row occurs >= 16 times or >= 10 % of the rows are
  copies, indexes the DISTINCT rows only; a call searches those and expands the best of them back
  into the exact top-K of the original corpus (``tfrs_topk_expand_duplicates``: same scores, same
  row order as the full search).  ``False`` switches the detection off, ``True`` forces it for any
  duplicate.
  """

  def __init__(self, query_model: Optional[Callable] = None, k: int = 10,
               name: Optional[str] = None, dedup="auto", check_finite: bool = False):
    """``check_finite`` (not in the reference's signature): candidates and queries must be finite -- the reference's
    ``tf.math.top_k`` (:605) tolerates NaN / Inf scores, the fp16-prefiltered search here does not (include/tfrs_hip.h,
    ``tfrs_index_nonfinite``).  Non-finite CANDIDATES always raise ``ValueError`` from ``index`` /
    ``index_from_dataset``.  Non-finite QUERIES are recorded by the search kernels without a host synchronisation:
    only their own result rows are affected, and the ``ValueError`` is raised by the NEXT call (deferred, like an
    asynchronous device error); ``check_finite=True`` synchronises after every call and raises at once."""
    super().__init__(k=k, name=name)
    self.query_model = query_model
    self._check_finite = bool(check_finite)
    self._index: Optional[_IndexHandle] = None
    self._ids: Optional[_Identifiers] = None
    self._n = 0
    self._d = 0
    self._dedup = dedup
    self._dup: Optional[_Duplicates] = None
    self._plain: Optional["BruteForce"] = None

  def index(self, candidates: ArrayLike, identifiers: Optional[ArrayLike] = None) -> "BruteForce":
    if isinstance(candidates, torch.Tensor):
      ndim, nrows = candidates.dim(), candidates.shape[0] if candidates.dim() else 0
    else:
      candidates = np.asarray(candidates)
      ndim, nrows = candidates.ndim, candidates.shape[0] if candidates.ndim else 0
    if ndim != 2:                                                       # :547-550
      raise ValueError(f"The candidates tensor must be 2D (got {tuple(candidates.shape)}).")
    if identifiers is not None and len(identifiers) != nrows:          # :552-557
      raise ValueError(
          "The candidates and identifiers tensors must have the same number of"
          f" rows (got {nrows} candidates rows and"
          f" {len(identifiers)} identifier rows). ")
    cand = _as_f32_matrix(candidates, "candidates")
    self._wide = None
    if cand.shape[1] > MAX_FUSED_DIM:
      # embedding dims above 128: layer-owned row-major copy (:571-580), scored block by block
      # through the GEMM kernels (scores are GEMM sums, f32 accuracy)
      self._wide = cand.clone()
      self._index = self._wide
      self._ids = _Identifiers(identifiers, cand.shape[0])
      self._n, self._d = cand.shape
      return self
    self._dup, self._plain = None, None
    packed_rows = cand
    if self._dedup is True or (self._dedup and cand.shape[0] >= 4096):
      found = (_find_duplicates(cand, 2, 0.0) if self._dedup is True else _find_duplicates(cand))
      if found is not None:
        canonical, self._dup = found
        packed_rows = cand.index_select(0, canonical)      # the distinct rows, in corpus order
    handle = _IndexHandle()
    _lib.check(handle._lib.tfrs_index_set(handle.handle, _lib.ptr(packed_rows), packed_rows.shape[0],
                                          packed_rows.shape[1], _lib.current_stream()))
    torch.cuda.current_stream().synchronize()  # `cand` may be a temporary upload
    _raise_if_nonfinite_candidates(handle)
    self._index = handle                       # the previous index (if any) is dropped
    self._ids = _Identifiers(identifiers, cand.shape[0])
    self._n, self._d = cand.shape
    return self

  @property
  def _index_rows(self) -> int:
    """Rows held by the device index: the distinct rows of a de-duplicated corpus, else all."""
    return self._dup.count if self._dup is not None else self._n

  def index_from_dataset(self, candidates: Iterable, total_rows: Optional[int] = None) -> "BruteForce":
    """``TopK.index_from_dataset`` (:179-215).  With ``total_rows`` (the dataset's cardinality)
    the blocks are packed straight into a device index reserved once
    (``tfrs_index_reserve`` / ``tfrs_index_append``): peak memory is the packed index plus ONE
    block instead of the reference's ``tf.concat`` of every block (:196-215), which is what a
    corpus of 100 M rows needs.  Without it the blocks are concatenated like the reference."""
    if total_rows is None:
      return super().index_from_dataset(candidates)
    _check_candidates_with_identifiers(candidates)
    for first in candidates:
      first_block = first[1] if isinstance(first, (tuple, list)) else first
      if first_block.shape[1] > MAX_FUSED_DIM:
        return super().index_from_dataset(candidates)     # wide dims: plain row-major copy
      break
    handle, ids, has_ids, n, d = None, [], None, 0, 0
    # The index stores every appended block in a pseudo-random row order (the filter bound is
    # taken from a sample of the stored stages, csrc/topk_api.hip), which only mixes rows WITHIN
    # a block: small dataset batches (`movies.batch(128)`) are therefore gathered into chunks of
    # >= _APPEND_CHUNK_ROWS rows before they are appended.  Row order, and with it the returned
    # identifiers, is unchanged.
    pending, pending_rows = [], 0

    def flush():
      nonlocal pending, pending_rows
      if not pending:
        return
      chunk = pending[0] if len(pending) == 1 else torch.cat(pending, dim=0)
      _lib.check(handle._lib.tfrs_index_append(handle.handle, _lib.ptr(chunk), chunk.shape[0],
                                               _lib.current_stream()))
      torch.cuda.current_stream().synchronize()   # the blocks may be temporary uploads
      pending, pending_rows = [], 0

    for element in candidates:
      if isinstance(element, (tuple, list)):
        i, c = element
        has_ids = True
        ids.append(i.cpu().numpy() if isinstance(i, torch.Tensor) else np.asarray(i))
      else:
        c, has_ids = element, False
      block = _as_f32_matrix(c, "candidates")
      if handle is None:
        d = block.shape[1]
        handle = _IndexHandle()
        _lib.check(handle._lib.tfrs_index_reserve(handle.handle, int(total_rows), d,
                                                  _lib.current_stream()))
      elif block.shape[1] != d:
        raise ValueError(f"Candidate blocks disagree on the embedding dimension ({block.shape[1]} vs {d}).")
      if n + block.shape[0] > total_rows:
        raise ValueError(f"The dataset holds more than total_rows={total_rows} candidates.")
      pending.append(block)
      pending_rows += block.shape[0]
      n += block.shape[0]
      if pending_rows >= _APPEND_CHUNK_ROWS:
        flush()
    if handle is None:
      raise ValueError("The candidate dataset is empty.")
    flush()
    _raise_if_nonfinite_candidates(handle)
    self._index = handle
    self._dup, self._plain = None, None        # (streamed ingest: blocks are indexed as they come)
    self._ids = _Identifiers(np.concatenate(ids, axis=0) if has_ids else None, n)
    self._n, self._d = n, d
    return self

  def _identifier_table(self) -> _Identifiers:
    return self._ids

  def _query_rows(self, queries, k: int, embedded: bool = False) -> Tuple[Tensor, Tensor]:
    if self._index is None:                                             # :594-598
      raise ValueError(NOT_INDEXED_MESSAGE)
    q = queries if embedded else self._embed(queries)                   # :600-601
    if q.shape[1] != self._d:
      raise ValueError(f"Query dimension {q.shape[1]} does not match the index ({self._d}).")
    lib = _lib.load()
    nq = q.shape[0]
    if getattr(self, "_wide", None) is not None:
      if k > self._n:
        raise ValueError(f"input must have at least k columns (k={k}, candidates={self._n})")
      scores = torch.zeros((nq, k), dtype=torch.float32, device=q.device)
      rows = torch.zeros((nq, k), dtype=torch.int32, device=q.device)
      _wide_topk_update(q, self._wide, 0, k, scores, rows, 0)
      self._last_call = None
      return scores, rows
    if k > self._n:
      raise ValueError(f"input must have at least k columns (k={k}, candidates={self._n})")
    if k > MAX_FUSED_K:
      if self._dup is not None:     # pages + expansion: (rare) search a plain copy of the corpus
        if self._plain is None:
          self._plain = BruteForce(k=self._k, dedup=False).index(self.candidates())
        return self._plain._query_rows_paged(q, k)
      return self._query_rows_paged(q, k)
    self._raise_if_nonfinite_queries()             # (deferred: recorded by an EARLIER call's kernels)
    kk = min(k, self._index_rows)                   # (de-duplicated: the best kk DISTINCT rows)
    scores = torch.empty((nq, kk), dtype=torch.float32, device=q.device)
    rows = torch.empty((nq, kk), dtype=torch.int32, device=q.device)
    ws = _workspace(lib.tfrs_bruteforce_topk_workspace_bytes(nq, self._index_rows, self._d, kk))
    _lib.check(lib.tfrs_bruteforce_topk(
        self._index.handle, _lib.ptr(q), nq, kk, _lib.ptr(scores), _lib.ptr(rows),
        _lib.ptr(ws), ws.numel(), _lib.current_stream()))               # :603-605
    self._last_call = (ws, nq, kk)
    if self._check_finite:
      torch.cuda.current_stream().synchronize()
      self._raise_if_nonfinite_queries()
    if self._dup is None:
      return scores, rows
    # every original row is a candidate with its distinct row's score: exact top-k of the corpus
    out_s = torch.empty((nq, k), dtype=torch.float32, device=q.device)
    out_r = torch.empty((nq, k), dtype=torch.int32, device=q.device)
    _lib.check(lib.tfrs_topk_expand_duplicates(
        _lib.ptr(scores), _lib.ptr(rows), nq, kk, _lib.ptr(self._dup.start), _lib.ptr(self._dup.rows), k,
        _lib.ptr(out_s), _lib.ptr(out_r), _lib.current_stream()))
    return out_s, out_r

  def _query_rows_paged(self, q: Tensor, k: int) -> Tuple[Tensor, Tensor]:
    """``k`` beyond the selection kernels' 1024 slots (``tf.math.top_k`` has no limit, :605): pages
    of up to 1024 results, each the best rows strictly after the previous page's last (score, row)
    in the result order (``tfrs_bruteforce_topk_below``); the pages are written side by side into
    the ``[B, k]`` outputs, which are therefore exactly the sorted top-k.  One all-f32 scan of the
    corpus per page."""
    if k > self._n:
      raise ValueError(f"input must have at least k columns (k={k}, candidates={self._n})")
    lib = _lib.load()
    nq = q.shape[0]
    scores = torch.empty((nq, k), dtype=torch.float32, device=q.device)
    rows = torch.empty((nq, k), dtype=torch.int32, device=q.device)
    ws = _workspace(lib.tfrs_bruteforce_topk_below_workspace_bytes(nq, self._n, self._d, MAX_FUSED_K))
    page_s = torch.empty((nq, MAX_FUSED_K), dtype=torch.float32, device=q.device)
    page_r = torch.empty((nq, MAX_FUSED_K), dtype=torch.int32, device=q.device)
    done = 0
    while done < k:
      kk = min(MAX_FUSED_K, k - done)
      ps, pr = (page_s, page_r) if kk == MAX_FUSED_K else (page_s[:, :kk].contiguous(), page_r[:, :kk].contiguous())
      last_s = None if done == 0 else scores[:, done - 1:]        # row stride k: element [q, done - 1]
      last_r = None if done == 0 else rows[:, done - 1:]
      _lib.check(lib.tfrs_bruteforce_topk_below(
          self._index.handle, _lib.ptr(q), nq, kk,
          None if last_s is None else ctypes.c_void_p(last_s.data_ptr()),
          None if last_r is None else ctypes.c_void_p(last_r.data_ptr()), k,
          _lib.ptr(ps), _lib.ptr(pr), _lib.ptr(ws), ws.numel(), _lib.current_stream()))
      scores[:, done:done + kk] = ps
      rows[:, done:done + kk] = pr
      done += kk
    self._last_call = None
    return scores, rows

  def nonfinite_flags(self, reset: int = 0) -> int:
    """The index handle's flag word (``tfrs_index_nonfinite``): bit 0 non-finite candidates, bit 1 non-finite
    queries in a call whose kernels have completed.  Reading it does not synchronise."""
    if not isinstance(self._index, _IndexHandle):
      return 0
    out = ctypes.c_int32(0)
    _lib.check(self._index._lib.tfrs_index_nonfinite(self._index.handle, int(reset), ctypes.byref(out)))
    return int(out.value)

  def _raise_if_nonfinite_queries(self) -> None:
    if self.nonfinite_flags() & 2:
      self.nonfinite_flags(reset=2)
      raise ValueError("BruteForce: the queries of this or an earlier call contained NaN or Inf (or a row norm beyond "
                       "the float32 range): the result rows of those queries hold non-finite scores and unspecified "
                       "indices; every other row is exact.  Queries must be finite (include/tfrs_hip.h).")

  def last_redo_count(self) -> int:
    """Queries of the most recent ``call`` that were answered by the exact-recompute path of
    the fp16-prefiltered search (0 on well-behaved data).  Synchronises the stream."""
    if getattr(self, "_last_call", None) is None:
      return 0
    ws, nq, k = self._last_call
    out = ctypes.c_int32(0)
    _lib.check(_lib.load().tfrs_bruteforce_topk_redo_count(
        _lib.ptr(ws), nq, self._index_rows, k, ctypes.byref(out), _lib.current_stream()))
    return int(out.value)

  def last_redo_reasons(self) -> dict:
    """``last_redo_count`` split by cause (include/tfrs_hip.h).  Synchronises the stream."""
    names = ("list_overflow", "statistical_bound", "retained_set", "longest_list")
    if getattr(self, "_last_call", None) is None:
      return dict.fromkeys(names, 0)
    ws, nq, k = self._last_call
    out = (ctypes.c_int32 * 4)()
    _lib.check(_lib.load().tfrs_bruteforce_topk_redo_reasons(
        _lib.ptr(ws), nq, self._index_rows, k, out, _lib.current_stream()))
    return {name: int(out[i]) for i, name in enumerate(names)}

  def call(self, queries, k: Optional[int] = None):
    k = k if k is not None else self._k
    scores, rows = self._query_rows(queries, k)
    return scores, self._ids.gather(rows)                               # :607

  def make_graphed_call(self, example_queries, k: Optional[int] = None):
    """``call`` for a fixed batch shape, captured once in a HIP graph and replayed.

    A small-batch query is a chain of ~8 short kernels (query norms, threshold pass, filter
    pass, exact re-scoring); replaying them from a graph removes the per-launch host cost that
    dominates the latency of single queries.  ``query_model`` (if any) and the identifier
    lookup stay outside the graph; the search itself -- same kernels, same results -- is
    inside.  Returns ``f(queries) -> (scores, identifiers)``; the returned score tensor is
    overwritten by the next call.  The index must not be re-indexed afterwards."""
    k = k if k is not None else self._k
    if self._index is None:
      raise ValueError(NOT_INDEXED_MESSAGE)
    if getattr(self, "_wide", None) is not None:
      raise NotImplementedError("make_graphed_call: embedding dims above 128 use per-block launches")
    static_q = self._embed(example_queries).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      for _ in range(2):
        self._query_rows(static_q, k, embedded=True)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
      scores, rows = self._query_rows(static_q, k, embedded=True)

    def graphed(queries):
      q = self._embed(queries)
      if q.shape != static_q.shape:
        raise ValueError(f"graphed call was captured for queries of shape {tuple(static_q.shape)}; "
                         f"got {tuple(q.shape)}")
      static_q.copy_(q, non_blocking=True)
      graph.replay()
      return scores, self._ids.gather(rows)

    graphed.graph = graph
    return graphed

  def candidates(self) -> Tensor:
    """The indexed candidate matrix (unpacked copy), for checkpointing."""
    if self._index is None:
      raise ValueError(NOT_INDEXED_MESSAGE)
    if getattr(self, "_wide", None) is not None:
      return self._wide.clone()
    out = torch.empty((self._index_rows, self._d), dtype=torch.float32, device=_device())
    _lib.check(_lib.load().tfrs_index_unpack(self._index.handle, _lib.ptr(out),
                                             _lib.current_stream()))
    if self._dup is not None:       # every original row from its distinct row
      out = out.index_select(0, self._dup.distinct_of_row.long())
    return out

  def is_exact(self) -> bool:
    return True

  # -- persistence (the role of SavedModel export in the reference, --------------------------
  #    factorized_top_k_test.py:152-165, basic_retrieval.ipynb "serving") ---------------------
  def state_dict(self) -> Dict[str, Any]:   # type: ignore[override]
    """Everything needed to rebuild the index: the row-major float32 candidates (unpacked from
    the device images), the identifiers (``None`` = row numbers) and ``k``."""
    if self._index is None:
      raise ValueError(NOT_INDEXED_MESSAGE)
    ids = self._ids
    identifiers = None
    if ids.host is not None:
      identifiers = ids.host
    elif ids.device is not None:
      identifiers = ids.device.cpu().numpy()
    return {"candidates": self.candidates().cpu().numpy(), "identifiers": identifiers, "k": self._k}

  def load_state_dict(self, state: Dict[str, Any]) -> "BruteForce":   # type: ignore[override]
    self._k = int(state.get("k", self._k))
    return self.index(state["candidates"], state.get("identifiers"))

  def save(self, path: str) -> None:
    """Writes the index to one ``.npz`` file (string identifiers are stored as a unicode array)."""
    st = self.state_dict()
    payload = {"candidates": st["candidates"], "k": np.asarray(st["k"])}
    if st["identifiers"] is not None:
      payload["identifiers"] = np.asarray(st["identifiers"])
    np.savez(path, **payload)

  @classmethod
  def load(cls, path: str, query_model: Optional[Callable] = None) -> "BruteForce":
    with np.load(path, allow_pickle=False) as f:
      layer = cls(query_model=query_model, k=int(f["k"]))
      return layer.index(f["candidates"], f["identifiers"] if "identifiers" in f.files else None)


class Streaming(TopK):
  """Retrieves the K highest scoring items from a large candidate stream
  (reference :336-512).  Keeps only a reference to the candidate iterable and
  re-reads it on every call; the running state is ``[B, <=K]`` in HBM.

  ``cache_packed_blocks`` (default on; not in the reference): when the candidates are a
  **list / tuple of GPU-resident tensors** (the serving case: device blocks held by the caller),
  the packed f32 + fp16 images built from them are kept between calls in ONE device index --
  rows in stream order, so global row numbers are the reference's counter (:477-488) -- and a
  call is then exactly ``BruteForce.call`` on it: no re-upload, no re-pack, no per-block
  threshold pass.  The cache keeps the tensor objects it packed alive and compares object
  identity, storage pointer and version counter on every call, so an address recycled by the
  allocator for a *new* tensor can never match.  Never cached (block-by-block path, device
  footprint of one block, exactly the reference's behaviour): lazily mapped / generated
  datasets (e.g. ``candidates.map(item_model)``: re-embedded on every pass), host arrays,
  tensors that take part in autograd (parameters, outputs with a ``grad_fn``), and corpora
  whose packed images would exceed ``cache_max_bytes`` (default 64 GiB).  Detached views of a
  trained table are safe: this package's fused optimizer kernels bump the table's version
  counter after writing through raw pointers.  A block written in place by a foreign raw
  kernel or through ``.data`` after the first call must be re-announced with
  ``index_from_dataset``.
  The cache holds 2.1x the candidate bytes (1.6x at dim 128).
  """

  def __init__(self, query_model: Optional[Callable] = None, k: int = 10,
               handle_incomplete_batches: bool = True,
               num_parallel_calls: Optional[int] = None, sorted_order: bool = True,
               cache_packed_blocks: bool = True, cache_max_bytes: int = 64 << 30,
               group_max_bytes: Optional[int] = None) -> None:
    super().__init__(k=k)
    self.query_model = query_model
    self._candidates = None
    self._handle_incomplete_batches = handle_incomplete_batches
    # `num_parallel_calls` / `sorted_order` (reference :343-350, :503: tf.data's parallel map and the
    # `sorted=` flag of tf.math.top_k) are accepted for API parity and have no effect here: the
    # blocks of a group are scored by one launch, and results always come back sorted
    self._num_parallel_calls = num_parallel_calls
    self._sorted = sorted_order
    # candidate bytes (float32 rows) searched per library call on the block-by-block path: the
    # blocks of a group are read IN PLACE (tfrs_streaming_topk_update_blocks), so this bounds how
    # many lazily produced blocks are alive at once plus the fp16 image of the group
    # rows kept alive per group of lazily produced blocks (+ their fp16 image for large batches): by default an
    # eighth of the device memory that is free at the first call, at most 8 GiB (a fixed 8 GiB
    # ran a `.map(model)` dataset out of memory on a busy GPU where the one-block-at-a-time path had fitted)
    self._group_max_bytes = None if group_max_bytes is None else max(int(group_max_bytes), 1)
    self._last_ids: Optional[_Identifiers] = None
    self._cache_blocks = cache_packed_blocks
    self._cache_max_bytes = int(cache_max_bytes)
    self._cache_key = None
    self._cache: Optional[BruteForce] = None
    self._cache_alive = None
    self._probe_fast = None
    self._base_row = 0

  def index_from_dataset(self, candidates: Iterable) -> "Streaming":
    _check_candidates_with_identifiers(candidates)                     # :386
    self._candidates = candidates                                      # :388
    self._cache_key, self._cache, self._cache_alive, self._probe_fast = None, None, None, None
    return self

  def index(self, candidates, identifiers=None) -> "Streaming":
    """Not implemented. Please call `index_from_dataset` instead (:392-402)."""
    raise NotImplementedError(
        "The streaming top k class only accepts datasets. "
        "Please call `index_from_dataset` instead.")

  def _identifier_table(self) -> _Identifiers:
    return self._last_ids

  # -- cached path ---------------------------------------------------------------------------
  @staticmethod
  def _cacheable(t) -> bool:
    """A tensor whose content the cache may assume stable while (object, storage, version) are:
    resident f32/integer data that is not part of an autograd graph."""
    return (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous()
            and not t.requires_grad and t.grad_fn is None)

  def _block_keys(self):
    """(key, blocks, ids) when the source is a list / tuple of cacheable GPU tensors, else None.
    Never iterates a lazy dataset (that would materialise it) and never touches the device.
    ``key`` holds the tensor OBJECTS (kept alive by the cache) next to their storage pointers
    and version counters: identity is part of the comparison, so a recycled address of a
    dead tensor cannot match."""
    src = self._candidates
    if not isinstance(src, (list, tuple)) or len(src) == 0:
      return None
    fast = getattr(self, "_probe_fast", None)
    if fast is not None and fast[0] is src and len(src) == len(fast[1]):
      same = True
      for element, (ref, tensors) in zip(src, fast[1]):
        if element is not ref:
          same = False
          break
        for t, ptr, ver in tensors:
          if t._version != ver or t.data_ptr() != ptr:
            same = False
            break
        if not same:
          break
      if same:
        return fast[2]
    self._probe_fast = None
    keys, blocks, ids, seen = [], [], [], []
    for element in src:
      block_ids = None
      if isinstance(element, (tuple, list)):
        if len(element) != 2:
          return None
        block_ids, block = element
      else:
        block = element
      if not (self._cacheable(block) and block.dim() == 2 and block.dtype == torch.float32):
        return None
      tensors = [(block, block.data_ptr(), block._version)]
      if block_ids is not None:
        if not self._cacheable(block_ids):
          return None     # host identifiers may change silently between calls
        tensors.append((block_ids, block_ids.data_ptr(), block_ids._version))
      keys.append(tuple((id(t), ptr, tuple(t.shape), ver) for t, ptr, ver in tensors))
      blocks.append(block)
      ids.append(block_ids)
      seen.append((element, tensors))
    d = blocks[0].shape[1]
    packed = sum(b.shape[0] for b in blocks) * (d * 6 + 32 + 4)    # f32 + fp16 images + row map
    if packed > self._cache_max_bytes:
      return None
    result = (tuple(keys), blocks, ids)
    self._probe_fast = (src, seen, result)
    return result

  def _cached_index(self, k: int) -> Optional["BruteForce"]:
    if not self._cache_blocks:
      return None
    if torch.cuda.is_current_stream_capturing():
      # a captured step (Model.fit / evaluate replay HIP graphs) runs no host code on replay, so the
      # (object, storage, version) comparison below could never notice that a block changed: under
      # capture the blocks are read in place, as the reference re-reads its dataset on every call
      return None
    probe = self._block_keys()
    if probe is None:
      self._cache_key, self._cache, self._cache_alive = None, None, None
      return None
    key, blocks, ids = probe
    if not self._handle_incomplete_batches and any(b.shape[0] < k for b in blocks):   # :431-436
      raise ValueError(BATCH_TOO_SMALL_MESSAGE.format(k=k))
    if key != self._cache_key:
      total = sum(b.shape[0] for b in blocks)
      if self._base_row + total > 0x7FFFFFFF:
        raise ValueError("Streaming: global row numbers exceed int32 (base_row + rows = %d)"
                         % (self._base_row + total))
      has_ids = ids[0] is not None
      bf = BruteForce(k=self._k)
      bf.index_from_dataset([(i, b) for i, b in zip(ids, blocks)] if has_ids else blocks,
                            total_rows=total)
      # the packed tensors stay referenced for as long as their images are served: id() in the
      # key is only meaningful while the object lives
      self._cache_key, self._cache, self._cache_alive = key, bf, (blocks, ids)
    return self._cache

  def _query_rows(self, queries, k: int) -> Tuple[Tensor, Tensor]:
    if self._candidates is None:                                        # :412-416
      raise ValueError(NOT_INDEXED_MESSAGE)
    q = self._embed(queries)                                            # :418-419
    cached = self._cached_index(k)
    if cached is not None:
      if q.shape[1] != cached._d:
        raise ValueError(f"Candidate dimension {cached._d} does not match queries ({q.shape[1]}).")
      kk = min(k, cached._n)     # handle_incomplete_batches semantics: fewer rows than k (:465-468)
      scores, rows = cached._query_rows(q, kk, embedded=True)
      self._last_ids = cached._ids
      return scores, (rows + self._base_row if self._base_row else rows)
    if k > MAX_FUSED_K:
      # more results than one pass of the selection kernels holds: the paged search of BruteForce
      # needs the corpus resident (one scan per 1024 results), so the stream is ingested once
      # for this call -- rows in stream order, i.e. the reference's counter (:477-488)
      bf = BruteForce(k=self._k).index_from_dataset(self._candidates)
      if q.shape[1] != bf._d:
        raise ValueError(f"Candidate dimension {bf._d} does not match queries ({q.shape[1]}).")
      scores, rows = bf._query_rows(q, min(k, bf._n), embedded=True)
      self._last_ids = bf._ids
      return scores, (rows + self._base_row if self._base_row else rows)
    lib = _lib.load()
    nq, d = q.shape
    # non-finite inputs: Streaming re-reads its dataset on every call, so the blocks are NOT validated (that would
    # double the traffic); the queries and the best score of every row are, without a synchronisation -- a violation
    # raises at the next call (see BruteForce.__init__ / include/tfrs_hip.h)
    if not hasattr(self, "_finite"):
      self._finite = _DeferredFinite()
    self._finite.check("Streaming")
    state_scores = torch.zeros((nq, k), dtype=torch.float32, device=q.device)
    state_rows = torch.zeros((nq, k), dtype=torch.int32, device=q.device)
    state_len = 0
    counter = self._base_row                                            # :421-422
    ids = []
    has_ids = False
    new_len = ctypes.c_int32(0)
    ws = None
    # Groups of consecutive blocks are searched where they lie (no packed copy) when the row
    # layout allows it: one launch chain per group instead of one per block.
    grouped = d in _RAW_DIMS and k <= MAX_FUSED_K
    group: list = []
    group_rows = 0
    group_base = counter

    def flush_group():
      nonlocal group, group_rows, group_base, state_len, ws
      if not group:
        return
      n = len(group)
      ptrs = (ctypes.c_void_p * n)(*[b.data_ptr() for b in group])
      rows = (ctypes.c_int64 * n)(*[b.shape[0] for b in group])
      need = lib.tfrs_streaming_topk_blocks_workspace_bytes(nq, group_rows, d, k)
      if ws is None or ws.numel() < need:
        ws = None
        ws = _workspace(need)
      _lib.check(lib.tfrs_streaming_topk_update_blocks(
          _lib.ptr(q), nq, d, ptrs, rows, n, group_base, group_base - self._base_row, k,
          _lib.ptr(state_scores), _lib.ptr(state_rows), state_len, ctypes.byref(new_len),
          _lib.ptr(ws), ws.numel(), _lib.current_stream()))            # :424-472 for every block of the group
      state_len = int(new_len.value)
      # The blocks are only referenced by the kernels just enqueued on THIS stream.  A block the dataset
      # produced on another stream would be handed back to that stream's allocator pool when the reference is
      # dropped, and could be overwritten before these kernels have read it: tell the allocator who reads it
      # (a no-op for blocks of the current stream).
      stream = torch.cuda.current_stream(q.device)
      for b in group:
        if b.is_cuda:
          b.record_stream(stream)
      group, group_rows = [], 0

    for element in self._candidates:
      if isinstance(element, (tuple, list)):
        block_ids, block = element
        has_ids = True
        # (device identifiers stay on the device: a `.cpu()` here would synchronise once per block)
        ids.append(block_ids if isinstance(block_ids, torch.Tensor) else np.asarray(block_ids))
      else:
        block = element
      block = _as_f32_matrix(block, "candidates")
      nb = block.shape[0]
      if block.shape[1] != d:
        raise ValueError(f"Candidate dimension {block.shape[1]} does not match queries ({d}).")
      if not self._handle_incomplete_batches and nb < k:               # :431-436, :34-54
        raise ValueError(BATCH_TOO_SMALL_MESSAGE.format(k=k))
      if counter + nb > 0x7FFFFFFF:
        raise ValueError("Streaming: global row numbers exceed int32 (%d)" % (counter + nb))
      if d > MAX_FUSED_DIM:
        state_len = _wide_topk_update(q, block, counter, k, state_scores, state_rows, state_len)
        counter += nb
        continue
      if grouped and nb > 0 and block.data_ptr() % 16 == 0:
        if not group:
          group_base = counter
        group.append(block)
        group_rows += nb
        counter += nb
        if len(group) == _RAW_MAX_BLOCKS or group_rows * d * 4 >= self._group_bytes(q.device):
          flush_group()
        continue
      flush_group()          # (a block the grouped path cannot take: keep the stream order)
      need = lib.tfrs_streaming_topk_workspace_bytes(nq, nb, d, k)
      if ws is None or ws.numel() < need:
        ws = _workspace(need)
      _lib.check(lib.tfrs_streaming_topk_update(
          _lib.ptr(q), nq, d, _lib.ptr(block), nb, counter, k, _lib.ptr(state_scores),
          _lib.ptr(state_rows), state_len, ctypes.byref(new_len), _lib.ptr(ws),
          ws.numel(), _lib.current_stream()))                          # :424-472
      state_len = int(new_len.value)
      counter += nb                                                     # :477-478
    flush_group()
    all_ids = None
    if has_ids:
      if all(isinstance(i, torch.Tensor) for i in ids):
        # along axis 0 as the reference concatenates them (2-D identifier blocks keep their rows), on the
        # device of the first block (a mix of host and device blocks would make torch.cat raise)
        all_ids = torch.cat([i.to(ids[0].device) for i in ids], dim=0)
      else:
        all_ids = np.concatenate([i.cpu().numpy() if isinstance(i, torch.Tensor) else i for i in ids], axis=0)
    self._last_ids = _Identifiers(all_ids, counter - self._base_row)
    if state_len > 0 and nq > 0:
      # (the carried state with the queries in one launch; a state shorter than k -- fewer candidates than k -- is copied first)
      self._finite.note(q, state_scores[:, :state_len])
    return state_scores[:, :state_len], state_rows[:, :state_len]

  def _group_bytes(self, device) -> int:
    if self._group_max_bytes is None:
      free = torch.cuda.mem_get_info(device)[0] if device.type == "cuda" else (8 << 30)
      self._group_max_bytes = max(64 << 20, min(8 << 30, free // 8))
    return self._group_max_bytes

  def _ids_of_rows(self, rows: Tensor):
    return self._last_ids.gather(rows - self._base_row if self._base_row else rows)

  def call(self, queries, k: Optional[int] = None):
    k = k if k is not None else self._k
    scores, rows = self._query_rows(queries, k)
    return scores, self._ids_of_rows(rows)                              # :438

  def is_exact(self) -> bool:
    return True


_INT32_MAX = 0x7FFFFFFF


_BASE_WORDS: dict = {}


def _base_row_words(base_row: int, dev) -> Tensor:
  """The two int32 words of a shard's int64 base row as a device tensor, built once per (device, base row):
  the wide exchange used to build them with ``torch.tensor([...])`` -- a host-to-device copy and a
  synchronisation point in every query."""
  key = (str(dev), int(base_row))
  words = _BASE_WORDS.get(key)
  if words is None:
    if len(_BASE_WORDS) > 64:
      _BASE_WORDS.clear()
    words = torch.tensor([base_row & 0xFFFFFFFF, base_row >> 32], dtype=torch.int64).to(torch.int32).to(dev)
    _BASE_WORDS[key] = words
  return words


def _exchange_and_merge_wide(scores: Tensor, local_rows: Tensor, k: int, group, merge: Optional[Callable],
                             base_row: int):
  """The same single exchange for corpora whose GLOBAL row numbers do not fit int32 ("ids
  beyond 2^31 need i64"; 8 x 288 GB of dim-64 rows are 9 * 10^9 rows; the reference's counter is int32,
  :380-382).  A shard always fits int32, so ranks exchange (score bits, LOCAL row) plus their int64 base
  row (two extra words of the same all_gather); the parts are ordered by base row and merged on the
  synthetic index ``part * k + position`` -- every part is already sorted (score desc, row asc), so part
  order then position IS global-row order among equal scores, and the 64-bit merge keys need no wider
  row field; ``base[part] + local[part, q, position]`` restores int64 global rows afterwards."""
  import torch.distributed as dist
  dev = scores.device
  nq = scores.shape[0]
  if scores.shape[1] < k:
    pad = k - scores.shape[1]
    scores = torch.cat([scores, scores.new_full((nq, pad), float("-inf"))], dim=1)
    local_rows = torch.cat([local_rows, local_rows.new_full((nq, pad), -1)], dim=1)
  world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
  words = 2 * nq * k + 2
  mine = torch.empty((words,), dtype=torch.int32, device=dev)
  mine[:nq * k].copy_(scores.contiguous().view(torch.int32).reshape(-1))
  mine[nq * k:2 * nq * k].copy_(local_rows.to(torch.int32).reshape(-1))
  mine[2 * nq * k:].copy_(_base_row_words(base_row, dev))     # (device-resident: no host copy in the query path)
  gathered = torch.empty((world, words), dtype=torch.int32, device=dev)
  if world > 1 or os.environ.get("TFRS_FORCE_EXCHANGE", "0") == "1":
    dist.all_gather_into_tensor(gathered.view(-1), mine, group=group)
  else:
    gathered[0].copy_(mine)
  tail = gathered[:, 2 * nq * k:].to(torch.int64)
  bases = (tail[:, 0] & 0xFFFFFFFF) | (tail[:, 1] << 32)                  # [world] int64
  order = torch.argsort(bases)                                            # parts in global-row order
  parts = gathered.index_select(0, order)
  bases = bases.index_select(0, order)
  part_scores = parts[:, :nq * k].contiguous().view(torch.float32).reshape(world, nq, k)
  part_rows = parts[:, nq * k:2 * nq * k].reshape(world, nq, k)
  synth = (torch.arange(world, device=dev, dtype=torch.int32)[:, None, None] * k
           + torch.arange(k, device=dev, dtype=torch.int32)[None, None, :]).expand(world, nq, k)
  synth = torch.where(part_rows < 0, torch.full_like(part_rows, -1), synth).contiguous()
  if merge is not None:
    out_s, out_i = merge(part_scores, synth, k)
    out_s, out_i = out_s.to(dev), out_i.to(dev)
  else:
    out_s = torch.empty((nq, k), dtype=torch.float32, device=dev)
    out_i = torch.empty((nq, k), dtype=torch.int32, device=dev)
    _lib.check(_lib.load().tfrs_topk_merge(
        _lib.ptr(part_scores.contiguous()), _lib.ptr(synth), world, nq, k, k, _lib.ptr(out_s), _lib.ptr(out_i),
        None, 0, _lib.current_stream()))
  valid = out_i >= 0
  idx = out_i.clamp_min(0).long()
  part, pos = idx // k, idx % k
  qidx = torch.arange(nq, device=dev)[:, None].expand(nq, k)
  local = part_rows[part, qidx, pos].long()
  rows64 = torch.where(valid, bases[part] + local, torch.full_like(local, -1))
  return out_s, rows64


def _exchange_and_merge(scores: Tensor, rows: Tensor, k: int, group, merge: Optional[Callable]):
  """The ONE exchange step of row-sharded top-K: this rank's (score bits, global row)[nq, k]
  lists go back to back into one int32 buffer [2, nq, k]; a single all_gather delivers
  [world, 2, nq, k] (2 * nq * k * 4 bytes per rank, one direct xGMI send per peer) and the merge
  kernel reads the parts in place with the (score desc, row asc) rule, so every rank ends
  with exactly the single-GPU result."""
  import torch.distributed as dist
  if not (dist.is_available() and dist.is_initialized()):
    return scores, rows
  world = dist.get_world_size(group)
  if world == 1 and os.environ.get("TFRS_FORCE_EXCHANGE", "0") != "1":
    return scores, rows
  nq = scores.shape[0]
  if scores.shape[1] < k:      # a shard with fewer than k rows: pad with empty slots (row -1)
    pad = k - scores.shape[1]
    scores = torch.cat([scores, scores.new_full((nq, pad), float("-inf"))], dim=1)
    rows = torch.cat([rows, rows.new_full((nq, pad), -1)], dim=1)
  mine = torch.empty((2, nq, k), dtype=torch.int32, device=scores.device)
  mine[0].copy_(scores.contiguous().view(torch.int32))
  mine[1].copy_(rows)
  gathered = torch.empty((world, 2, nq, k), dtype=torch.int32, device=scores.device)
  dist.all_gather_into_tensor(gathered.view(world * 2 * nq, k), mine.view(2 * nq, k), group=group)
  if merge is not None:
    return merge(gathered[:, 0].contiguous().view(torch.float32), gathered[:, 1].contiguous(), k)
  out_s = torch.empty((nq, k), dtype=torch.float32, device=scores.device)
  out_i = torch.empty((nq, k), dtype=torch.int32, device=scores.device)
  base = gathered.view(-1)
  _lib.check(_lib.load().tfrs_topk_merge_strided(
      base.data_ptr(), base.data_ptr() + nq * k * 4, world, 2 * nq * k, nq, k, k,
      _lib.ptr(out_s), _lib.ptr(out_i), _lib.current_stream()))
  return out_s, out_i


def _global_identifiers(rows: Tensor, local_ids: Optional[Tensor], base_row: int, n_local: int, group):
  """identifiers[global row] when every rank only holds the identifiers of its own rows: the
  owner fills its entries, one all_reduce(sum) of the [nq, k] int64 matrix completes it."""
  import torch.distributed as dist
  if local_ids is None:
    return rows
  local = rows.long() - base_row
  mine = (local >= 0) & (local < n_local)
  out = torch.zeros(rows.shape, dtype=torch.int64, device=rows.device)
  out[mine] = local_ids.to(rows.device).long()[local[mine]]
  if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
    dist.all_reduce(out, group=group)
  return out.to(local_ids.dtype)


def _check_shard_rows(base_row: int, n_local: int) -> None:
  if base_row < 0 or n_local > _INT32_MAX:
    raise ValueError(f"a shard holds at most 2^31 - 1 rows and base_row must be >= 0 (got {n_local} rows "
                     f"from row {base_row})")


def _any_rank_beyond_int32(base_row: int, n_local: int, group) -> bool:
  """True when some rank's global rows exceed int32 (decided once, at index time, with one small host
  collective, so that every rank takes the same exchange path): results are then int64 row numbers."""
  import torch.distributed as dist
  mine = base_row + n_local > _INT32_MAX
  if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
    return mine
  flags = [None] * dist.get_world_size(group)
  dist.all_gather_object(flags, bool(mine), group=group)
  return any(flags)


class ShardedBruteForce(TopK):
  """Brute-force retrieval over a corpus that is row-sharded across the GPUs of a node
  (one process per GPU, ``torch.distributed`` backend ``nccl`` = RCCL over xGMI).

  Rank r owns candidate rows ``[base_row, base_row + n_local)`` of the global corpus.
  ``call`` scores the (replicated) query batch against the local shard, all-gathers the
  per-shard ``(score, global row)[B, K]`` lists -- the only exchange step of the path,
  ``2 * B * K * 4`` bytes per rank -- and merges them with the same
  (score desc, row asc) rule, so every rank holds exactly the single-GPU result.
  ``identifiers`` (numeric, this shard's rows) are resolved by their owners and completed with
  one small all_reduce.  Not part of the reference (which has no multi-device top-K).
  """

  def __init__(self, query_model: Optional[Callable] = None, k: int = 10,
               process_group=None, name: Optional[str] = None,
               local_search: Optional[Callable] = None, merge: Optional[Callable] = None):
    super().__init__(k=k, name=name)
    self.query_model = query_model
    self._group = process_group
    self._local = BruteForce(k=k)
    self._base_row = 0
    self._n_local = 0
    self._local_ids: Optional[Tensor] = None
    self._ids = _Identifiers(None, 0)
    # injection points so the collective logic can be exercised on CPU (gloo) in tests
    self._local_search = local_search
    self._merge = merge

  def _set_identifiers(self, identifiers, n_local: int) -> None:
    self._local_ids = None
    if identifiers is None:
      return
    if len(identifiers) != n_local:
      raise ValueError("The candidates and identifiers tensors must have the same number of"
                       f" rows (got {n_local} candidates rows and {len(identifiers)} identifier rows). ")
    ids = identifiers if isinstance(identifiers, torch.Tensor) else np.asarray(identifiers)
    if not isinstance(ids, torch.Tensor):
      if ids.dtype.kind not in "iub":
        raise NotImplementedError("ShardedBruteForce resolves integer identifiers only; map other "
                                  "identifier types from the returned global row numbers on the host.")
      ids = torch.as_tensor(ids)
    self._local_ids = ids

  def index(self, candidates: ArrayLike, identifiers: Optional[ArrayLike] = None,
            base_row: int = 0) -> "ShardedBruteForce":
    n_local = len(candidates)
    _check_shard_rows(int(base_row), n_local)
    self._set_identifiers(identifiers, n_local)
    self._base_row, self._n_local = int(base_row), n_local
    self._wide = _any_rank_beyond_int32(int(base_row), n_local, self._group)
    if self._local_search is None:
      self._local.index(candidates)
    else:
      self._cand = candidates
    return self

  def index_from_dataset(self, candidates: Iterable, total_rows: Optional[int] = None,
                         base_row: int = 0) -> "ShardedBruteForce":
    """This rank's shard from an iterable of blocks (streamed into the device index when
    ``total_rows`` -- the shard's row count -- is given, see ``BruteForce.index_from_dataset``)."""
    self._local.index_from_dataset(candidates, total_rows=total_rows)
    _check_shard_rows(int(base_row), self._local._n)
    ids = self._local._ids
    self._set_identifiers(ids.device if ids.device is not None else
                          (None if ids.is_range else ids.host), self._local._n)
    self._base_row, self._n_local = int(base_row), self._local._n
    self._wide = _any_rank_beyond_int32(int(base_row), self._local._n, self._group)
    return self

  def _identifier_table(self) -> _Identifiers:
    return self._ids

  def _query_rows(self, queries, k: int) -> Tuple[Tensor, Tensor]:
    if self._local_search is None:
      kk = min(k, self._local._n)
      scores, rows = self._local._query_rows(self._embed(queries), kk)
    else:
      scores, rows = self._local_search(queries, self._cand, k)
    if getattr(self, "_wide", False):       # global rows beyond int32: exchange local rows + int64 bases
      return _exchange_and_merge_wide(scores, rows, k, self._group, self._merge, self._base_row)
    rows = rows + self._base_row
    return _exchange_and_merge(scores, rows, k, self._group, self._merge)

  def call(self, queries, k: Optional[int] = None):
    k = k if k is not None else self._k
    scores, rows = self._query_rows(queries, k)
    return scores, _global_identifiers(rows, self._local_ids, self._base_row, self._n_local, self._group)

  def is_exact(self) -> bool:
    return True


class ShardedStreaming(Streaming):
  """``Streaming`` over a candidate stream that is row-sharded across ranks (BASELINE.json
  configs[2]: "Streaming top-100 sharded over 8 MI355X"): rank r streams ITS blocks, whose rows
  carry the global row numbers ``base_row, base_row + 1, ...``; the per-shard results are
  exchanged and merged exactly like ``ShardedBruteForce``.  Returns global row numbers
  (identifiers in the stream are resolved for the local rows only when every rank passes
  them; otherwise map rows on the host)."""

  def __init__(self, query_model: Optional[Callable] = None, k: int = 10,
               handle_incomplete_batches: bool = True, process_group=None,
               cache_packed_blocks: bool = True, merge: Optional[Callable] = None) -> None:
    super().__init__(query_model=query_model, k=k,
                     handle_incomplete_batches=handle_incomplete_batches,
                     cache_packed_blocks=cache_packed_blocks)
    self._group = process_group
    self._merge = merge

  def index_from_dataset(self, candidates: Iterable, base_row: int = 0,
                         total_rows: Optional[int] = None) -> "ShardedStreaming":
    """``total_rows`` (the GLOBAL number of candidates, the same value on every rank): when it exceeds
    int32 the layer streams with shard-local row numbers and returns int64 global rows; without it a
    stream whose global rows pass 2^31 - 1 raises, as the int32 counter of the reference would wrap."""
    super().index_from_dataset(candidates)
    if base_row < 0:
      raise ValueError("base_row must be non-negative")
    self._wide = total_rows is not None and int(total_rows) > _INT32_MAX
    self._global_base = int(base_row)
    self._base_row = 0 if self._wide else int(base_row)
    return self

  def call(self, queries, k: Optional[int] = None):
    k = k if k is not None else self._k
    scores, rows = self._query_rows(queries, k)       # local shard; global row numbers unless wide
    if getattr(self, "_wide", False):
      return _exchange_and_merge_wide(scores, rows, k, self._group, self._merge, self._global_base)
    return _exchange_and_merge(scores, rows, k, self._group, self._merge)
