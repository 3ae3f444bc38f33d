"""Row-sharded embedding table across the GPUs of a node (BASELINE configs[4]:
100 tables x 10M rows x dim 32, table rows sharded over 8 MI355X).

Rank r owns rows ``[r * rows_per_rank, (r + 1) * rows_per_rank)`` of the ``[vocab, dim]`` table
(its ``embeddings`` parameter is only that shard).  A lookup of the local batch of ids is the
classic two-exchange pattern, one process per GPU, ``torch.distributed`` backend ``nccl``
(= RCCL over xGMI):

    ids --owner bucketing (HIP: count / scan / stable place, csrc/shard_route.hip)-->
        all_to_all (ids) --> local HIP gather --> all_to_all (rows) --> HIP gather through the
        inverse permutation (rows land in their final positions)

with ONE small collective and ONE device->host copy per lookup for the split sizes of both
all-to-alls (``torch.distributed`` needs them on the host).  That copy is the only thing that can
make the host wait for the device: ``stage(ids)`` moves it OFF the critical path -- called for the
NEXT batch right after the current step has been enqueued, it routes the ids, exchanges the counts
and starts their copy into pinned host memory without waiting; the lookup of that batch one step
later finds the counts already on the host (the "input dist one batch ahead" pipelining of
production recommenders).  Without ``stage`` the lookup does the same work inline and waits.  ``stage``
is explicit and collective: the lookup that follows must be of the staged tensor on every rank (anything
else raises rather than letting the ranks disagree on which collectives to run).

and the backward mirrors it: gradient rows travel to the owners (one all_to_all), where they
become ``(ids, rows)`` slices for ``optimizers.Adagrad`` (fused sparse update on the shard) or
a dense shard gradient.  Exchanged bytes per rank and step: ``B * 8`` (ids) + ``2 * B * dim * 4``
(rows forward, gradient rows backward).  Not part of the reference (its multi-device embedding
is TPUEmbedding); numerics are exactly those of ``Embedding``.

Backends without ``all_to_all`` (gloo, used by the CPU tests) fall back to an
all-gather-based emulation with identical semantics.
"""

from typing import Callable, List, Optional

import torch
import torch.distributed as dist

from recommenders_amd.layers import embedding as emb


def _world(group) -> int:
  return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def _single(group) -> bool:
  """One rank: nothing to exchange -- unless TFRS_FORCE_EXCHANGE=1 asks for the collectives anyway
  (a single-GPU box then executes the RCCL branch; results are unchanged)."""
  if _world(group) != 1:
    return False
  import os
  return not (os.environ.get("TFRS_FORCE_EXCHANGE", "0") == "1" and dist.is_available() and dist.is_initialized())


def _rank(group) -> int:
  return dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0


def all_to_all_v(send: torch.Tensor, send_counts: List[int], recv_counts: List[int],
                 group=None) -> torch.Tensor:
  """Variable-size all-to-all along dim 0: rank r receives ``recv_counts[p]`` leading rows from
  every peer p, in peer order."""
  world = _world(group)
  out = torch.empty((sum(recv_counts),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
  if _single(group):
    out.copy_(send)
    return out
  if dist.get_backend(group) == "nccl":
    dist.all_to_all_single(out, send.contiguous(), recv_counts, send_counts, group=group)
    return out
  # emulation for backends without all_to_all: everybody shares (counts, payload), picks its part
  gathered: List = [None] * world
  dist.all_gather_object(gathered, (list(send_counts), send.detach().cpu()), group=group)
  me = _rank(group)
  parts = []
  for counts, payload in gathered:
    lo = sum(counts[:me])
    parts.append(payload[lo:lo + counts[me]])
  out.copy_(torch.cat(parts, dim=0) if parts else out)
  return out


def _route_hip(flat: torch.Tensor, input_dim: int, rows_per_rank: int, world: int):
  """Owner bucketing on the device (``tfrs_shard_route_ids``: count, scan, stable placement -- no
  sort, no host synchronisation): ``(send_ids int64, perm int32, order int32, counts int64[world])``."""
  from recommenders_amd import _lib
  lib = _lib.load()
  n = flat.numel()
  dev = flat.device
  send_ids = torch.empty((n,), dtype=torch.int64, device=dev)
  perm = torch.empty((n,), dtype=torch.int32, device=dev)
  order = torch.empty((n,), dtype=torch.int32, device=dev)
  counts = torch.empty((world,), dtype=torch.int64, device=dev)
  ws = torch.empty((max(int(lib.tfrs_shard_route_workspace_bytes(n, world)), 256),), dtype=torch.uint8, device=dev)
  _lib.check(lib.tfrs_shard_route_ids(
      _lib.ptr(flat), 1 if flat.dtype == torch.int64 else 0, n, input_dim, rows_per_rank, world,
      _lib.ptr(send_ids), _lib.ptr(perm), _lib.ptr(order), _lib.ptr(counts), _lib.ptr(ws), ws.numel(),
      _lib.current_stream()))
  return send_ids, perm, order, counts


def _route_torch(flat: torch.Tensor, input_dim: int, rows_per_rank: int, world: int):
  """The same routing with torch ops.  TEST DOUBLE ONLY: reached when the layer was built with injected
  ``local_gather`` / ``local_scatter`` (the gloo tests of the exchange logic on boxes without a GPU);
  a native layer routes on the device and has no fallback."""
  flat = flat.long()
  bad = (flat < 0) | (flat >= input_dim)
  owner = torch.div(flat, rows_per_rank, rounding_mode="floor")
  owner = torch.where(bad, torch.zeros_like(owner), owner)
  order = torch.argsort(owner, stable=True)
  perm = torch.empty_like(order)
  perm[order] = torch.arange(order.numel(), device=order.device)
  send_ids = torch.where(bad, torch.full_like(flat, -1), flat - owner * rows_per_rank)[order]
  counts = torch.bincount(owner, minlength=world)
  return send_ids, perm.to(torch.int32), order.to(torch.int32), counts


class _Staged:
  """Routing of one batch of ids whose split sizes are (being) copied to the host."""

  __slots__ = ("ids", "version", "send_ids", "perm", "order", "host", "event", "lists")

  def counts(self, group):
    """``(send_counts, recv_counts)``; waits only for the copy started by ``stage`` (long finished
    when staged a step ahead)."""
    if self.lists is None:
      self.event.synchronize()
      world, me = _world(group), _rank(group)
      m = self.host.view(world, world).tolist()
      self.lists = ([int(v) for v in m[me]], [int(m[p][me]) for p in range(world)])
    return self.lists


def _exchange_counts(counts: torch.Tensor, group):
  """Split sizes of both all-to-alls from ONE collective and ONE device->host copy: every rank
  contributes its ``counts[world]`` row to the ``[world, world]`` matrix (round 2 paid two host
  synchronisations and a separate all-to-all of the counts per lookup)."""
  world, me = _world(group), _rank(group)
  if _single(group):
    return None, None                           # nothing to exchange: no synchronisation at all
  if dist.get_backend(group) == "nccl":
    matrix = torch.empty((world * world,), dtype=torch.int64, device=counts.device)
    dist.all_gather_into_tensor(matrix, counts.contiguous(), group=group)
    m = matrix.view(world, world).tolist()      # the one host synchronisation of the lookup
  else:
    rows = [None] * world
    dist.all_gather_object(rows, counts.tolist(), group=group)
    m = rows
  return [int(v) for v in m[me]], [int(m[p][me]) for p in range(world)]


class _ShardedLookup(torch.autograd.Function):

  @staticmethod
  def forward(ctx, shard, ids, layer):
    group, world = layer._group, _world(layer._group)
    flat = ids.reshape(-1)
    if flat.dtype not in (torch.int32, torch.int64):
      flat = flat.long()
    flat = flat.contiguous()
    # ids outside [0, input_dim) read as a zero row and receive no gradient, like the plain
    # gather kernel; they are routed to rank 0 as row -1 so that every rank's split sizes stay
    # consistent (an unchecked owner >= world would desynchronise the all_to_all and hang)
    staged = layer._take_staged(ids)
    if staged is not None:
      send_ids, perm, order = staged.send_ids, staged.perm, staged.order
      send_counts, recv_counts = staged.counts(group)
    else:
      # (the torch routing only exists for layers built with injected local_gather / local_scatter
      # doubles -- the gloo tests on boxes without a GPU; a native layer always routes on the device)
      route = _route_hip if layer._native else _route_torch
      send_ids, perm, order, counts = route(flat, layer.input_dim, layer.rows_per_rank, world)
      send_counts, recv_counts = _exchange_counts(counts, group)
    if send_counts is None:
      recv_ids = send_ids
    else:
      recv_ids = all_to_all_v(send_ids, send_counts, recv_counts, group)   # shard-local rows I serve
    rows = layer._gather(shard, recv_ids)                             # HIP gather on the owner
    back = rows if send_counts is None else all_to_all_v(rows, recv_counts, send_counts, group)
    # rows arrive in send-slot order; lookup i sits in slot perm[i]: one gather through perm puts
    # every row in its final position (no argsort, no index_put)
    out = layer._gather(back, perm)
    ctx.save_for_backward(order, recv_ids)
    ctx.counts = (send_counts, recv_counts)
    ctx.layer = layer
    ctx.table_ref = shard
    ctx.vocab = shard.shape[0]
    return out.reshape(tuple(ids.shape) + (shard.shape[1],))

  @staticmethod
  def backward(ctx, grad_out):
    order, recv_ids = ctx.saved_tensors
    send_counts, recv_counts = ctx.counts
    layer = ctx.layer
    g = grad_out.reshape(-1, grad_out.shape[-1]).contiguous()
    g = layer._gather(g, order)                                       # gradient rows in send-slot order
    recv_g = g if send_counts is None else all_to_all_v(g, send_counts, recv_counts, layer._group)
    table = ctx.table_ref
    if getattr(table, "_tfrs_sparse_grad", False):                    # slices for optimizers.Adagrad
      table._tfrs_slices.append((recv_ids, recv_g))
      return None, None, None
    return layer._scatter(recv_g, recv_ids, ctx.vocab), None, None


class ShardedEmbedding(torch.nn.Module):
  """``Embedding(input_dim, output_dim)`` whose rows are sharded over the process group."""

  def __init__(self, input_dim: int, output_dim: int, process_group=None,
               device: Optional[torch.device] = None,
               local_gather: Optional[Callable] = None, local_scatter: Optional[Callable] = None):
    super().__init__()
    self.input_dim, self.output_dim = int(input_dim), int(output_dim)
    self._group = process_group
    world, rank = _world(process_group), _rank(process_group)
    self.rows_per_rank = (self.input_dim + world - 1) // world
    lo = min(self.input_dim, rank * self.rows_per_rank)
    hi = min(self.input_dim, lo + self.rows_per_rank)
    self.row_range = (lo, hi)
    dev = device if device is not None else (
        torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu"))
    w = torch.empty((max(hi - lo, 1), self.output_dim), dtype=torch.float32, device=dev)
    w.uniform_(-0.05, 0.05)
    self.embeddings = torch.nn.Parameter(w)
    self.embeddings._tfrs_embedding = True
    self.embeddings._tfrs_row_sharded = True   # rank-local rows: excluded from the DP gradient sum
    # injection points: a CPU convenience so the exchange logic can be exercised under gloo on boxes
    # without a GPU (the tests then also take the torch routing); with a GPU the tests run the HIP
    # routing, gather and scatter
    self._native = local_gather is None and local_scatter is None
    self._gather = local_gather if local_gather is not None else emb.gather_rows
    self._scatter = local_scatter if local_scatter is not None else emb.scatter_add_rows

  # -- split sizes one batch ahead -----------------------------------------------------------------
  def stage(self, ids: torch.Tensor) -> None:
    """Routes ``ids`` (a batch that will be looked up LATER, typically the next one) and starts the
    exchange + device->host copy of the all-to-all split sizes without waiting for them.  A COLLECTIVE:
    every rank calls it, or none.  The next lookup must then be of the same tensor object, unmodified, on
    every rank -- it takes the staged routing and issues no collective for the split sizes.  A lookup of
    anything else RAISES instead of falling back to the inline exchange: the choice would be made per rank
    (object identity, version counter), and one rank taking the inline collective while its peers take the
    staged path hangs the job.  ``unstage()`` (every rank) or staging another batch drops a staged batch
    that will not be looked up.  World size 1: no-op."""
    group, world = self._group, _world(self._group)
    self._staged = None
    if _single(group) or not isinstance(ids, torch.Tensor) or ids.device != self.embeddings.device:
      return
    flat = ids.reshape(-1)
    if flat.dtype not in (torch.int32, torch.int64):
      return
    flat = flat.contiguous()
    st = _Staged()
    st.ids, st.version, st.lists = ids, ids._version, None
    route = _route_hip if self._native else _route_torch
    st.send_ids, st.perm, st.order, counts = route(flat, self.input_dim, self.rows_per_rank, world)
    if dist.get_backend(group) == "nccl":
      matrix = torch.empty((world * world,), dtype=torch.int64, device=counts.device)
      dist.all_gather_into_tensor(matrix, counts.contiguous(), group=group)
      st.host = torch.empty((world * world,), dtype=torch.int64, pin_memory=True)
      st.host.copy_(matrix, non_blocking=True)
      st.event = torch.cuda.Event()
      st.event.record()
    else:                                   # host-side collective (gloo): the counts are here already
      rows = [None] * world
      dist.all_gather_object(rows, counts.tolist(), group=group)
      me = _rank(group)
      st.host, st.event = None, None
      st.lists = ([int(v) for v in rows[me]], [int(rows[p][me]) for p in range(world)])
    self._staged = st

  def unstage(self) -> None:
    """Drops a staged batch that will not be looked up (call on every rank, like ``stage``)."""
    self._staged = None

  def _take_staged(self, ids) -> Optional[_Staged]:
    st, self._staged = getattr(self, "_staged", None), None
    if st is None:
      return None
    if st.ids is not ids or ids._version != st.version:
      raise RuntimeError(
          "ShardedEmbedding: a batch was staged with stage(ids) but the lookup is of "
          + ("another tensor" if st.ids is not ids else "the staged tensor after an in-place write")
          + ". The staged split sizes cannot be used and the inline exchange is a collective the other "
          "ranks would not join; look up the staged tensor unmodified, or call unstage() on every rank "
          "first. (The staged batch has been dropped on this rank.)")
    return st

  def forward(self, ids: torch.Tensor) -> torch.Tensor:
    if not isinstance(ids, torch.Tensor):
      ids = torch.as_tensor(ids)
    if ids.device != self.embeddings.device:
      ids = ids.to(self.embeddings.device)
    return _ShardedLookup.apply(self.embeddings, ids, self)
