"""Row-sharded embedding table across the GPUs of a node (BASELINE configs[4]:
100 tables x 10M rows x dim 32, table rows sharded over 8 MI355X).

Rank r owns rows ``[r * rows_per_rank, (r + 1) * rows_per_rank)`` of the ``[vocab, dim]`` table
(its ``embeddings`` parameter is only that shard).  A lookup of the local batch of ids is the
classic two-exchange pattern, one process per GPU, ``torch.distributed`` backend ``nccl``
(= RCCL over xGMI):

    ids --bucket by owner--> all_to_all (ids) --> local HIP gather --> all_to_all (rows) --> unpermute

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


def _rank(group) -> int:
  return dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0


def all_to_all_v(send: torch.Tensor, send_counts: List[int], recv_counts: List[int],
                 group=None) -> torch.Tensor:
  """Variable-size all-to-all along dim 0: rank r receives ``recv_counts[p]`` leading rows from
  every peer p, in peer order."""
  world = _world(group)
  out = torch.empty((sum(recv_counts),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
  if world == 1:
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


class _ShardedLookup(torch.autograd.Function):

  @staticmethod
  def forward(ctx, shard, ids, layer):
    group, world = layer._group, _world(layer._group)
    flat = ids.reshape(-1).long()
    # ids outside [0, input_dim) read as a zero row and receive no gradient, like the plain
    # gather kernel; they are routed to rank 0 as row -1 so that every rank's split sizes stay
    # consistent (an unchecked owner >= world would desynchronise the all_to_all and hang)
    bad = (flat < 0) | (flat >= layer.input_dim)
    owner = torch.div(flat, layer.rows_per_rank, rounding_mode="floor")
    owner = torch.where(bad, torch.zeros_like(owner), owner)
    flat = torch.where(bad, torch.full_like(flat, -1), flat)
    order = torch.argsort(owner, stable=True)
    send_counts = torch.bincount(owner, minlength=world).tolist()
    counts_t = torch.tensor(send_counts, dtype=torch.int64, device=flat.device)
    recv_counts = all_to_all_v(counts_t, [1] * world, [1] * world, group).tolist()
    send_ids = torch.where(bad, flat, flat - owner * layer.rows_per_rank)[order]   # shard-local rows
    recv_ids = all_to_all_v(send_ids, send_counts, recv_counts, group)
    rows = layer._gather(shard, recv_ids)                             # HIP gather on the owner
    back = all_to_all_v(rows, recv_counts, send_counts, group)        # rows in `order` order
    out = torch.empty_like(back)
    out[order] = back
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
    g = grad_out.reshape(-1, grad_out.shape[-1]).contiguous()[order]
    recv_g = all_to_all_v(g, send_counts, recv_counts, layer._group)  # gradient rows of MY shard
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
    # injection points so the exchange logic can be exercised on CPU (gloo) in tests
    self._gather = local_gather if local_gather is not None else emb.gather_rows
    self._scatter = local_scatter if local_scatter is not None else emb.scatter_add_rows

  def forward(self, ids: torch.Tensor) -> torch.Tensor:
    if not isinstance(ids, torch.Tensor):
      ids = torch.as_tensor(ids)
    return _ShardedLookup.apply(self.embeddings, ids.to(self.embeddings.device), self)
