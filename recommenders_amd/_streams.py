"""Side branches of a captured step (round 6).

The README train step is a chain of ~12 short kernels; replayed from a HIP graph they run strictly one after the other
although the metric update (``FactorizedTopK``: rank counts over the candidate corpus + hits update, ~29 us of the
131 us step at the MovieLens shapes) depends only on the batch's embeddings, not on the loss or its backward.  Under
stream capture a piece of work wrapped in ``forked(device)`` is recorded on a second stream -- a parallel branch of the
graph -- and joined where the enclosing ``scope`` says so: ``tfrs.Model.train_step`` joins in front of the optimizer
step (the metric branch gathers candidate embeddings from the tables the optimizer is about to update) and at its end.

Outside a ``scope``, and outside stream capture, ``forked`` runs its body inline on the current stream: the eager step
is bound by host time, and a second stream there would need ``record_stream`` bookkeeping for every tensor that
crosses streams.  Inside a captured graph all memory comes from the capture's private pool, which lives as long as
the graph does.
"""

import contextlib
import os
import threading
from typing import Dict, List

import torch

_local = threading.local()
_side_streams: Dict[int, "torch.cuda.Stream"] = {}


def _side_stream(device: torch.device) -> "torch.cuda.Stream":
  idx = device.index if device.index is not None else torch.cuda.current_device()
  if idx not in _side_streams:
    _side_streams[idx] = torch.cuda.Stream(device=idx)
  return _side_streams[idx]


class scope:
  """``with scope() as branches: ... branches.join() ...``: side branches opened inside are joined into the current
  stream by ``join()`` and, at the latest, when the scope ends."""

  def __enter__(self) -> "scope":
    self._outer = getattr(_local, "scope", None)
    self._pending: List["torch.cuda.Stream"] = []
    _local.scope = self
    return self

  def join(self) -> None:
    cur = torch.cuda.current_stream() if self._pending else None
    for side in self._pending:
      cur.wait_stream(side)
    self._pending = []

  def __exit__(self, *exc) -> None:
    try:
      self.join()
    finally:
      _local.scope = self._outer


def _branching(device: torch.device):
  """The active scope if a side branch may be opened now (a scope is active, the device is a GPU, the current stream is
  being captured, not switched off), else None."""
  sc = getattr(_local, "scope", None)
  if (sc is None or device.type != "cuda" or os.environ.get("TFRS_STEP_BRANCHES", "1") == "0"
      or not torch.cuda.is_available() or not torch.cuda.is_current_stream_capturing()):
    return None
  return sc


def mark(device: torch.device):
  """A fork point: everything issued on the current stream so far.  ``forked(device, after=mark(device))`` later in
  program order opens a branch that depends on that prefix only -- the metric update of ``tasks.Retrieval`` is written
  behind the loss (reference order, tasks/retrieval.py:212-226) but needs the embeddings alone, so its branch forks in
  front of the loss kernels and runs beside them.  None outside a capturing scope."""
  if _branching(device) is None or os.environ.get("TFRS_STEP_BRANCHES", "1") == "2":   # 2: fork at the branch itself
    return None
  ev = torch.cuda.Event()
  ev.record(torch.cuda.current_stream())
  return ev


@contextlib.contextmanager
def forked(device: torch.device, after=None):
  """Runs the body on the device's side stream when a ``scope`` is active AND the current stream is being captured
  (the branch becomes a parallel path of the graph); inline otherwise (and always with ``TFRS_STEP_BRANCHES=0``,
  the A/B switch of tools/exp_trainstep_graph.py).  ``after``: a ``mark()`` taken earlier -- the branch then waits for
  that point of the current stream instead of for its head."""
  sc = _branching(device)
  if sc is None:
    yield
    return
  side = _side_stream(device)
  if after is not None:
    side.wait_event(after)
  else:
    side.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(side):
    yield
  if side not in sc._pending:
    sc._pending.append(side)
