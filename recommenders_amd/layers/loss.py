"""Logit post-processing of the retrieval loss (reference ``layers/loss.py``).

In the fused training path these transforms are folded into the softmax kernel's logit
function (``csrc/softmax.hip``: ``make_logit``).  The classes below keep the reference's
public layer names for code that applies them to an explicit ``[B, C]`` logits tensor
(e.g. batch metrics); the element-wise ones are torch glue, the top-k of ``HardNegativeMining`` runs the
library's selection kernel.
"""

from typing import Tuple

import numpy as np
import torch

MAX_FLOAT = float(np.finfo(np.float32).max / 100.0)   # loss.py:22
MIN_FLOAT = float(np.finfo(np.float32).min / 100.0)   # loss.py:23


_PAGE = 1024   # include/tfrs_hip.h TFRS_MAX_K: what one selection call returns per row


def _topk_columns(keyed: torch.Tensor, k: int) -> torch.Tensor:
  """Column indices of the ``k`` largest entries of every row (``tf.math.top_k(..., sorted=False)`` of
  loss.py:104-105; here sorted, ties to the lower column) through the library's score-block selection
  (``tfrs_topk_update_from_scores``) -- rows of any width and, like the reference, ANY ``k``: beyond the library's
  page of 1024 the selection runs page by page, the columns already taken pushed to -inf in a scratch copy (each
  page is the next 1024 of the same (value descending, column ascending) order, so the union is the top ``k``)."""
  import ctypes
  from recommenders_amd import _lib
  if not keyed.is_cuda:
    raise ValueError("HardNegativeMining: the logits must live on the GPU (this package has no CPU path; "
                     "the reference's layers/loss.py:61-111 is the CPU implementation)")
  keyed = keyed.detach().contiguous().to(torch.float32)
  nq, nc = keyed.shape
  if nc > 0x7FFFFFFF:
    raise ValueError("HardNegativeMining: more than 2^31 - 1 columns")
  k = min(k, nc)
  lib, stream = _lib.load(), _lib.current_stream()
  pages = []
  taken = 0
  while taken < k:
    kk = min(_PAGE, k - taken)
    vals = torch.empty((nq, kk), dtype=torch.float32, device=keyed.device)
    cols = torch.empty((nq, kk), dtype=torch.int32, device=keyed.device)
    new_len = ctypes.c_int32(0)
    _lib.check(lib.tfrs_topk_update_from_scores(_lib.ptr(keyed), nq, nc, nc, 0, kk, _lib.ptr(vals), _lib.ptr(cols), 0,
                                                ctypes.byref(new_len), stream))
    pages.append(cols.long())
    taken += kk
    if taken < k:
      if len(pages) == 1:
        # never the caller's tensor; -inf entries become the lowest finite float, so that the -inf written over the
        # columns already taken sorts strictly below everything still to come (ties between an original -inf and an
        # original -FLT_MAX -- the reference's masks use MIN_FLOAT = -FLT_MAX / 100 -- are the only order given up)
        keyed = keyed.clamp(min=-3.4028234663852886e38)
      keyed.scatter_(1, pages[-1], float("-inf"))
  return pages[0] if len(pages) == 1 else torch.cat(pages, dim=1)


class HardNegativeMining(torch.nn.Module):
  """Keeps the positive and the ``num_hard_negatives`` highest negatives per row
  (loss.py:61-111)."""

  def __init__(self, num_hard_negatives: int) -> None:
    super().__init__()
    self._num_hard_negatives = num_hard_negatives

  def forward(self, logits: torch.Tensor, labels: torch.Tensor
              ) -> Tuple[torch.Tensor, torch.Tensor]:
    num_sampled = min(self._num_hard_negatives + 1, logits.shape[1])   # :91
    cols = _topk_columns(logits + labels * MAX_FLOAT, num_sampled)     # :104-105
    return torch.gather(logits, 1, cols), torch.gather(labels, 1, cols)       # :108-109


class RemoveAccidentalHits(torch.nn.Module):
  """Pushes logits of negatives that share the positive's id to MIN_FLOAT
  (loss.py:114-147)."""

  def forward(self, labels: torch.Tensor, logits: torch.Tensor,
              candidate_ids: torch.Tensor) -> torch.Tensor:
    ids = candidate_ids.reshape(-1)
    pos_ids = ids[torch.argmax(labels, dim=1)]                         # :139-140
    dup = (pos_ids.unsqueeze(1) == ids.unsqueeze(0)).to(labels.dtype) - labels   # :142-146
    return logits + dup * MIN_FLOAT                                    # :147


class SamplingProbablityCorrection(torch.nn.Module):
  """``logits - log(clip(p, 1e-6, 1))`` (loss.py:150-158)."""

  def forward(self, logits: torch.Tensor, candidate_sampling_probability: torch.Tensor
              ) -> torch.Tensor:
    return logits - torch.log(torch.clamp(candidate_sampling_probability, 1e-6, 1.0))
