"""DLRM ``DotInteraction`` on MI355X.

Mirror of ``tensorflow_recommenders/layers/feature_interaction/dot_interaction.py:22-104``:
``DotInteraction(self_interaction=False, skip_gather=False)``, called on a list of
``[batch, dim]`` tensors; returns the row-major lower triangle of the per-sample Gram
matrix (or the ``F*F`` matrix with the upper part zeroed when ``skip_gather``).
"""

from typing import List, Optional

import torch

from recommenders_amd import _lib


def _out_dim(f: int, self_interaction: bool, skip_gather: bool) -> int:
  if skip_gather:
    return f * f
  return f * (f + 1) // 2 if self_interaction else f * (f - 1) // 2


class _DotInteractionFn(torch.autograd.Function):

  @staticmethod
  def forward(ctx, x, self_interaction, skip_gather):
    x = x.contiguous()
    b, f, d = x.shape
    out = torch.empty((b, _out_dim(f, self_interaction, skip_gather)), dtype=torch.float32,
                      device=x.device)
    _lib.check(_lib.load().tfrs_dot_interaction_fwd(
        _lib.ptr(x), b, f, d, int(self_interaction), int(skip_gather), _lib.ptr(out),
        _lib.current_stream()))
    ctx.save_for_backward(x)
    ctx.flags = (bool(self_interaction), bool(skip_gather))
    return out

  @staticmethod
  def backward(ctx, dout):
    (x,) = ctx.saved_tensors
    b, f, d = x.shape
    dx = torch.empty_like(x)
    dout = dout.contiguous()
    _lib.check(_lib.load().tfrs_dot_interaction_bwd(
        _lib.ptr(x), _lib.ptr(dout), b, f, d, int(ctx.flags[0]), int(ctx.flags[1]),
        _lib.ptr(dx), _lib.current_stream()))
    return dx, None, None


class DotInteraction(torch.nn.Module):

  def __init__(self, self_interaction: bool = False, skip_gather: bool = False,
               name: Optional[str] = None, **kwargs) -> None:
    super().__init__()
    self._self_interaction = self_interaction
    self._skip_gather = skip_gather
    self.name = name or "dot_interaction"

  def forward(self, inputs: List[torch.Tensor]) -> torch.Tensor:
    dims = {int(t.shape[1]) for t in inputs}
    if len(dims) != 1:                                                 # :73-79
      raise ValueError("Input tensors` dimensions must be equal, original"
                       f"error message: got feature dims {sorted(dims)}")
    batch, dim = inputs[0].shape
    # concat_features: [batch, num_features, feature_dim]  (:74-76)
    x = torch.cat([t.to(torch.float32) for t in inputs], dim=-1).reshape(batch, -1, dim)
    return _DotInteractionFn.apply(x, self._self_interaction, self._skip_gather)

  call = forward
