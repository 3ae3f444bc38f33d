"""Convenience blocks (mirror of ``tensorflow_recommenders/layers/blocks.py:24-61``).

``MLP(units, use_bias=True, activation="relu", final_activation=None)`` is a stack of Keras
``Dense`` layers: ``y = act(x @ kernel + bias)`` with ``kernel`` in ``[in, out]`` layout,
``glorot_uniform`` kernel / ``zeros`` bias initialisers and lazy building on the first call.
Every matmul (forward and both backward GEMMs) runs on the HIP GEMM kernels; a named activation is applied
in the forward product's epilogue (``tfrs_dense_fwd_act``) and differentiated in one element-wise pass of the
backward (``tfrs_act_pointwise_bwd``); only a user callable stays a torch op.
"""

import math
from typing import Callable, List, Optional, Union

import torch

from recommenders_amd.layers.feature_interaction.dcn import _DenseFn, activation_code, dense_act

Activation = Optional[Union[str, Callable[[torch.Tensor], torch.Tensor]]]

# (names resolve to torch functions only for code that asks `get_activation` for a callable; `Dense` itself runs a
# named activation in its product's epilogue)
from recommenders_amd.layers.feature_interaction.dcn import _ACTIVATIONS as _DCN_ACTIVATIONS  # noqa: E402

_ACTIVATIONS = {k: (v if v is not None else (lambda x: x)) for k, v in _DCN_ACTIVATIONS.items()}


def get_activation(spec: Activation) -> Callable[[torch.Tensor], torch.Tensor]:
  if callable(spec):
    return spec
  if spec not in _ACTIVATIONS:
    raise ValueError(f"Unknown activation: {spec!r}")
  return _ACTIVATIONS[spec]


class Dense(torch.nn.Module):
  """``tf.keras.layers.Dense(units, activation=None, use_bias=True)``."""

  def __init__(self, units: int, activation: Activation = None, use_bias: bool = True,
               device: Optional[torch.device] = None):
    super().__init__()
    self.units = int(units)
    self.use_bias = use_bias
    self._activation = get_activation(activation)
    # a Keras activation NAME runs in the product's epilogue (tfrs_dense_fwd_act); a callable stays a torch op
    self._act_code = activation_code("linear" if activation is None else activation)
    self._device = device
    self.kernel: Optional[torch.nn.Parameter] = None
    self.bias: Optional[torch.nn.Parameter] = None

  def build(self, in_dim: int, device: torch.device) -> None:
    lim = math.sqrt(6.0 / (in_dim + self.units))          # glorot_uniform
    k = torch.empty((in_dim, self.units), dtype=torch.float32, device=device).uniform_(-lim, lim)
    self.kernel = torch.nn.Parameter(k)
    if self.use_bias:
      self.bias = torch.nn.Parameter(torch.zeros((self.units,), dtype=torch.float32, device=device))

  def forward(self, x: torch.Tensor) -> torch.Tensor:
    if self.kernel is None:
      self.build(x.shape[-1], self._device or x.device)
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1]).to(torch.float32)
    if self._act_code is not None:
      return dense_act(x2, self.kernel, self.bias, self._act_code).reshape(*lead, self.units)
    y = _DenseFn.apply(x2, self.kernel, self.bias)
    return self._activation(y).reshape(*lead, self.units)


class MLP(torch.nn.Module):
  """Sequential multi-layer perceptron block (reference blocks.py:24-61)."""

  def __init__(self, units: List[int], use_bias: bool = True, activation: Activation = "relu",
               final_activation: Activation = None):
    super().__init__()
    layers = [Dense(n, activation=activation, use_bias=use_bias) for n in units[:-1]]   # :46-49
    layers.append(Dense(units[-1], activation=final_activation, use_bias=use_bias))      # :50-52
    self._sublayers = torch.nn.ModuleList(layers)

  def forward(self, x: torch.Tensor) -> torch.Tensor:                                   # :54-59
    for layer in self._sublayers:
      x = layer(x)
    return x
