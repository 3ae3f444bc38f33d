"""``MultiLayerDCN`` on MI355X: N stacked low-rank cross layers.

Mirror of ``tensorflow_recommenders/layers/feature_interaction/multi_layer_dcn.py:37-173``
(``x_{l+1} = x0 * (V_l (U_l x_l) + b_l) + x_l``, :147-153), built from the same fused
kernels as ``Cross``.
"""

from typing import Optional

import torch

from recommenders_amd.layers.feature_interaction.dcn import (_CrossActFn, _DenseFn,
                                                             _initialize)


class MultiLayerDCN(torch.nn.Module):

  def __init__(self, projection_dim: Optional[int] = 1, num_layers: Optional[int] = 3,
               use_bias: bool = True, kernel_initializer="truncated_normal",
               bias_initializer="zeros", kernel_regularizer=None, bias_regularizer=None,
               **kwargs):
    super().__init__()
    self._projection_dim = projection_dim
    self._num_layers = num_layers
    self._use_bias = use_bias
    self._kernel_initializer = kernel_initializer
    self._bias_initializer = bias_initializer
    self._kernel_regularizer = kernel_regularizer
    self._bias_regularizer = bias_regularizer
    self.name = kwargs.get("name", "multi_layer_dcn")
    self.built = False

  def build(self, input_shape, device=None) -> None:                   # :112-134
    d = int(input_shape[-1])
    p = int(self._projection_dim)
    dev = device if device is not None else torch.device("cuda")
    self.u_kernels = torch.nn.ParameterList(
        [torch.nn.Parameter(_initialize(self._kernel_initializer, (d, p), dev))
         for _ in range(self._num_layers)])
    self.v_kernels = torch.nn.ParameterList(
        [torch.nn.Parameter(_initialize(self._kernel_initializer, (p, d), dev))
         for _ in range(self._num_layers)])
    self.biases = (torch.nn.ParameterList(
        [torch.nn.Parameter(_initialize(self._bias_initializer, (d,), dev))
         for _ in range(self._num_layers)]) if self._use_bias else None)
    self.built = True

  def forward(self, x0: torch.Tensor) -> torch.Tensor:                 # :136-153
    if not self.built:
      self.build(x0.shape, x0.device)
    x0 = x0.to(torch.float32)
    xl = x0
    for i in range(self._num_layers):
      h = _DenseFn.apply(xl, self.u_kernels[i], None)
      xl = _CrossActFn.apply(x0, xl, h, self.v_kernels[i],
                             self.biases[i] if self.biases is not None else None, 0.0, 0, False)
    return xl

  call = forward

  def get_config(self):                                                # :155-173
    return {
        "projection_dim": self._projection_dim,
        "num_layers": self._num_layers,
        "use_bias": self._use_bias,
        "kernel_initializer": self._kernel_initializer,
        "bias_initializer": self._bias_initializer,
        "kernel_regularizer": self._kernel_regularizer,
        "bias_regularizer": self._bias_regularizer,
        "name": self.name,
    }

  @classmethod
  def from_config(cls, config):
    return cls(**config)
