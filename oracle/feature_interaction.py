"""Oracle for Cross / MultiLayerDCN / DotInteraction (test infrastructure only).

Follows ``layers/feature_interaction/dcn.py:151-186``,
``multi_layer_dcn.py:136-153`` and ``dot_interaction.py:53-104``.  Pinned by
``dcn_test.py:29-50,68-101``, ``multi_layer_dcn_test.py:28-60`` and
``dot_interaction_test.py:25-64`` through ``tests/golden/feature_interaction.json``.
Keras ``Dense`` is ``act(x @ kernel + bias)`` with ``kernel[in, out]``
(SURVEY.md App. A.5).
"""

from typing import Callable, List, Optional, Sequence

import numpy as np


def cross(x0, x=None, kernel=None, bias=None, u=None, v=None,
          diag_scale: float = 0.0,
          preactivation: Optional[Callable[[np.ndarray], np.ndarray]] = None):
  """dcn.py:151-186.  Full rank: ``kernel[d, d]``; low rank: ``u[d, p]``,
  ``v[p, d]`` (:131-148).  float64 accumulation, float32 result."""
  x0 = np.asarray(x0, dtype=np.float32)
  x = x0 if x is None else np.asarray(x, dtype=np.float32)              # :167-168
  if x0.shape[-1] != x.shape[-1]:                                        # :170-174
    raise ValueError("`x0` and `x` dimension mismatch!")
  x64 = x.astype(np.float64)
  if kernel is not None:
    prod = x64 @ np.asarray(kernel, dtype=np.float64)                    # :176-177
  else:
    prod = (x64 @ np.asarray(u, dtype=np.float64)) @ np.asarray(v, dtype=np.float64)  # :178-179
  if bias is not None:
    prod = prod + np.asarray(bias, dtype=np.float64)
  if preactivation is not None:
    prod = preactivation(prod)
  if diag_scale:
    prod = prod + diag_scale * x64                                       # :183-184
  return (x0.astype(np.float64) * prod + x64).astype(np.float32)         # :186


def cross_grads(x0, x, kernel, bias, dy, diag_scale: float = 0.0):
  """Analytic backward of the full-rank, linear-preactivation cross (what
  ``tape.gradient`` gives): z = xW + b + diag*x;  y = x0*z + x."""
  x0 = np.asarray(x0, np.float64); x = np.asarray(x, np.float64)
  w = np.asarray(kernel, np.float64); dy = np.asarray(dy, np.float64)
  z = x @ w + (0 if bias is None else np.asarray(bias, np.float64)) + diag_scale * x
  dz = dy * x0
  dx0 = dy * z
  dx = dz @ w.T + diag_scale * dz + dy
  dw = x.T @ dz
  db = dz.sum(axis=0)
  return tuple(a.astype(np.float32) for a in (dx0, dx, dw, db))


def cross_yardsticks(x0, x, kernel, bias, dy, diag_scale: float = 0.0):
  """Sum of |terms| of every entry of (y, dx0, dx, dW, db) of the full-rank cross above: the
  scale its floating-point errors are measured in (tests/conftest.py float_gate)."""
  x0 = np.abs(np.asarray(x0, np.float64)); x = np.abs(np.asarray(x, np.float64))
  w = np.abs(np.asarray(kernel, np.float64)); dy = np.abs(np.asarray(dy, np.float64))
  z = x @ w + (0 if bias is None else np.abs(np.asarray(bias, np.float64))) + diag_scale * x
  dz = dy * x0
  return (x0 * z + x, dy * z, dz @ w.T + diag_scale * dz + dy, x.T @ dz, dz.sum(axis=0))


def dot_interaction_yardsticks(inputs, dy=None, self_interaction=False, skip_gather=False):
  """Sum of |terms| of every entry of the forward (and, given dy, of the backward)."""
  ab = [np.abs(np.asarray(a, np.float64)) for a in inputs]
  fwd = dot_interaction(ab, self_interaction, skip_gather).astype(np.float64)
  if dy is None:
    return fwd
  return fwd, dot_interaction_grad(ab, np.abs(np.asarray(dy, np.float64)), self_interaction,
                                   skip_gather).astype(np.float64)


def multi_layer_dcn(x0, us: Sequence[np.ndarray], vs: Sequence[np.ndarray],
                    biases: Optional[Sequence[Optional[np.ndarray]]] = None):
  """multi_layer_dcn.py:145-153: ``xl = x0 * (V_l(U_l xl) + b_l) + xl``."""
  x0_64 = np.asarray(x0, dtype=np.float64)
  xl = x0_64
  for i, (u, v) in enumerate(zip(us, vs)):
    prod = (xl @ np.asarray(u, np.float64)) @ np.asarray(v, np.float64)
    if biases is not None and biases[i] is not None:
      prod = prod + np.asarray(biases[i], np.float64)
    xl = x0_64 * prod + xl
  return xl.astype(np.float32)


def dot_interaction(inputs: List[np.ndarray], self_interaction: bool = False,
                    skip_gather: bool = False):
  """dot_interaction.py:69-104."""
  dims = {np.asarray(a).shape[1] for a in inputs}
  if len(dims) != 1:                                                     # :77-79
    raise ValueError("Input tensors` dimensions must be equal")
  num_features = len(inputs)
  b = np.asarray(inputs[0]).shape[0]
  d = dims.pop()
  x = np.concatenate([np.asarray(a, np.float32) for a in inputs], axis=-1
                     ).reshape(b, -1, d)                                 # :74-76
  xact = np.einsum("bfd,bgd->bfg", x.astype(np.float64), x.astype(np.float64))  # :82
  ones = np.ones_like(xact)
  if self_interaction:                                                   # :84-88
    lower = np.tril(ones)
    upper = ones - lower
    out_dim = num_features * (num_features + 1) // 2
  else:                                                                  # :89-93
    upper = np.triu(ones)
    lower = ones - upper
    out_dim = num_features * (num_features - 1) // 2
  if skip_gather:                                                        # :95-100
    act = np.where(upper.astype(bool), 0.0, xact)
    out_dim = num_features * num_features
  else:
    act = xact[lower.astype(bool)]                                       # :102 (row-major order)
  return act.reshape(b, out_dim).astype(np.float32)                      # :103


def dot_interaction_grad(inputs: List[np.ndarray], dy, self_interaction=False,
                         skip_gather=False):
  """Backward of the above wrt the concatenated ``[B, F, D]`` features:
  dX = (G + G^T) X with G the lower-triangular scatter of dy."""
  b = np.asarray(inputs[0]).shape[0]
  f = len(inputs)
  d = np.asarray(inputs[0]).shape[1]
  x = np.concatenate([np.asarray(a, np.float64) for a in inputs], axis=-1).reshape(b, f, d)
  g = np.zeros((b, f, f))
  mask = np.tril(np.ones((f, f)), 0 if self_interaction else -1).astype(bool)
  dy = np.asarray(dy, np.float64)
  if skip_gather:
    g = np.where(mask[None], dy.reshape(b, f, f), 0.0)
  else:
    g[:, mask] = dy
  return (np.einsum("bfg,bgd->bfd", g, x) + np.einsum("bgf,bgd->bfd", g, x)
          ).astype(np.float32)
