"""DCN ``Cross`` layer on MI355X.

Mirror of ``tensorflow_recommenders/layers/feature_interaction/dcn.py:23-208``: same
constructor arguments, ``call(x0, x=None)``, error messages and ``get_config``.
``x_{i+1} = x0 * (W x_i + bias + diag_scale * x_i) + x_i`` runs as ONE fused kernel
(f32-MFMA GEMM with the cross formula in its epilogue, ``tfrs_cross_fwd``); the
low-rank form runs the ``U`` projection through ``tfrs_dense_fwd`` and the ``V``
projection through the same fused epilogue (``tfrs_cross_fwd_ex``).
"""

import math
import os
from typing import Callable, Optional, Union

import torch

from recommenders_amd import _lib

_ACTIVATIONS = {
    None: None, "linear": None, "relu": torch.relu, "sigmoid": torch.sigmoid,
    "tanh": torch.tanh, "swish": torch.nn.functional.silu, "silu": torch.nn.functional.silu,
    "gelu": torch.nn.functional.gelu,
}


def _get_activation(spec):
  if callable(spec):
    return spec
  if spec in _ACTIVATIONS:
    return _ACTIVATIONS[spec]
  raise ValueError(f"Unknown activation: {spec!r}")


# Activations the GEMM epilogues apply themselves (include/tfrs_hip.h TFRS_ACT_*): a Keras activation NAME runs
# fused in the product's epilogue, its derivative in the backward's one element-wise pass
# (tfrs_act_pointwise_bwd); a user CALLABLE is opaque and keeps the torch route.
_ACT_CODES = {None: 0, "linear": 0, "relu": 1, "sigmoid": 2, "tanh": 3, "swish": 4, "silu": 4, "gelu": 5}
_ACT_FROM_OUTPUT = (1, 2, 3)      # relu / sigmoid / tanh: act'(p) follows from y = act(p): y is what is saved


def activation_code(spec) -> Optional[int]:
  """The library's code of a Keras activation name; ``None`` for a callable (not fusable)."""
  if callable(spec):
    return None
  if spec in _ACT_CODES:
    return _ACT_CODES[spec]
  raise ValueError(f"Unknown activation: {spec!r}")


def _initialize(spec: Union[str, Callable], shape, device) -> torch.Tensor:
  """Keras initialiser names used by the reference: truncated_normal (stddev 0.05,
  resampled beyond 2 sigma), zeros, ones, glorot_uniform."""
  t = torch.empty(shape, dtype=torch.float32, device=device)
  if callable(spec):
    return spec(t)
  if spec == "truncated_normal":
    return torch.nn.init.trunc_normal_(t, mean=0.0, std=0.05, a=-0.1, b=0.1)
  if spec == "zeros":
    return t.zero_()
  if spec == "ones":
    return t.fill_(1.0)
  if spec == "glorot_uniform":
    fan = sum(shape) if len(shape) == 2 else shape[0]
    lim = math.sqrt(6.0 / fan)
    return t.uniform_(-lim, lim)
  raise ValueError(f"Unknown initializer: {spec!r}")


# Products of at least this many multiply-adds run on the split-fp16 GEMM (f32-grade results on
# the fp16 matrix cores, csrc/gemm16.hip); smaller ones, where the operand conversion and the
# 128 x 128 tiling do not pay, on the f32-MFMA kernel.  TFRS_GEMM_MODE=f32 forces the latter.
_F16_GEMM_MIN_MACS = 1 << 32


# The split-fp16 GEMM keeps its operand images in a caller-provided workspace (gigabytes at the
# ranking-model shapes).  One grow-only buffer per (device, stream) is reused: a fresh
# torch.empty per call made the caching allocator go back to hipMalloc whenever a forward pass
# followed a training step with a different allocation pattern (+16 ms per DLRM forward).
# Reuse is safe because every user of the buffer is enqueued on that same stream.
_GEMM_WS = {}


def _gemm_workspace(nbytes: int, device: torch.device) -> torch.Tensor:
  key = (device, torch.cuda.current_stream(device).cuda_stream)
  ws = _GEMM_WS.get(key)
  if ws is None or ws.numel() < nbytes:
    _GEMM_WS.pop(key, None)
    ws = None                      # release the old buffer before growing
    ws = torch.empty((int(nbytes * 1.25),), dtype=torch.uint8, device=device)
    _GEMM_WS[key] = ws
  return ws


def _use_f16_gemm(m: int, n: int, k: int) -> bool:
  if os.environ.get("TFRS_GEMM_MODE", "") == "f32":
    return False
  if os.environ.get("TFRS_GEMM_MODE", "") == "f16":
    return True
  if min(m, n, k) < 128:
    return False
  macs = m * n * k
  # measured (tools/exp_dense_shapes.py): 2x at >= 2^36, break-even near 2^32, but 3.6x already at
  # 2^31 for the long-K products of weight gradients (m, n small; k = batch)
  return macs >= _F16_GEMM_MIN_MACS or (macs >= (1 << 28) and k >= 4096)


def dense(x: torch.Tensor, kernel: torch.Tensor, bias: Optional[torch.Tensor] = None
          ) -> torch.Tensor:
  """``x @ kernel + bias`` (Keras Dense layout ``[in, out]``) via ``tfrs_dense_fwd`` /
  ``tfrs_dense_fwd_f16``."""
  x = x.contiguous()
  kernel = kernel.contiguous()
  out = torch.empty((x.shape[0], kernel.shape[1]), dtype=torch.float32, device=x.device)
  m, k, n = x.shape[0], kernel.shape[0], kernel.shape[1]
  if _use_f16_gemm(m, n, k):
    lib = _lib.load()
    ws = _gemm_workspace(lib.tfrs_gemm_f16_workspace_bytes(m, n, k), x.device)
    _lib.check(lib.tfrs_dense_fwd_f16(
        _lib.ptr(x), _lib.ptr(kernel), _lib.ptr(bias), m, k, n, _lib.ptr(out), _lib.ptr(ws),
        ws.numel(), _lib.current_stream()))
    return out
  _lib.check(_lib.load().tfrs_dense_fwd(
      _lib.ptr(x), _lib.ptr(kernel), _lib.ptr(bias), x.shape[0], kernel.shape[0],
      kernel.shape[1], _lib.ptr(out), _lib.current_stream()))
  return out


def dense_backward(x: torch.Tensor, kernel: torch.Tensor, dy: torch.Tensor, need_dx: bool = True,
                   need_dk: bool = True, need_db: bool = False):
  """Gradients of ``x @ kernel + bias`` through ``tfrs_dense_bwd``: ``dx = dy @ kernel^T``,
  ``dkernel = x^T @ dy``, ``dbias = sum_rows dy`` -- the transposed operands are read in place
  (no ``.t().contiguous()`` copies of activations)."""
  x, kernel, dy = x.contiguous(), kernel.contiguous(), dy.contiguous()
  m, k, n = x.shape[0], kernel.shape[0], kernel.shape[1]
  # (an empty batch returns from the C entry point before any launch: the weight gradients of
  # an empty batch are zeros, not uninitialised memory)
  alloc = torch.zeros_like if m == 0 else torch.empty_like
  dx = torch.empty_like(x) if need_dx else None
  dk = alloc(kernel) if need_dk else None
  db = (torch.zeros if m == 0 else torch.empty)((n,), dtype=torch.float32, device=x.device) if need_db else None
  lib = _lib.load()
  f16 = 1 if _use_f16_gemm(m, n, k) else 0
  ws = _gemm_workspace(lib.tfrs_dense_bwd_workspace_bytes(m, k, n, f16), x.device)
  _lib.check(lib.tfrs_dense_bwd(_lib.ptr(x), _lib.ptr(kernel), _lib.ptr(dy), m, k, n, _lib.ptr(dx),
                                _lib.ptr(dk), _lib.ptr(db), f16, _lib.ptr(ws), ws.numel(),
                                _lib.current_stream()))
  return dx, dk, db


class _DenseFn(torch.autograd.Function):
  """Dense with gradients, every GEMM on the HIP kernels."""

  @staticmethod
  def forward(ctx, x, kernel, bias):
    ctx.save_for_backward(x, kernel)
    ctx.has_bias = bias is not None
    return dense(x, kernel, bias)

  @staticmethod
  def backward(ctx, dy):
    x, kernel = ctx.saved_tensors
    return dense_backward(x, kernel, dy, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                          ctx.has_bias and ctx.needs_input_grad[2])


def _pointwise_bwd(act: int, ref_is_output: bool, ref, dy, x0=None, x=None, diag: float = 0.0,
                  want_dp=True, want_dx0=False, want_dxd=False):
  """``tfrs_act_pointwise_bwd``: dp = dy * x0 * act'(ref), dx0 = dy * (act(ref) + diag * x),
  dxd = dy * (1 + diag * x0) in one pass (each optional)."""
  dy = dy.contiguous()
  dp = torch.empty_like(dy) if want_dp else None
  dx0 = torch.empty_like(dy) if want_dx0 else None
  dxd = torch.empty_like(dy) if want_dxd else None
  _lib.check(_lib.load().tfrs_act_pointwise_bwd(
      int(act), 1 if ref_is_output else 0, _lib.ptr(ref), _lib.ptr(dy), _lib.ptr(x0), _lib.ptr(x), float(diag),
      dy.numel(), _lib.ptr(dp), _lib.ptr(dx0), _lib.ptr(dxd), _lib.current_stream()))
  return dp, dx0, dxd


class _DenseActFn(torch.autograd.Function):
  """``act(x @ kernel + bias)`` with the activation in the product's epilogue (``tfrs_dense_fwd_act``); backward:
  one element-wise pass ``dz = dy * act'`` (from the saved output for relu / sigmoid / tanh, else from the saved
  pre-activation), then the two products of ``tfrs_dense_bwd``."""

  @staticmethod
  def forward(ctx, x, kernel, bias, act):
    x, kernel = x.contiguous(), kernel.contiguous()
    m, k, n = x.shape[0], kernel.shape[0], kernel.shape[1]
    out = torch.empty((m, n), dtype=torch.float32, device=x.device)
    need_grad = any(ctx.needs_input_grad[:3])
    from_output = act in _ACT_FROM_OUTPUT
    pre = torch.empty_like(out) if (need_grad and not from_output) else None
    lib = _lib.load()
    f16 = 1 if _use_f16_gemm(m, n, k) else 0
    ws = _gemm_workspace(lib.tfrs_gemm_f16_workspace_bytes(m, n, k), x.device) if f16 else None
    _lib.check(lib.tfrs_dense_fwd_act(
        _lib.ptr(x), _lib.ptr(kernel), _lib.ptr(bias), m, k, n, int(act), _lib.ptr(out), _lib.ptr(pre), f16,
        _lib.ptr(ws), ws.numel() if ws is not None else 0, _lib.current_stream()))
    ctx.save_for_backward(x, kernel, out if from_output else pre)
    ctx.act, ctx.from_output, ctx.has_bias = int(act), from_output, bias is not None
    return out

  @staticmethod
  def backward(ctx, dy):
    x, kernel, ref = ctx.saved_tensors
    dz, _, _ = _pointwise_bwd(ctx.act, ctx.from_output, ref, dy)
    dx, dk, db = dense_backward(x, kernel, dz, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                ctx.has_bias and ctx.needs_input_grad[2])
    return dx, dk, db, None


def dense_act(x: torch.Tensor, kernel: torch.Tensor, bias: Optional[torch.Tensor], act: int) -> torch.Tensor:
  """Keras ``Dense(activation=name)`` on the HIP kernels: plain product for ``act == 0``."""
  if act == 0:
    return _DenseFn.apply(x, kernel, bias)
  return _DenseActFn.apply(x, kernel, bias, act)


class _CrossActFn(torch.autograd.Function):
  """``y = x0 * (act(a @ K + b) + diag * x) + x`` (``tfrs_cross_fwd_act``; dcn.py:173-186): ``a`` is ``x`` itself
  (full rank, ``K = W``) or ``h = x @ U`` computed by the caller (low rank, ``K = V``).  The forward's epilogue also
  stores ``p = a @ K + b``; the backward is one element-wise pass -- dp = dy * x0 * act'(p),
  dx0 = dy * (act(p) + diag * x), dxd = dy * (1 + diag * x0) -- and the products of ``tfrs_dense_bwd``:
  full rank ``dx = dp K^T + dxd`` (the addend rides in the product's epilogue), ``dK = x^T dp``; low rank
  ``dh = dp K^T``, ``dK = h^T dp`` and ``dxd`` is the gradient of ``x``'s direct terms."""

  @staticmethod
  def forward(ctx, x0, x, a, kernel, bias, diag, act, full_rank):
    x0, x, kernel = x0.contiguous(), x.contiguous(), kernel.contiguous()
    a = x if full_rank else a.contiguous()
    b, d, ka = x0.shape[0], x0.shape[1], kernel.shape[0]
    y = torch.empty_like(x0)
    need_grad = any(ctx.needs_input_grad[:5])
    pre = torch.empty_like(x0) if need_grad else None
    lib = _lib.load()
    f16 = 1 if _use_f16_gemm(b, d, ka) else 0
    ws = _gemm_workspace(lib.tfrs_gemm_f16_workspace_bytes(b, d, ka), x0.device) if f16 else None
    _lib.check(lib.tfrs_cross_fwd_act(
        _lib.ptr(x0), _lib.ptr(x), _lib.ptr(a), ka, _lib.ptr(kernel), _lib.ptr(bias), float(diag), int(act),
        b, d, _lib.ptr(y), _lib.ptr(pre), f16, _lib.ptr(ws), ws.numel() if ws is not None else 0,
        _lib.current_stream()))
    ctx.save_for_backward(x0, x, a, kernel, pre)
    ctx.diag, ctx.act, ctx.full_rank, ctx.has_bias = float(diag), int(act), bool(full_rank), bias is not None
    return y

  @staticmethod
  def backward(ctx, dy):
    x0, x, a, kernel, pre = ctx.saved_tensors
    # inputs: (x0, x, a, kernel, bias, ...).  Only the gradients autograd asks for are computed.
    need_x0, need_x, need_a, need_k, need_b = ctx.needs_input_grad[:5]
    need_b = need_b and ctx.has_bias
    if ctx.full_rank:
      need_a = need_x                   # a IS x: da = dp K^T + dxd is the whole gradient of x
    need_dp = need_a or need_k or need_b
    need_dxd = need_x
    dp, dx0, dxd = _pointwise_bwd(ctx.act, False, pre, dy, x0, x, ctx.diag, need_dp, need_x0, need_dxd)
    b, d, ka = x0.shape[0], x0.shape[1], kernel.shape[0]
    da = dk = db = None
    if need_dp:
      lib = _lib.load()
      f16 = 1 if _use_f16_gemm(b, d, ka) else 0      # (m, n, k) of p = a @ kernel, as dense_backward passes them
      zeros = b == 0
      if need_a:
        da = torch.empty((b, ka), dtype=torch.float32, device=x0.device)
      if need_k:
        dk = (torch.zeros_like if zeros else torch.empty_like)(kernel)
      if need_b:
        db = (torch.zeros if zeros else torch.empty)((d,), dtype=torch.float32, device=x0.device)
      ws = _gemm_workspace(lib.tfrs_dense_bwd_workspace_bytes(b, ka, d, f16), x0.device)
      _lib.check(lib.tfrs_dense_bwd_add(
          _lib.ptr(a), _lib.ptr(kernel), _lib.ptr(dp), _lib.ptr(dxd) if ctx.full_rank else None, b, ka, d,
          _lib.ptr(da), _lib.ptr(dk), _lib.ptr(db), f16, _lib.ptr(ws), ws.numel(), _lib.current_stream()))
    if ctx.full_rank:
      return dx0, da, None, dk, db, None, None, None
    return dx0, dxd, da, dk, db, None, None, None


class _CrossFn(torch.autograd.Function):
  """Fused full-rank cross:  y = x0 * (x @ W + b + diag * x) + x."""

  @staticmethod
  def forward(ctx, x0, x, kernel, bias, diag):
    x0, x, kernel = x0.contiguous(), x.contiguous(), kernel.contiguous()
    y = torch.empty_like(x0)
    b, d = x0.shape
    u = None
    if _use_f16_gemm(b, d, d):
      lib = _lib.load()
      ws = _gemm_workspace(lib.tfrs_gemm_f16_workspace_bytes(b, d, d), x0.device)
      if any(ctx.needs_input_grad[:4]):
        # training: the epilogue also stores u = x W + b + diag x, which the backward multiplies
        # dy by -- 4 b d bytes instead of recomputing the product
        u = torch.empty_like(x0)
        _lib.check(lib.tfrs_cross_fwd_f16_train(
            _lib.ptr(x0), _lib.ptr(x), _lib.ptr(kernel), _lib.ptr(bias), float(diag), b, d,
            _lib.ptr(y), _lib.ptr(u), _lib.ptr(ws), ws.numel(), _lib.current_stream()))
      else:
        _lib.check(lib.tfrs_cross_fwd_f16(
            _lib.ptr(x0), _lib.ptr(x), _lib.ptr(kernel), _lib.ptr(bias), float(diag), b, d,
            _lib.ptr(y), _lib.ptr(ws), ws.numel(), _lib.current_stream()))
    else:
      _lib.check(_lib.load().tfrs_cross_fwd(
          _lib.ptr(x0), _lib.ptr(x), _lib.ptr(kernel), _lib.ptr(bias), float(diag),
          b, d, _lib.ptr(y), _lib.current_stream()))
    ctx.save_for_backward(x0, x, kernel, bias, u)
    ctx.diag = float(diag)
    return y

  @staticmethod
  def backward(ctx, dy):
    """One ``tfrs_cross_bwd[_f16[_saved]]`` call: dx0 = dy * u (u = x W + b + diag x: saved by the
    split-fp16 training forward, else recomputed in the first GEMM's epilogue),
    dx = dz W^T + dy + diag dz, dW = x^T dz, db = sum_rows dz with dz = dy * x0 formed in the
    operand loads -- no transposes, no dz in HBM."""
    x0, x, kernel, bias, u = ctx.saved_tensors
    dy = dy.contiguous()
    b, d = x0.shape
    alloc = torch.zeros_like if b == 0 else torch.empty_like     # empty batch: zero weight gradients
    dx0, dx, dk = torch.empty_like(x0), torch.empty_like(x), alloc(kernel)
    db = alloc(bias) if bias is not None else None
    lib = _lib.load()
    f16 = 1 if (u is not None or _use_f16_gemm(b, d, d)) else 0
    ws = _gemm_workspace(lib.tfrs_cross_bwd_workspace_bytes(b, d, f16), x0.device)
    if u is not None:
      _lib.check(lib.tfrs_cross_bwd_f16_saved(
          _lib.ptr(x0), _lib.ptr(x), _lib.ptr(u), _lib.ptr(kernel), ctx.diag, _lib.ptr(dy), b, d,
          _lib.ptr(dx0), _lib.ptr(dx), _lib.ptr(dk), _lib.ptr(db), _lib.ptr(ws), ws.numel(),
          _lib.current_stream()))
      return dx0, dx, dk, db, None
    fn = lib.tfrs_cross_bwd_f16 if f16 else lib.tfrs_cross_bwd
    _lib.check(fn(_lib.ptr(x0), _lib.ptr(x), _lib.ptr(kernel), _lib.ptr(bias), ctx.diag,
                  _lib.ptr(dy), b, d, _lib.ptr(dx0), _lib.ptr(dx), _lib.ptr(dk), _lib.ptr(db),
                  _lib.ptr(ws), ws.numel(), _lib.current_stream()))
    return dx0, dx, dk, db, None


class _CrossStackFn(torch.autograd.Function):
  """A stack of fused full-rank cross layers on ONE ``x0`` (reference dcn.py:47-56: ``x1 = Cross()(x0, x0)``,
  ``x2 = Cross()(x0, x1)``, ...) as one autograd node.  Layer by layer the arithmetic is ``_CrossFn``'s, kernel for
  kernel; what changes is the gradient of ``x0``: every layer contributes ``dy_l * u_l`` (and the first layer, whose
  ``x`` IS ``x0``, its ``dx`` as well), which the autograd engine added up with ``[batch, d]`` torch additions -- 1.5 ms
  of a 54 ms DCN-v2 step at BASELINE configs[3].  Here the backward's element-wise pass accumulates in place
  (``tfrs_cross_bwd_f16_saved_acc``)."""

  @staticmethod
  def forward(ctx, x0, diags, *params):
    lib = _lib.load()
    x0 = x0.contiguous()
    b, d = x0.shape
    n = len(diags)
    ws = _gemm_workspace(lib.tfrs_gemm_f16_workspace_bytes(b, d, d), x0.device)
    xs, us = [x0], []
    kernels = [params[2 * l].contiguous() for l in range(n)]
    biases = [params[2 * l + 1] for l in range(n)]
    x = x0
    for l in range(n):
      y, u = torch.empty_like(x0), torch.empty_like(x0)
      _lib.check(lib.tfrs_cross_fwd_f16_train(
          _lib.ptr(x0), _lib.ptr(x), _lib.ptr(kernels[l]), _lib.ptr(biases[l]), float(diags[l]), b, d,
          _lib.ptr(y), _lib.ptr(u), _lib.ptr(ws), ws.numel(), _lib.current_stream()))
      us.append(u)
      if l + 1 < n:
        xs.append(y)
      x = y
    ctx.save_for_backward(*xs, *us, *kernels)
    ctx.n, ctx.diags, ctx.has_bias = n, tuple(float(v) for v in diags), tuple(bb is not None for bb in biases)
    return x

  @staticmethod
  def backward(ctx, dy):
    n = ctx.n
    saved = ctx.saved_tensors
    xs, us, kernels = saved[:n], saved[n:2 * n], saved[2 * n:3 * n]
    x0 = xs[0]
    b, d = x0.shape
    lib = _lib.load()
    ws = _gemm_workspace(lib.tfrs_cross_bwd_workspace_bytes(b, d, 1), x0.device)
    alloc = torch.zeros_like if b == 0 else torch.empty_like
    g = dy.contiguous()
    dx0 = torch.empty_like(x0)
    grads = [None] * (2 * n)
    for l in range(n - 1, -1, -1):
      dx, dk = torch.empty_like(x0), alloc(kernels[l])
      db = alloc(kernels[l][0]) if ctx.has_bias[l] else None
      _lib.check(lib.tfrs_cross_bwd_f16_saved_acc(
          _lib.ptr(x0), _lib.ptr(xs[l]), _lib.ptr(us[l]), _lib.ptr(kernels[l]), ctx.diags[l], _lib.ptr(g), b, d,
          _lib.ptr(dx0) if l < n - 1 else None, 1 if l == 0 else 0, _lib.ptr(dx0), _lib.ptr(dx), _lib.ptr(dk),
          _lib.ptr(db), _lib.ptr(ws), ws.numel(), _lib.current_stream()))
      grads[2 * l], grads[2 * l + 1] = dk, db
      g = dx
    return (dx0, None) + tuple(grads)


def cross_stack(x0: torch.Tensor, layers) -> Optional[torch.Tensor]:
  """``x = x0; for layer in layers: x = layer(x0, x)`` through ``_CrossStackFn`` when every layer is a plain full-rank
  ``Cross`` (no preactivation, no projection) of the width of ``x0``, the products take the split-fp16 path and a
  gradient is wanted; ``None`` otherwise (the caller runs the loop)."""
  layers = list(layers)
  if (len(layers) < 2 or not torch.is_grad_enabled() or x0.dim() != 2 or not x0.is_cuda
      or x0.dtype != torch.float32 or x0.shape[0] == 0):
    return None
  b, d = x0.shape
  for layer in layers:
    if type(layer) is not Cross or layer._projection_dim is not None or callable(layer._preactivation_spec):
      return None
    if activation_code(layer._preactivation_spec) != 0:
      return None
    if not layer.built:
      layer.build(x0.shape, x0.device)                                  # (as Cross.forward does, :161-162)
    if tuple(layer.kernel.shape) != (d, d) or layer.kernel.device != x0.device:
      return None
  if not _use_f16_gemm(b, d, d):
    return None
  params = []
  for layer in layers:
    params += [layer.kernel, layer.bias]
  if not (x0.requires_grad or any(p is not None and p.requires_grad for p in params)):
    return None
  return _CrossStackFn.apply(x0, tuple(float(layer._diag_scale or 0.0) for layer in layers), *params)


class Cross(torch.nn.Module):
  """Cross layer of the Deep & Cross Network (reference dcn.py:23-208)."""

  def __init__(self, projection_dim: Optional[int] = None, diag_scale: Optional[float] = 0.0,
               use_bias: bool = True, preactivation=None,
               kernel_initializer="truncated_normal", bias_initializer="zeros",
               kernel_regularizer=None, bias_regularizer=None, **kwargs):
    super().__init__()
    self._projection_dim = projection_dim
    self._diag_scale = diag_scale
    self._use_bias = use_bias
    self._preactivation_spec = preactivation
    self._preactivation = _get_activation(preactivation)
    self._kernel_initializer = kernel_initializer
    self._bias_initializer = bias_initializer
    self._kernel_regularizer = kernel_regularizer
    self._bias_regularizer = bias_regularizer
    self.name = kwargs.get("name", "cross")
    self.built = False
    if self._diag_scale < 0:                                           # :112-115
      raise ValueError(
          "`diag_scale` should be non-negative. Got `diag_scale` = {}".format(self._diag_scale))

  def build(self, input_shape, device=None) -> None:                   # :117-149
    last_dim = int(input_shape[-1])
    dev = device if device is not None else torch.device("cuda")
    if self._projection_dim is None:
      self.kernel = torch.nn.Parameter(
          _initialize(self._kernel_initializer, (last_dim, last_dim), dev))
    else:
      p = int(self._projection_dim)
      self.kernel_u = torch.nn.Parameter(_initialize(self._kernel_initializer, (last_dim, p), dev))
      self.kernel_v = torch.nn.Parameter(_initialize(self._kernel_initializer, (p, last_dim), dev))
    self.bias = (torch.nn.Parameter(_initialize(self._bias_initializer, (last_dim,), dev))
                 if self._use_bias else None)
    self.built = True

  def regularization_losses(self):
    """Keras ``layer.losses`` equivalent: regulariser callables applied to the weights."""
    out = []
    if self._kernel_regularizer is not None:
      for name in ("kernel", "kernel_u", "kernel_v"):
        if hasattr(self, name):
          out.append(self._kernel_regularizer(getattr(self, name)))
    if self._bias_regularizer is not None and self.bias is not None:
      out.append(self._bias_regularizer(self.bias))
    return out

  def forward(self, x0: torch.Tensor, x: Optional[torch.Tensor] = None) -> torch.Tensor:
    if not self.built:
      self.build(x0.shape, x0.device)                                  # :161-162
    if x is None:
      x = x0                                                           # :164-165
    if x0.shape[-1] != x.shape[-1]:                                    # :167-171
      raise ValueError(
          "`x0` and `x` dimension mismatch! Got `x0` dimension {}, and x "
          "dimension {}. This case is not supported yet.".format(x0.shape[-1], x.shape[-1]))
    x0 = x0.to(torch.float32)
    x = x.to(torch.float32)
    diag = float(self._diag_scale or 0.0)
    act = activation_code(self._preactivation_spec)
    if act == 0 and self._projection_dim is None:
      return _CrossFn.apply(x0, x, self.kernel, self.bias, diag)       # :173-186 fused, linear full rank
    if act is not None:
      # a named preactivation and / or the low-rank form: activation and cross formula in the product's epilogue
      if self._projection_dim is None:
        return _CrossActFn.apply(x0, x, None, self.kernel, self.bias, diag, act, True)
      h = _DenseFn.apply(x, self.kernel_u, None)                       # :176
      return _CrossActFn.apply(x0, x, h, self.kernel_v, self.bias, diag, act, False)
    # a user CALLABLE as preactivation is opaque: product on the HIP kernel, the callable and the cross
    # formula element-wise in torch (the reference's unfused order, :173-186)
    if self._projection_dim is None:
      prod = self._preactivation(_DenseFn.apply(x, self.kernel, self.bias))
    else:
      prod = self._preactivation(
          _DenseFn.apply(_DenseFn.apply(x, self.kernel_u, None), self.kernel_v, self.bias))
    if diag:
      prod = prod + diag * x
    return x0 * prod + x

  call = forward

  def get_config(self):                                                # :188-208
    return {
        "projection_dim": self._projection_dim,
        "diag_scale": self._diag_scale,
        "use_bias": self._use_bias,
        "preactivation": self._preactivation_spec,
        "kernel_initializer": self._kernel_initializer,
        "bias_initializer": self._bias_initializer,
        "kernel_regularizer": self._kernel_regularizer,
        "bias_regularizer": self._bias_regularizer,
        "name": self.name,
    }

  @classmethod
  def from_config(cls, config):
    return cls(**config)
