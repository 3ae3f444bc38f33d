"""Salted hashing of feature values into table buckets, on the GPU.

``Hashing(num_bins, salt=[s0, s1])`` restates ``tf.keras.layers.Hashing`` as
``UnifiedEmbedding`` configures it (``layers/feature_multiplexing/unified_embedding.py:
116-119,155-159,198-205``): the bucket of a value is SipHash-2-4 under the key ``(s0, s1)``
of its bytes -- the decimal string for integers (``tf.as_string``), the UTF-8 bytes for
strings -- modulo ``num_bins``.  Integer tensors are hashed where they live in HBM
(``tfrs_hash_bucket_strong_ids``); strings are packed on the host into one byte buffer +
offsets and hashed by ``tfrs_hash_bucket_strong_bytes``.  Output: int64 buckets of the
input's shape on the GPU (ragged inputs keep their row splits).

The unsalted Keras mode (FarmHash64) is not used on this path and raises.
"""

from typing import Optional, Sequence, Union

import numpy as np
import torch

from recommenders_amd import _lib
from recommenders_amd.layers.tpu_embedding_layer import RaggedIds, SparseIds

_U64 = (1 << 64) - 1


def _device(device: Optional[torch.device]) -> torch.device:
  if device is not None:
    return torch.device(device)
  if not torch.cuda.is_available():
    raise RuntimeError("recommenders_amd ops need a GPU; there is no CPU fallback.")
  return torch.device("cuda")


def pack_strings(arr: np.ndarray, dev: torch.device):
  """Strings (str -> UTF-8, or bytes) of any shape -> (byte blob, offsets[n + 1]) on ``dev``."""
  flat = np.asarray(arr).reshape(-1)
  enc = [x if isinstance(x, (bytes, np.bytes_)) else str(x).encode("utf-8") for x in flat]
  offsets = np.zeros(len(enc) + 1, dtype=np.int64)
  np.cumsum([len(e) for e in enc], out=offsets[1:])
  blob = (np.frombuffer(b"".join(enc), dtype=np.uint8).copy() if offsets[-1]
          else np.zeros(1, dtype=np.uint8))
  return torch.from_numpy(blob).to(dev), torch.from_numpy(offsets).to(dev)


class Hashing(torch.nn.Module):
  """``tf.keras.layers.Hashing(num_bins, salt=[s0, s1])``."""

  def __init__(self, num_bins: int, mask_value=None,
               salt: Optional[Union[int, Sequence[int]]] = None,
               device: Optional[torch.device] = None):
    super().__init__()
    if num_bins is None or int(num_bins) <= 0:
      raise ValueError(f"The `num_bins` for `Hashing` cannot be `None` or non-positive "
                       f"values. Received: num_bins={num_bins}.")
    if mask_value is not None:
      raise NotImplementedError("Hashing(mask_value=...) is not on the hot path")
    if salt is None:
      raise NotImplementedError(
          "unsalted Hashing (FarmHash64) is not on the hot path; UnifiedEmbedding always "
          "passes salt=[feature_no, chunk_id]")
    if isinstance(salt, (int, np.integer)):
      salt = [int(salt), int(salt)]      # Keras duplicates a scalar salt into the 128-bit key
    salt = [int(s) for s in salt]
    if len(salt) != 2:
      raise ValueError(f"`salt` must be an int or a list/tuple of 2 ints; got {salt!r}")
    self.num_bins = int(num_bins)
    self.salt = salt
    self._device = device

  def _hash_tensor(self, ids: torch.Tensor) -> torch.Tensor:
    dev = ids.device if ids.is_cuda else _device(self._device)
    if ids.dtype not in (torch.int32, torch.int64):
      if ids.dtype.is_floating_point or ids.dtype == torch.bool:
        raise ValueError(f"Hashing needs integer or string input; got {ids.dtype}")
      ids = ids.long()
    flat = ids.to(dev).reshape(-1).contiguous()
    out = torch.empty(flat.shape, dtype=torch.int64, device=dev)
    _lib.check(_lib.load().tfrs_hash_bucket_strong_ids(
        _lib.ptr(flat), 1 if flat.dtype == torch.int64 else 0, flat.numel(), self.num_bins,
        self.salt[0] & _U64, self.salt[1] & _U64, _lib.ptr(out), _lib.current_stream()))
    return out.reshape(ids.shape)

  def _hash_strings(self, arr: np.ndarray) -> torch.Tensor:
    dev = _device(self._device)
    d_blob, d_off = pack_strings(arr, dev)
    n = d_off.numel() - 1
    out = torch.empty((n,), dtype=torch.int64, device=dev)
    _lib.check(_lib.load().tfrs_hash_bucket_strong_bytes(
        _lib.ptr(d_blob), _lib.ptr(d_off), n, self.num_bins, self.salt[0] & _U64,
        self.salt[1] & _U64, _lib.ptr(out), _lib.current_stream()))
    return out.reshape(arr.shape)

  def forward(self, inputs):
    if isinstance(inputs, RaggedIds):
      return inputs.with_values(self.forward(inputs.values))
    if isinstance(inputs, SparseIds):
      return SparseIds(inputs.indices, self.forward(inputs.values), inputs.dense_shape)
    if isinstance(inputs, torch.Tensor):
      return self._hash_tensor(inputs)
    arr = np.asarray(inputs)
    if arr.dtype.kind in ("U", "S", "O"):
      return self._hash_strings(arr)
    if arr.dtype.kind in ("i", "u"):
      return self._hash_tensor(torch.as_tensor(arr.astype(np.int64)))
    raise ValueError(f"Hashing needs integer or string input; got dtype {arr.dtype}")
