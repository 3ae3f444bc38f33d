"""Oracle for the salted hashing front-end of ``UnifiedEmbedding`` (test infrastructure only).

The reference hashes every feature once per chunk with
``tf.keras.layers.Hashing(num_bins=buckets_per_table, salt=[feature_no, chunk_id])``
(``layers/feature_multiplexing/unified_embedding.py:116-119,155-159,198-205``).  With a
salt, Keras ``Hashing`` calls ``tf.strings.to_hash_bucket_strong(x, num_bins, key=salt)``
after turning integer inputs into their decimal strings (``tf.as_string``); TensorFlow's
strong hash is SipHash-2-4 (64-bit) of the string bytes under the 128-bit key
``(salt[0], salt[1])``, and the bucket is ``hash mod num_bins`` on unsigned 64-bit values.
Those semantics live in TensorFlow / tf-keras / highwayhash (not vendored under
/root/reference), so they are restated here from the published SipHash
definition (Aumasson & Bernstein, 2012).

Pinning: ``siphash24`` is checked against the SipHash paper's known-answer vectors
(key 00..0f, messages 00..len-1; ``tests/test_oracle_golden.py``).  The TensorFlow glue
around it (decimal conversion, key order, modulo) is PARITY UNPINNED: the reference's
test (``unified_embedding_test.py:73-150``) asserts output shapes only and TensorFlow
cannot be run here.
"""

from typing import Iterable, Sequence, Union

import numpy as np

_MASK = (1 << 64) - 1


def _rotl(x: int, b: int) -> int:
  return ((x << b) | (x >> (64 - b))) & _MASK


def siphash24(k0: int, k1: int, data: bytes) -> int:
  """SipHash-2-4 with key (k0, k1) (little-endian 64-bit halves) -> unsigned 64-bit."""
  v0 = (k0 ^ 0x736F6D6570736575) & _MASK
  v1 = (k1 ^ 0x646F72616E646F6D) & _MASK
  v2 = (k0 ^ 0x6C7967656E657261) & _MASK
  v3 = (k1 ^ 0x7465646279746573) & _MASK

  def sipround():
    nonlocal v0, v1, v2, v3
    v0 = (v0 + v1) & _MASK
    v1 = _rotl(v1, 13) ^ v0
    v0 = _rotl(v0, 32)
    v2 = (v2 + v3) & _MASK
    v3 = _rotl(v3, 16) ^ v2
    v0 = (v0 + v3) & _MASK
    v3 = _rotl(v3, 21) ^ v0
    v2 = (v2 + v1) & _MASK
    v1 = _rotl(v1, 17) ^ v2
    v2 = _rotl(v2, 32)

  n = len(data)
  full = n - (n % 8)
  for off in range(0, full, 8):
    m = int.from_bytes(data[off:off + 8], "little")
    v3 ^= m
    sipround()
    sipround()
    v0 ^= m
  tail = data[full:] + b"\x00" * (7 - (n % 8)) + bytes([n & 0xFF])
  m = int.from_bytes(tail, "little")
  v3 ^= m
  sipround()
  sipround()
  v0 ^= m
  v2 ^= 0xFF
  for _ in range(4):
    sipround()
  return (v0 ^ v1 ^ v2 ^ v3) & _MASK


def _as_bytes(x) -> bytes:
  """``tf.as_string`` for integers (decimal, '-' for negatives); strings as UTF-8."""
  if isinstance(x, (bytes, np.bytes_)):
    return bytes(x)
  if isinstance(x, (str, np.str_)):
    return str(x).encode("utf-8")
  return str(int(x)).encode("ascii")


def hash_bucket_strong(values: Union[np.ndarray, Iterable], num_bins: int,
                       salt: Sequence[int]) -> np.ndarray:
  """Keras ``Hashing(num_bins, salt=[s0, s1])`` on integers or strings; shape preserved,
  int64 out (``unified_embedding.py:198-205``)."""
  if num_bins is None or num_bins <= 0:
    raise ValueError("`num_bins` must be a positive integer")
  if len(salt) != 2:
    raise ValueError("`salt` must hold two integers")
  arr = np.asarray(values)
  flat = arr.reshape(-1)
  out = np.empty(flat.shape, dtype=np.int64)
  k0, k1 = int(salt[0]) & _MASK, int(salt[1]) & _MASK
  for i, v in enumerate(flat):
    out[i] = siphash24(k0, k1, _as_bytes(v)) % num_bins
  return out.reshape(arr.shape)
