"""Unified Embedding: several features multiplexed into a few shared hashed tables.

Mirrors ``layers/feature_multiplexing/unified_embedding.py``: ``UnifiedEmbeddingConfig``
(``:68-134``) owns ``num_tables`` tables of ``buckets_per_table x dim_per_table``; every
``add_feature(name, num_chunks)`` claims ``num_chunks`` lookups, assigned to the tables
round-robin, each with its own hash salt ``[feature_no, chunk_id]``.  ``UnifiedEmbedding.call``
(``:186-215``) hashes every feature once per chunk, looks the buckets up through the
``TPUEmbedding`` layer and concatenates a feature's chunks (sorted by chunk name) along the
last axis; the outputs come back as a list in the order the features were added.

Everything runs on the GPU.  A dense feature (ids or strings of any rank) takes ONE kernel:
every lookup's SipHash bucket is computed in registers and the table row is copied straight
to its slot of the concatenated ``[..., num_chunks * dim]`` output, so neither the buckets
nor the per-chunk activations nor a concat pass touch HBM.  Ragged / sparse features go
through ``layers/hashing.py`` and the combiner kernels of ``layers/tpu_embedding_layer.py``
chunk by chunk.
"""

import ctypes
from typing import Dict, List

import numpy as np
import torch

from recommenders_amd import _lib
from recommenders_amd.layers import embedding as emb
from recommenders_amd.layers.hashing import Hashing, pack_strings
from recommenders_amd.layers.tpu_embedding_layer import (FeatureConfig, RaggedIds, SparseIds,
                                                         TableConfig, TPUEmbedding)

_U64 = (1 << 64) - 1


def _fusable_dim(d: int) -> bool:
  per_row = d // 4
  return d % 4 == 0 and 1 <= per_row <= 64 and (per_row & (per_row - 1)) == 0


class _UnifiedLookupFn(torch.autograd.Function):
  """All chunks of one dense feature in one launch: hash under each chunk's salt, gather
  from each chunk's table, write the concatenation (``tfrs_unified_embedding_fwd``).  The
  buckets are kept; the backward hands every chunk's ``(buckets, grad columns)`` to its table
  as IndexedSlices (sparse optimizers) or scatter-adds them into dense table gradients."""

  @staticmethod
  def forward(ctx, values, salts, num_bins, table_index, *tables):
    dev = tables[0].device
    if dev.type != "cuda":
      raise RuntimeError("recommenders_amd ops need tensors on the GPU; there is no CPU fallback.")
    if isinstance(values, torch.Tensor) and (values.dtype.is_floating_point
                                             or values.dtype == torch.bool):
      raise ValueError(f"UnifiedEmbedding needs integer or string features; got {values.dtype}")
    d = tables[0].shape[1]
    n_chunks = len(salts)
    if isinstance(values, torch.Tensor):
      ids = values.to(dev)
      if ids.dtype not in (torch.int32, torch.int64):
        ids = ids.long()
      flat = ids.reshape(-1).contiguous()
      shape, n = tuple(ids.shape), flat.numel()
      blob = offsets = None
    else:
      blob, offsets = pack_strings(values, dev)
      shape, n = tuple(values.shape), offsets.numel() - 1
      flat = None
    out = torch.empty((n, n_chunks * d), dtype=torch.float32, device=dev)
    buckets = torch.empty((n, n_chunks), dtype=torch.int64, device=dev)
    per_chunk = [tables[table_index[c]] for c in range(n_chunks)]
    ptrs = (ctypes.c_void_p * n_chunks)(*[t.data_ptr() for t in per_chunk])
    s0 = (ctypes.c_uint64 * n_chunks)(*[int(s[0]) & _U64 for s in salts])
    s1 = (ctypes.c_uint64 * n_chunks)(*[int(s[1]) & _U64 for s in salts])
    _lib.check(_lib.load().tfrs_unified_embedding_fwd(
        _lib.ptr(flat), 1 if (flat is not None and flat.dtype == torch.int64) else 0,
        _lib.ptr(blob), _lib.ptr(offsets), n, n_chunks, ptrs, s0, s1, int(num_bins), d,
        _lib.ptr(out), _lib.ptr(buckets), _lib.current_stream()))
    ctx.save_for_backward(buckets)
    ctx.table_index = table_index
    ctx.tables = tables
    ctx.d = d
    return out.reshape(shape + (n_chunks * d,))

  @staticmethod
  def backward(ctx, grad_out):
    (buckets,) = ctx.saved_tensors
    n_chunks, d = buckets.shape[1], ctx.d
    g = grad_out.reshape(-1, n_chunks, d)
    grads = [None] * len(ctx.tables)
    for c in range(n_chunks):
      t = ctx.table_index[c]
      table = ctx.tables[t]
      ids_c, rows_c = buckets[:, c].contiguous(), g[:, c, :].contiguous()
      if getattr(table, "_tfrs_sparse_grad", False):
        table._tfrs_slices.append((ids_c, rows_c))
        continue
      part = emb.scatter_add_rows(rows_c, ids_c, table.shape[0])
      grads[t] = part if grads[t] is None else grads[t] + part
    return (None, None, None, None) + tuple(grads)


class _UnifiedMultiFn(torch.autograd.Function):
  """EVERY dense integer feature of the layer in one launch (``tfrs_unified_embedding_fwd_multi``): a
  unit is one (feature, chunk); the kernel hashes the unit's feature ids under the unit's salt and copies
  the table row into the unit's column block of its feature's own ``[n, chunks * d]`` output.  (One launch
  per feature -- ``_UnifiedLookupFn`` -- was launch-bound at the DCN-v2 shapes: 26 x 22 us.)  ``meta`` =
  (per feature: (chunk salts, table index of each chunk)), num_bins; the tensors that follow are the
  features' id tensors, then the distinct tables."""

  @staticmethod
  def forward(ctx, meta, num_bins, n_features, *tensors):
    id_tensors, tables = tensors[:n_features], tensors[n_features:]
    dev = tables[0].device
    d = tables[0].shape[1]
    flats, shapes = [], []
    for ids in id_tensors:
      ids = ids.to(dev)
      shapes.append(tuple(ids.shape))
      flats.append(ids.reshape(-1).contiguous())
    n = flats[0].numel()
    i64 = flats[0].dtype == torch.int64
    outs = [torch.empty((n, len(salts) * d), dtype=torch.float32, device=dev) for salts, _ in meta]
    unit_ids, unit_tables, unit_outs, s0, s1, chunks, cidx = [], [], [], [], [], [], []
    for f, (salts, index) in enumerate(meta):
      for c, salt in enumerate(salts):
        unit_ids.append(flats[f].data_ptr())
        unit_tables.append(tables[index[c]].data_ptr())
        unit_outs.append(outs[f].data_ptr())
        s0.append(int(salt[0]) & _U64)
        s1.append(int(salt[1]) & _U64)
        chunks.append(len(salts))
        cidx.append(c)
    nu = len(unit_ids)
    buckets = torch.empty((n, nu), dtype=torch.int64, device=dev)
    vp, u64, i32 = ctypes.c_void_p * nu, ctypes.c_uint64 * nu, ctypes.c_int32 * nu
    _lib.check(_lib.load().tfrs_unified_embedding_fwd_multi(
        nu, vp(*unit_ids), 1 if i64 else 0, n, vp(*unit_tables), u64(*s0), u64(*s1), int(num_bins), d,
        vp(*unit_outs), i32(*chunks), i32(*cidx), _lib.ptr(buckets), _lib.current_stream()))
    ctx.save_for_backward(buckets)
    ctx.meta, ctx.tables, ctx.d = meta, tables, d
    ctx.n_features = n_features
    return tuple(o.reshape(shape + (o.shape[1],)) for o, shape in zip(outs, shapes))

  @staticmethod
  def backward(ctx, *grads):
    (buckets,) = ctx.saved_tensors
    d = ctx.d
    table_grads = [None] * len(ctx.tables)
    u = 0
    for f, (salts, index) in enumerate(ctx.meta):
      g = grads[f]
      nc = len(salts)
      g = None if g is None else g.reshape(-1, nc, d)
      for c in range(nc):
        if g is not None:
          table = ctx.tables[index[c]]
          ids_c, rows_c = buckets[:, u].contiguous(), g[:, c, :].contiguous()
          if getattr(table, "_tfrs_sparse_grad", False):
            table._tfrs_slices.append((ids_c, rows_c))
          else:
            part = emb.scatter_add_rows(rows_c, ids_c, table.shape[0])
            t = index[c]
            table_grads[t] = part if table_grads[t] is None else table_grads[t] + part
        u += 1
    return (None, None, None) + (None,) * ctx.n_features + tuple(table_grads)


class UnifiedEmbeddingConfig:
  """Describes the shared tables and the features multiplexed into them."""

  def __init__(self, buckets_per_table: int, dim_per_table: int, num_tables: int, name: str,
               **kwargs):
    self._buckets_per_table = buckets_per_table
    self._dim_per_table = dim_per_table
    self._num_tables = num_tables
    self._name = name
    self._next_table = 0
    self._num_features = 0
    self._table_configs = [
        TableConfig(vocabulary_size=buckets_per_table, dim=dim_per_table,
                    name=f"{name}_{i}", **kwargs)
        for i in range(num_tables)
    ]
    self._embed_configs: Dict[str, Dict[str, FeatureConfig]] = {}
    self._hashing_configs: Dict[str, Dict[str, dict]] = {}

  def add_feature(self, name: str, num_chunks: int, **kwargs) -> None:
    """Claims ``num_chunks`` table lookups for ``name``; the feature's embedding has
    ``num_chunks * dim_per_table`` dimensions.  ``kwargs`` go to each ``FeatureConfig``."""
    embed, hashing = {}, {}
    for chunk_id in range(num_chunks):
      chunk_name = f"{self._name}_{name}_lookup_{chunk_id}"
      embed[chunk_name] = FeatureConfig(table=self._table_configs[self._next_table],
                                        name=chunk_name, **kwargs)
      hashing[chunk_name] = {"num_bins": self._buckets_per_table,
                             "salt": [self._num_features, chunk_id]}
      self._next_table = (self._next_table + 1) % self._num_tables
    self._num_features += 1
    self._embed_configs[name] = embed
    self._hashing_configs[name] = hashing

  @property
  def embedding_config(self):
    return self._embed_configs

  @property
  def hashing_config(self):
    return self._hashing_configs


class UnifiedEmbedding(torch.nn.Module):
  """Hash -> shared-table lookup -> per-feature concatenation."""

  def __init__(self, config: UnifiedEmbeddingConfig, optimizer=None, device=None,
               fuse: bool = True):
    super().__init__()
    self.fuse = fuse          # False: one Hashing + lookup per chunk, then concat (same values)
    if not config.embedding_config:
      raise ValueError("UnifiedEmbeddingConfig has no features; call add_feature first")
    self._embedding_layer = TPUEmbedding(feature_config=config.embedding_config,
                                         optimizer=optimizer, device=device)
    self._embed_config = config.embedding_config
    self._hash_config = config.hashing_config
    self._hashing_layers = {
        name: {chunk: Hashing(device=device, **params) for chunk, params in chunks.items()}
        for name, chunks in self._hash_config.items()
    }

  @property
  def embedding_layer(self) -> TPUEmbedding:
    return self._embedding_layer

  def _multi_feature_groups(self, features) -> Dict[tuple, List[str]]:
    """Features that one ``tfrs_unified_embedding_fwd_multi`` launch can take together: dense integer
    tensors on the layer's device, no sequence features, a fusable table dim, grouped by (number of ids,
    id width, buckets per table, dim); groups of one stay on the per-feature kernel."""
    groups: Dict[tuple, List[str]] = {}
    dev = next(iter(self._embedding_layer.embedding_tables.values())).device
    for name, layers in self._hashing_layers.items():
      value = features[name]
      if not (isinstance(value, torch.Tensor) and value.dtype in (torch.int32, torch.int64)
              and value.device == dev and value.numel() > 0):
        continue
      chunks = sorted(layers)
      configs = [self._embed_config[name][c] for c in chunks]
      if not (_fusable_dim(configs[0].table.dim) and all(f.max_sequence_length == 0 for f in configs)):
        continue
      key = (value.numel(), value.dtype, layers[chunks[0]].num_bins, configs[0].table.dim)
      groups.setdefault(key, []).append(name)
    return {k: v for k, v in groups.items() if len(v) > 1}

  def forward(self, features: Dict[str, object]) -> List[torch.Tensor]:
    """``features``: {feature name: ids/strings} holding at least every configured feature
    (extra keys are ignored).  Returns one ``[..., num_chunks * dim_per_table]`` tensor per
    feature, in ``add_feature`` order."""
    for name in self._hashing_layers:
      if name not in features:
        raise KeyError(f"feature {name!r} is missing from the inputs")
    tables = self._embedding_layer.embedding_tables
    outputs: Dict[str, torch.Tensor] = {}
    hashed, serving = {}, {}
    # dense integer features with the same number of ids and the same id width go out in ONE launch
    multi = self._multi_feature_groups(features) if self.fuse else {}
    for names in multi.values():
      distinct, meta = [], []
      for name in names:
        layers = self._hashing_layers[name]
        chunks = sorted(layers)
        configs = [self._embed_config[name][c] for c in chunks]
        for f in configs:
          if f.table not in distinct:
            distinct.append(f.table)
        meta.append((tuple(tuple(layers[c].salt) for c in chunks),
                     tuple(distinct.index(f.table) for f in configs)))
      num_bins = self._hashing_layers[names[0]][sorted(self._hashing_layers[names[0]])[0]].num_bins
      outs = _UnifiedMultiFn.apply(tuple(meta), num_bins, len(names), *[features[n] for n in names],
                                   *[tables[t] for t in distinct])
      outputs.update(zip(names, outs))
    for name, layers in self._hashing_layers.items():
      if name in outputs:
        continue
      value = features[name]
      chunks = sorted(layers)                   # concatenation order of the reference (:209-211)
      configs = [self._embed_config[name][c] for c in chunks]
      dense = not isinstance(value, (RaggedIds, SparseIds))
      if dense and not isinstance(value, torch.Tensor):
        value = np.asarray(value)
        if value.dtype.kind in ("i", "u"):
          value = torch.as_tensor(value.astype(np.int64))
      if dense and self.fuse and _fusable_dim(configs[0].table.dim) and all(
          f.max_sequence_length == 0 for f in configs):
        distinct = []
        for f in configs:
          if f.table not in distinct:
            distinct.append(f.table)
        index = tuple(distinct.index(f.table) for f in configs)
        salts = tuple(tuple(layers[c].salt) for c in chunks)
        outputs[name] = _UnifiedLookupFn.apply(value, salts, layers[chunks[0]].num_bins, index,
                                               *[tables[t] for t in distinct])
      else:                                     # ragged / sparse / odd dims: one op per chunk
        hashed[name] = {c: layers[c](value) for c in chunks}
        serving[name] = self._embed_config[name]
    if hashed:
      embedded = self._embedding_layer(hashed, serving_config=serving)
      for name, parts in embedded.items():
        outputs[name] = torch.cat([parts[k] for k in sorted(parts)], dim=-1)
    return [outputs[name] for name in self._hashing_layers]
