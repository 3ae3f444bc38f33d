"""Embedding layer of the Ranking model that splits tables by size.

Mirrors ``experimental/layers/embedding/partial_tpu_embedding.py:26-142``: features whose
table has more than ``size_threshold`` rows go to one ``TPUEmbedding`` layer (the large,
shardable tables), the rest to plain ``Embedding`` layers, one per distinct ``TableConfig``;
``call`` takes and returns ``{feature name: tensor}`` dictionaries.  On MI355X both kinds
run the same HBM gather kernels; the split is kept because it decides which tables a
deployment row-shards (``layers/sharded_embedding.py``) and which it replicates.
"""

from typing import Dict, Optional

import numpy as np
import torch

from recommenders_amd.layers import embedding as embedding_lib
from recommenders_amd.layers.tpu_embedding_layer import FeatureConfig, TPUEmbedding


class PartialTPUEmbedding(torch.nn.Module):

  def __init__(self, feature_config: Dict[str, FeatureConfig], optimizer=None,
               pipeline_execution_with_tensor_core: bool = False,
               batch_size: Optional[int] = None, size_threshold: Optional[int] = 10_000,
               device: Optional[torch.device] = None):
    super().__init__()
    large = {}
    per_table = {}
    layers = {}
    for name, feature in feature_config.items():
      table = feature.table
      if size_threshold is not None and table.vocabulary_size > size_threshold:
        large[name] = feature
        continue
      if table not in per_table:       # several features may share one table
        layer = embedding_lib.Embedding(table.vocabulary_size, table.dim, device=device)
        if table.initializer is not None:
          with torch.no_grad():
            init = np.asarray(table.initializer((table.vocabulary_size, table.dim)), np.float32)
            layer.embeddings.copy_(torch.from_numpy(init).reshape(layer.embeddings.shape))
        per_table[table] = layer
      layers[name] = per_table[table]
    self._keras_embedding_layers = layers
    self._small = torch.nn.ModuleList(list(per_table.values()))
    self._tpu_embedding = None
    if large:
      self._tpu_embedding = TPUEmbedding(large, optimizer, pipeline_execution_with_tensor_core,
                                         batch_size, device=device)

  def forward(self, inputs: Dict[str, object]) -> Dict[str, torch.Tensor]:
    output = {}
    large_inputs = {}
    for key, val in inputs.items():
      if key not in self._keras_embedding_layers:
        large_inputs[key] = val
        continue
      if not isinstance(val, (torch.Tensor, int, np.integer, np.ndarray)):
        raise ValueError("Only dense tensor input is supported for plain embedding layers, "
                         f"but got: {type(val)}")
      output[key] = self._keras_embedding_layers[key](val)
    if self._tpu_embedding is not None:
      output.update(self._tpu_embedding(large_inputs))
    return output

  @property
  def tpu_embedding(self) -> Optional[TPUEmbedding]:
    """The ``TPUEmbedding`` holding the large tables, or ``None``."""
    return self._tpu_embedding

  @property
  def keras_embedding_layers(self) -> Dict[str, embedding_lib.Embedding]:
    """Feature name -> plain ``Embedding`` layer (small tables)."""
    return self._keras_embedding_layers
