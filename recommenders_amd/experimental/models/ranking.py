"""The configurable DLRM / DCN ranking model (mirror of
``tensorflow_recommenders/experimental/models/ranking.py:27-257``): same constructor
arguments, ``compute_loss`` input formats and ``call`` data flow

    sparse ids --embedding_layer--> [B, D] per feature --+
    dense features --bottom_stack--> [B, D] -------------+--> feature_interaction
        --(concat_dense: concat with the bottom-stack output)--> top_stack --> [B]

Defaults as in the reference: ``bottom_stack = MLP([256, 64, 16], final_activation="relu")``,
``top_stack = MLP([512, 256, 1], final_activation="sigmoid")``,
``feature_interaction = DotInteraction()``, task = ``tasks.Ranking`` with per-example binary
cross-entropy, AUC / accuracy / prediction-mean / label-mean metrics (:98-133).
The embedding lookups, the Gram/cross interaction and every dense matmul are HIP kernels.
"""

from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from recommenders_amd import losses
from recommenders_amd.layers import blocks
from recommenders_amd.layers import embedding as embedding_lib
from recommenders_amd.layers.feature_interaction import dot_interaction as dot_lib
from recommenders_amd.metrics import basic as basic_metrics
from recommenders_amd.metrics.factorized_top_k import Mean
from recommenders_amd.models import base
from recommenders_amd.tasks import ranking as ranking_task


class EmbeddingDict(torch.nn.Module):
  """``{feature: ids[B]} -> {feature: embeddings[B, dim]}``: one table per feature (the role the
  reference's tests give to ``TPUEmbedding`` with one ``TableConfig`` per feature,
  ``ranking_test.py:30-59``)."""

  def __init__(self, vocab_sizes: Dict[str, int], dim: int):
    super().__init__()
    self.tables = torch.nn.ModuleDict({str(name): embedding_lib.Embedding(int(v), dim)
                                       for name, v in vocab_sizes.items()})

  def forward(self, features: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {name: self.tables[str(name)](ids) for name, ids in features.items()}


class ConcatCross(torch.nn.Module):
  """``tf.keras.Sequential([Concatenate(), Cross()])`` -- the reference's DCN recipe (:41-46)."""

  def __init__(self, cross: Optional[torch.nn.Module] = None):
    super().__init__()
    from recommenders_amd.layers.feature_interaction import dcn
    self.cross = cross if cross is not None else dcn.Cross()

  def forward(self, inputs: Sequence[torch.Tensor]) -> torch.Tensor:
    return self.cross(torch.cat(list(inputs), dim=-1))


class Ranking(base.Model):
  """A configurable ranking model (reference :27-257)."""

  def __init__(self, embedding_layer: torch.nn.Module,
               bottom_stack: Optional[torch.nn.Module] = None,
               feature_interaction: Optional[torch.nn.Module] = None,
               top_stack: Optional[torch.nn.Module] = None, concat_dense: bool = True,
               task: Optional[torch.nn.Module] = None):
    super().__init__()
    self._embedding_layer = embedding_layer
    self._concat_dense = concat_dense
    self._bottom_stack = (bottom_stack if bottom_stack is not None
                          else blocks.MLP(units=[256, 64, 16], final_activation="relu"))     # :100-104
    self._top_stack = (top_stack if top_stack is not None
                       else blocks.MLP(units=[512, 256, 1], final_activation="sigmoid"))     # :105-109
    self._feature_interaction = (feature_interaction if feature_interaction is not None
                                 else dot_lib.DotInteraction())                               # :110-114
    if task is not None:
      self._task = task
    else:                                                                                     # :118-133
      self._task = ranking_task.Ranking(
          loss=losses.BinaryCrossentropy(reduction="none"),
          metrics=[basic_metrics.AUC(name="auc"), basic_metrics.BinaryAccuracy(name="accuracy")],
          prediction_metrics=[Mean("prediction_mean")],
          label_metrics=[Mean("label_mean")])

  def compute_loss(self, inputs, training: bool = False) -> torch.Tensor:                     # :135-206
    if len(inputs) == 2:
      features, labels = inputs
      sample_weight = None
    elif len(inputs) == 3:
      features, labels, sample_weight = inputs
    else:
      raise ValueError(
          "Inputs should be either a tuple of (features, labels), "
          "or a tuple of (features, labels, sample weights). "
          f"Got a length {len(inputs)} tuple instead: {inputs}.")
    outputs = self(features)
    loss = self._task(labels, outputs, sample_weight=sample_weight)
    loss = loss.mean()
    replicas = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    return loss / replicas        # gradients are summed across replicas (:203-206)

  def forward(self, inputs: Dict[str, torch.Tensor]) -> torch.Tensor:                          # :208-236
    dense_features = inputs["dense_features"]
    sparse_features = inputs["sparse_features"]
    sparse_embeddings = self._embedding_layer(sparse_features)
    vecs: List[torch.Tensor] = []
    for _, e in sorted(sparse_embeddings.items(), key=lambda kv: str(kv[0])):   # tf.nest.flatten order
      vecs.append(e.reshape(e.shape[0], -1) if e.dim() > 2 else e)              # squeeze [B,1,D]
    dense_embedding_vec = self._bottom_stack(dense_features.to(torch.float32))
    interaction_output = self._feature_interaction(vecs + [dense_embedding_vec])
    if self._concat_dense:
      out = torch.cat([dense_embedding_vec, interaction_output], dim=1)
    else:
      out = interaction_output
    prediction = self._top_stack(out)
    return prediction.reshape(-1)

  @property
  def embedding_trainable_variables(self) -> List[torch.nn.Parameter]:                        # :238-249
    return [p for p in self._embedding_layer.parameters() if p.requires_grad]

  @property
  def dense_trainable_variables(self) -> List[torch.nn.Parameter]:                            # :251-257
    emb = {id(p) for p in self._embedding_layer.parameters()}
    return [p for p in self.parameters() if p.requires_grad and id(p) not in emb]
