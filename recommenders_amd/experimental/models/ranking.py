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


class _TableView:
  """One feature's rows of the fused table (``EmbeddingDict.tables[name].embeddings``)."""

  def __init__(self, embeddings: torch.Tensor):
    self.embeddings = embeddings


class _GatherWithTailFn(torch.autograd.Function):
  """``x = table[rows]`` for ``rows[B, F + 1]`` whose last column is the invalid id -1 (reads zeros,
  takes no gradient), then ``x[:, -1, :] = tail``: the ``[B, F + 1, D]`` block DotInteraction reads,
  from one gather launch.  Backward: the table receives the ``(rows, dx)`` slice of the lookup
  (``layers.embedding._emit_table_grad``), ``tail`` the last feature's gradient."""

  @staticmethod
  def forward(ctx, table, rows, tail):
    ctx.save_for_backward(rows)
    ctx.vocab = table.shape[0]
    ctx.table_ref = table
    x = embedding_lib.gather_rows(table, rows).contiguous()
    x[:, -1, :] = tail
    return x

  @staticmethod
  def backward(ctx, dx):
    dx = dx.contiguous()
    return embedding_lib._emit_table_grad(ctx, dx), None, dx[:, -1, :].contiguous()


class EmbeddingDict(torch.nn.Module):
  """``{feature: ids[B]} -> {feature: embeddings[B, dim]}``: one table per feature (the role the
  reference's tests give to ``TPUEmbedding`` with one ``TableConfig`` per feature,
  ``ranking_test.py:30-59``).

  The tables are the row ranges of ONE parameter ``embeddings[sum(vocab), dim]`` (as on the TPU,
  where the embedding layer owns one sharded store): the lookups of a step are one gather launch
  and their gradients one ``(ids, rows)`` slice, i.e. one sort + one fused Adagrad update instead
  of one chain of ~15 small launches per feature -- 100 features at batch 131072 spent 12 ms of a
  49 ms train step in those chains.  ``tables[name].embeddings`` is a view of a feature's rows.
  Ids are not range-checked (like ``layers.embedding.Embedding``): an id >= its table's size
  reads the next table's rows."""

  def __init__(self, vocab_sizes: Dict[str, int], dim: int, device: Optional[torch.device] = None):
    super().__init__()
    self._names = [str(name) for name in vocab_sizes]
    self._sizes = [int(v) for v in vocab_sizes.values()]
    if any(v < 1 for v in self._sizes):
      raise ValueError("EmbeddingDict: every vocabulary needs at least one row.")
    starts, total = [], 0
    for v in self._sizes:
      starts.append(total)
      total += v
    if total > 0xFFFFFFFF:
      raise ValueError("EmbeddingDict: more than 2^32 rows in total.")
    self._starts = dict(zip(self._names, starts))
    dev = device if device is not None else (
        torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu"))
    w = torch.empty((total, int(dim)), dtype=torch.float32, device=dev)
    w.uniform_(-0.05, 0.05)  # Keras "uniform" initialiser
    self.embeddings = torch.nn.Parameter(w)
    self.embeddings._tfrs_embedding = True   # lets optimizers.Adagrad ask for sliced gradients
    self.register_buffer("_start_rows", torch.tensor(starts, dtype=torch.int64, device=dev),
                         persistent=False)

  @property
  def tables(self) -> Dict[str, _TableView]:
    return {name: _TableView(self.embeddings[lo:lo + n])
            for name, lo, n in zip(self._names, self._starts.values(), self._sizes)}

  def can_stack(self, features: Dict[str, torch.Tensor]) -> bool:
    """True when ``stacked`` applies: every feature is a vector of ids ``[B]`` of one length."""
    shapes = {tuple(t.shape) for t in features.values() if isinstance(t, torch.Tensor)}
    return (len(features) > 0 and len(shapes) == 1 and len(next(iter(shapes))) == 1
            and all(isinstance(t, torch.Tensor) and str(k) in self._starts for k, t in features.items()))

  def stacked(self, features: Dict[str, torch.Tensor], tail: torch.Tensor) -> torch.Tensor:
    """``[B, F + 1, dim]``: the features' embeddings in ``str(name)`` order (the order
    ``Ranking.call`` flattens the embedding dict in) followed by ``tail[B, dim]`` -- the layout
    DotInteraction consumes, produced by ONE gather with the ids laid out ``[B, F + 1]`` (last
    column: the invalid id -1, which reads zeros and takes no gradient) instead of F lookups,
    a concat of F + 1 tensors and its backward."""
    keys = sorted(features, key=str)
    dev = self.embeddings.device
    starts = torch.tensor([self._starts[str(k)] for k in keys], dtype=torch.int64, device=dev)
    rows = torch.stack([features[k].to(dev).long() for k in keys], dim=1) + starts        # [B, F]
    rows = torch.cat([rows, rows.new_full((rows.shape[0], 1), -1)], dim=1)                  # [B, F + 1]
    return _GatherWithTailFn.apply(self.embeddings, rows, tail.to(torch.float32))             # [B, F + 1, dim]

  def forward(self, features: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    keys = list(features)
    ids = []
    for key in keys:
      if str(key) not in self._starts:
        raise KeyError(f"EmbeddingDict: no table for feature {key!r}.")
      t = features[key]
      t = t if isinstance(t, torch.Tensor) else torch.as_tensor(t)
      if t.dtype not in (torch.int32, torch.int64):
        t = t.long()
      ids.append(t.to(self.embeddings.device))
    if not keys:
      return {}
    if len(keys) > 1 and all(i.shape == ids[0].shape for i in ids):
      if [str(k) for k in keys] == self._names:
        starts = self._start_rows
      else:
        starts = torch.tensor([self._starts[str(k)] for k in keys], dtype=torch.int64,
                              device=self.embeddings.device)
      rows = torch.stack([i.long() for i in ids]) + starts.view((-1,) + (1,) * ids[0].dim())
      out = embedding_lib._GatherFn.apply(self.embeddings, rows)        # [F, ..., dim]: one launch
      return dict(zip(keys, out.unbind(0)))
    return {key: embedding_lib._GatherFn.apply(self.embeddings, i.long() + self._starts[str(key)])
            for key, i in zip(keys, ids)}


class ConcatCross(torch.nn.Module):
  """``tf.keras.Sequential([Concatenate(), Cross()])`` -- the reference's DCN recipe (:41-46).
  ``num_layers`` > 1 stacks that many ``Cross`` layers on the same ``x0`` (``x_{l+1} = cross_l(x0, x_l)``,
  ``dcn.py:47-56``: "x0 = input; x1 = Cross()(x0, x0); x2 = Cross()(x0, x1)"): the "3 Cross layers" of
  BASELINE.json configs[3]."""

  def __init__(self, cross: Optional[torch.nn.Module] = None, num_layers: int = 1):
    super().__init__()
    from recommenders_amd.layers.feature_interaction import dcn
    if num_layers < 1:
      raise ValueError("ConcatCross: num_layers must be >= 1")
    if cross is not None and num_layers != 1:
      raise ValueError("ConcatCross: give either one `cross` layer or `num_layers`")
    self.layers = torch.nn.ModuleList([cross if cross is not None else dcn.Cross()] +
                                      [dcn.Cross() for _ in range(num_layers - 1)])

  @property
  def cross(self) -> torch.nn.Module:
    return self.layers[0]

  def _stack(self, x0: torch.Tensor) -> torch.Tensor:
    from recommenders_amd.layers.feature_interaction import dcn
    # (training, plain full-rank layers on the split-fp16 path: one autograd node that accumulates x0's gradient in place)
    fused = dcn.cross_stack(x0.to(torch.float32) if x0.dtype != torch.float32 else x0, self.layers)
    if fused is not None:
      return fused
    x = x0
    for layer in self.layers:
      x = layer(x0, x)
    return x

  def forward(self, inputs: Sequence[torch.Tensor]) -> torch.Tensor:
    return self._stack(torch.cat(list(inputs), dim=-1))

  def forward_stacked(self, x: torch.Tensor, prefix: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The same on an already concatenated block ``x[B, F, D]`` (``Ranking.call`` fast path)."""
    out = self._stack(x.reshape(x.shape[0], -1))
    return out if prefix is None else torch.cat([prefix, out], dim=1)


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
    dense_embedding_vec = self._bottom_stack(dense_features.to(torch.float32))
    fi = self._feature_interaction
    if (isinstance(self._embedding_layer, EmbeddingDict) and hasattr(fi, "forward_stacked")
        and self._embedding_layer.can_stack(sparse_features)
        and dense_embedding_vec.shape[1] == self._embedding_layer.embeddings.shape[1]):
      # fast path: one gather straight into the [B, F + 1, D] block the interaction reads (the
      # concatenation of the feature vectors, `forward_stacked`), the bottom-stack output as its last
      # feature; same values as the generic path below
      x = self._embedding_layer.stacked(sparse_features, dense_embedding_vec)
      out = fi.forward_stacked(x, dense_embedding_vec if self._concat_dense else None)
      return self._top_stack(out).reshape(-1)
    sparse_embeddings = self._embedding_layer(sparse_features)
    vecs: List[torch.Tensor] = []
    for _, e in sorted(sparse_embeddings.items(), key=lambda kv: str(kv[0])):   # tf.nest.flatten order
      vecs.append(e.reshape(e.shape[0], -1) if e.dim() > 2 else e)              # squeeze [B,1,D]
    if self._concat_dense and isinstance(fi, dot_lib.DotInteraction):
      # the pairs are written next to the bottom-stack output: no concat / slice copies of the
      # [B, F (F - 1) / 2] block in either direction
      out = fi.forward_concat(vecs + [dense_embedding_vec], dense_embedding_vec)
    else:
      interaction_output = fi(vecs + [dense_embedding_vec])
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
