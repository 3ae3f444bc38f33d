"""Mirrors tensorflow_recommenders/experimental/layers/embedding/__init__.py:17."""

from recommenders_amd.experimental.layers.embedding.partial_tpu_embedding import PartialTPUEmbedding  # noqa: F401
