"""Layers (mirrors tensorflow_recommenders/layers/__init__.py:18-23)."""

from recommenders_amd.layers import blocks, embedding, sharded_embedding, factorized_top_k, feature_interaction, loss  # noqa: F401
from recommenders_amd.layers import feature_multiplexing, hashing  # noqa: F401
