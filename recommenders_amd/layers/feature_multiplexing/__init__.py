"""Feature multiplexing (mirrors tensorflow_recommenders/layers/feature_multiplexing/__init__.py:17-18)."""

from recommenders_amd.layers.feature_multiplexing.unified_embedding import UnifiedEmbedding  # noqa: F401
from recommenders_amd.layers.feature_multiplexing.unified_embedding import UnifiedEmbeddingConfig  # noqa: F401
