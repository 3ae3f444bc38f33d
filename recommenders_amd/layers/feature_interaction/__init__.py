"""Feature interaction layers (mirrors layers/feature_interaction/__init__.py:17-19)."""

from recommenders_amd.layers.feature_interaction.dcn import Cross  # noqa: F401
from recommenders_amd.layers.feature_interaction.dot_interaction import DotInteraction  # noqa: F401
from recommenders_amd.layers.feature_interaction.multi_layer_dcn import MultiLayerDCN  # noqa: F401
