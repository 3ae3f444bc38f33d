"""Metrics (mirrors tensorflow_recommenders/metrics/__init__.py:17-18)."""

from recommenders_amd.metrics.factorized_top_k import Factorized, FactorizedTopK, Mean  # noqa: F401
from recommenders_amd.metrics.basic import AUC, BinaryAccuracy, RootMeanSquaredError  # noqa: F401
