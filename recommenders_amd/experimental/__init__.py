"""Experimental components (mirrors tensorflow_recommenders/experimental)."""

from recommenders_amd.experimental import layers, models  # noqa: F401
