"""Experimental layers (mirrors tensorflow_recommenders/experimental/layers/__init__.py)."""

from recommenders_amd.experimental.layers import embedding  # noqa: F401
