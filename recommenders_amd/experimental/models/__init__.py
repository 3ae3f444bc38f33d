"""Experimental models (mirrors tensorflow_recommenders/experimental/models/__init__.py)."""

from recommenders_amd.experimental.models.ranking import Ranking  # noqa: F401
