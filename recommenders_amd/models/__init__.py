"""Models (mirrors tensorflow_recommenders/models/__init__.py)."""

from recommenders_amd.models.base import Model  # noqa: F401
