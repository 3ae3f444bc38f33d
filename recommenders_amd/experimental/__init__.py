"""Experimental components (mirrors tensorflow_recommenders/experimental)."""

from recommenders_amd.experimental import models  # noqa: F401
