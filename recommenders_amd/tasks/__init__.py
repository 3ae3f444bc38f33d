"""Tasks (mirrors tensorflow_recommenders/tasks/__init__.py:17-19)."""

from recommenders_amd.tasks.base import Task  # noqa: F401
from recommenders_amd.tasks.retrieval import Retrieval  # noqa: F401
from recommenders_amd.tasks.ranking import Ranking  # noqa: F401
