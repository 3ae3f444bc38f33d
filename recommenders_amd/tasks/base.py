"""Task marker class (reference tasks/base.py:19-26)."""


class Task:
  """Marker base class for tasks."""
