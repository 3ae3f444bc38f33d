"""pytest configuration: marker registration and shared helpers."""

import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
  config.addinivalue_line(
      "markers", "gpu: needs a real MI355X (run with `pytest -m gpu` via gpurun)")


def load_golden(name):
  with open(os.path.join(GOLDEN, name)) as f:
    return json.load(f)


@pytest.fixture(scope="session")
def golden():
  return load_golden
