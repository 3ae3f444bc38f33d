"""A few cases of each tools/fuzz_*.py fuzzer per test run (fixed seeds; the tools run more):
random shapes and awkward data through the default paths against the all-f32 scan / float64."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")


def _load(name):
  spec = importlib.util.spec_from_file_location(name, os.path.join(TOOLS, name + ".py"))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


@pytest.mark.parametrize("name,seed,cases", [("fuzz_topk", 11, 5), ("fuzz_streaming", 12, 3),
                                             ("fuzz_gemm", 13, 5), ("fuzz_embedding", 14, 5)])
def test_fuzzers_find_nothing(name, seed, cases, monkeypatch):
  for key in ("TFRS_TOPK_FILTER", "TFRS_TOPK_STAT", "TFRS_GEMM_MODE"):
    monkeypatch.delenv(key, raising=False)
  mod = _load(name)
  assert (mod.main(seed, cases, light=True) if name == "fuzz_embedding" else mod.main(seed, cases)) == 0
