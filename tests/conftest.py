"""pytest configuration: marker registration and shared helpers."""

import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _seed_global_rng(request):
  """Layers initialise their weights from torch's GLOBAL generator; the float gates are 4 x the error observed
  for THIS quantity, so the inputs of a test must not change from run to run (unseeded, the observed error of the
  low-rank Cross gradients moved by 1.5 x between two runs).  Seeded per test from its node id."""
  import zlib
  try:
    import torch
  except ImportError:      # pragma: no cover
    yield
    return
  torch.manual_seed(zlib.crc32(request.node.nodeid.encode()) & 0x7FFFFFFF)
  yield


def pytest_configure(config):
  config.addinivalue_line(
      "markers", "gpu: needs a real MI355X (run with `pytest -m gpu` via gpurun)")


def load_golden(name):
  with open(os.path.join(GOLDEN, name)) as f:
    return json.load(f)


@pytest.fixture(scope="session")
def golden():
  return load_golden


# ------------------------------------------------------------------------------------------------
# Float gates: every floating-point comparison of a HIP kernel with the
# float64 oracle measures   err = max |got - ref| / yardstick   where the yardstick is the SUM OF
# THE ABSOLUTE VALUES OF THE TERMS that make up the entry (the error model of a floating-point dot
# product; for a split-fp16 product hi*hi + hi*lo + lo*hi with f32 accumulation the expected error
# is ~2^-21 of it, for an f32 fma chain ~D * 2^-24), NOT the largest entry of the tensor.  The
# observed value of every gate is appended to gpurun_out/observed_errors.jsonl so the gates can be
# audited: each limit below is set at <= 4x the largest value observed on MI355X
# (profiles/r03_observed_errors.md).
# ------------------------------------------------------------------------------------------------
_ERRLOG = os.path.join(ROOT, "gpurun_out", "observed_errors.jsonl")
# Per-gate limits: 4 x the largest error observed on MI355X, generated from a full run
# (tools/summarize_errors.py --gates); a gate without an entry falls back to the limit the test passes.
try:
  with open(os.path.join(GOLDEN, "float_gates.json")) as _f:
    _GATES = json.load(_f)["gates"]
except (OSError, ValueError, KeyError):      # pragma: no cover
  _GATES = {}


def _gate_limit(name, fallback):
  """(hard, soft): the family ceiling the test passes is the CONTRACT; the per-gate entry of float_gates.json (4 x the error observed for this quantity in
  round 3, FROZEN: never regenerated together with a kernel change) is an early-warning tier --
  exceeding it emits a warning and is recorded, because a max-statistic with a 4 x margin moves
  with the inputs (a renamed test reseeds them), the compiler and the reduction order
  (ADVICE round 3)."""
  import re
  key = re.sub(r"\.d(biases|u_kernels|v_kernels)\.\d", ".dparam", name)
  soft = min(_GATES.get(key, fallback), fallback)
  # Round 5: a HARD per-gate limit again -- 16 x the error observed for this quantity when the
  # tier was frozen (= 4 x the frozen tier), never above the family ceiling.  Between the tier and that limit a
  # run warns; above it the test fails: a kernel regression of an order of magnitude can no longer hide under a
  # family ceiling that is several orders looser.  (Largest ratio over every recorded run so far: 3.3 x the
  # tier = 13 x the frozen observation, softmax_f16.sizes.dq.)
  hard = min(fallback, 4.0 * _GATES[key]) if key in _GATES else fallback
  return hard, soft


def _record_error(name, err, limit):
  try:
    os.makedirs(os.path.dirname(_ERRLOG), exist_ok=True)
    with open(_ERRLOG, "a") as f:
      f.write(json.dumps({"gate": name, "observed": err, "limit": limit,
                          "test": os.environ.get("PYTEST_CURRENT_TEST", "")}) + "\n")
  except OSError:
    pass


def float_gate(name, got, ref, yardstick, limit, floor=1e-30, floor_rel=0.0):
  """Asserts max |got - ref| / max(yardstick, floor) <= limit and records the observed value.
  `limit` is the family's ceiling and the limit in force; the per-gate entry of
  tests/golden/float_gates.json (4 x the error observed for THIS quantity, frozen) only warns.
  Accepts numpy arrays or torch tensors (torch: evaluated on the tensors' device in float64);
  `yardstick` broadcasts against `ref`.  `floor_rel` > 0 adds floor_rel * max(yardstick) to every
  entry's yardstick: the error model of a product whose operands share ONE power-of-two scale per
  tile (split-fp16 with a common scale: absolute error 2^-22 of the tile's largest term, so an
  entry whose own terms are all tiny is accurate relative to its neighbours' terms, not to its
  own) -- used only where the docstring of the test says why."""
  limit, soft = _gate_limit(name, limit)
  try:
    import torch
  except ImportError:      # pragma: no cover
    torch = None
  if torch is not None and isinstance(got, torch.Tensor):
    ref_t = ref if isinstance(ref, torch.Tensor) else torch.as_tensor(ref, device=got.device)
    y_t = yardstick if isinstance(yardstick, torch.Tensor) else torch.as_tensor(yardstick, device=got.device)
    y_t = y_t.double().abs()
    if floor_rel:
      y_t = y_t + floor_rel * y_t.max()
    ratio = (got.detach().double() - ref_t.double()).abs() / y_t.clamp_min(floor)
    err = float(ratio.max().item()) if ratio.numel() else 0.0
  else:
    import numpy as np
    g = np.asarray(got, dtype=np.float64)
    r = np.asarray(ref, dtype=np.float64)
    y = np.abs(np.asarray(yardstick, dtype=np.float64))
    if floor_rel:
      y = y + floor_rel * y.max()
    y = np.maximum(y, floor)
    err = float(np.max(np.abs(g - r) / y)) if g.size else 0.0
  _record_error(name, err, limit)
  if err > soft:
    import warnings
    warnings.warn(f"{name}: observed {err:.3e} above the frozen round-3 tier {soft:.3e} "
                  f"(ceiling {limit:.3e})")
  assert err <= limit, f"{name}: observed {err:.3e} > gate {limit:.3e} (relative to the sum of |terms|)"
  return err
