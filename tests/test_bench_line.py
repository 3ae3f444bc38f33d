"""bench.py's last stdout line must stay a few KB (round 5's 23.5 KB line was not parsed by the driver:
BENCH_r05.json `parsed: null`).  bench_compact.py needs no GPU: worst-case result objects are built here."""

import copy
import json
import os

import bench_compact

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _round5_detail():
  with open(os.path.join(ROOT, "profiles", "r05_bench.json")) as f:
    return json.load(f)


def _check_contract(line: str, n_gpus: int):
  assert "\n" not in line
  obj = json.loads(line)
  for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "legs"):
    assert key in obj, key
  assert obj["n_gpus"] == n_gpus
  assert "workload" in obj["config"] and "model" not in obj["config"]
  for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
    assert key in obj["roofline"], key
  for leg in obj["legs"].values():
    assert set(leg) <= {"ms", "frac"}
  return obj


def test_compact_line_of_the_round5_result_is_small_and_complete():
  detail = _round5_detail()
  assert len(json.dumps(detail)) > 20000          # the object that broke the driver's parser
  line = bench_compact.dumps(detail)
  assert len(line) < bench_compact.LIMIT_BYTES
  obj = _check_contract(line, 1)
  for key in ("value", "unit", "cores", "kind", "sample"):
    assert key in obj["cpu_baseline"], key
  assert obj["secondary"]["unit"] == "steps/s"
  for leg in ("gather", "streaming.batch_1", "cross.forward", "cross.training_pair", "dot_interaction.forward",
              "dot_interaction.backward", "segment_sum", "sparse_adagrad", "dcn_v2_step", "dlrm_shard_step",
              "scale_workload"):
    assert leg in obj["legs"], leg
  assert abs(obj["value"] - detail["value"]) <= 1e-4 * detail["value"]
  assert abs(obj["roofline"]["frac"] - detail["roofline"]["frac"]) <= 1e-4


def test_compact_line_worst_case_stays_under_six_kilobytes():
  detail = _round5_detail()
  worst = copy.deepcopy(detail)
  # every text blown up, every known section doubled, the N > 1 keys present, new roofline keys present
  long_text = "x" * 5000
  worst["config"]["workload"] = long_text
  worst["config"]["parallelism"] = long_text
  worst["roofline"]["kernel"] = long_text
  worst["roofline"].update({"measured_ceiling": 1650.123456789, "frac_of_measured_ceiling": 0.7012345678,
                            "shader_mhz": 1630.123456, "measured_copy_gbs": 6291.123456,
                            "measured_ceiling_note": long_text, "peak_note": long_text})
  worst["cpu_baseline"]["sample"] = long_text
  worst["n_gpus"], worst["rccl_ranks"] = 8, 8
  worst["single_gpu_same_workload"] = {"value": 97446.6, "unit": "queries/s", "ms_per_step": 84.06, "steps": 3}
  worst["speedup_vs_single_gpu_same_workload"] = 7.123456789
  for i in range(40):     # far more legs than bench.py has
    worst["config_legs"]["an_extra_leg_with_a_long_descriptive_name_%02d" % i] = {
        "ms_per_step": 1.2345678 + i, "roofline": {"frac": 0.123456789, "kernel": long_text}, "note": long_text}
  line = bench_compact.dumps(worst)
  assert len(line) < 6000
  obj = _check_contract(line, 8)
  assert obj["speedup_vs_single_gpu_same_workload"] == 7.1235
  assert obj.get("legs_truncated") is True
  # the contract sections survive whatever is dropped
  assert obj["roofline"]["measured_ceiling"] == 1650.1 and obj["roofline"]["shader_mhz"] == 1630.1


def test_compact_handles_a_minimal_result_and_non_finite_numbers():
  line = bench_compact.dumps({"metric": "m", "value": float("nan"), "unit": "u", "n_gpus": 1, "steps": 1, "warmup": 0,
                              "ms_per_step": 1.0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "dtype": "f32", "data": "synthetic", "config": {"workload": "w"},
                              "roofline": {"bound": "hbm", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s",
                                           "frac": 1.25e-4, "traffic": None}})
  obj = json.loads(line)        # strict JSON: NaN became null
  assert obj["value"] is None and obj["legs"] == {}
