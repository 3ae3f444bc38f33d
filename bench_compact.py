"""The ONE line bench.py prints last: a compact view of the full result object.

Round 5's line had grown to 23.5 KB and the driver could not parse it (`BENCH_r05.json: parsed = null`); the
full object now goes to `gpurun_out/bench_detail.json` (and to an earlier, prefixed stdout line), and the LAST stdout
line is `compact(result)`: the contract keys (metric, value, unit, n_gpus, steps, warmup, ms_per_step,
higher_is_better, scaling, vs_baseline, dtype, data, config), `roofline`, `cpu_baseline`, `secondary` (the
"train steps/sec" half of BASELINE.json's metric) and a `legs` map name -> {ms, frac} of every other measurement
of the run.  No torch import: tests/test_host_cpu.py builds worst-case results on the CPU and checks the size.
"""

import json

LIMIT_BYTES = 4000

_TOP = ("metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step", "step_ms_median",
        "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "redo_queries_last_step",
        "speedup_vs_single_gpu_same_workload")
_CONFIG = ("workload", "corpus_rows_total", "corpus_rows_per_gpu", "rows", "dim", "batch", "k", "block_rows",
           "parallelism", "tables", "rows_per_table", "rows_per_gpu")
_ROOF = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_ms",
         "measured_ceiling", "frac_of_measured_ceiling", "shader_mhz", "measured_copy_gbs")
_CPU = ("value", "unit", "cores", "kind", "sample")
_MS_KEYS = ("ms", "ms_per_step", "ms_per_call", "ms_median", "predicted_ms_per_step")
_SKIP = ("config", "roofline", "cpu_baseline", "redo_reasons", "parity_max_rel_err_vs_float64")


def _num(x):
  """Floats to 5 significant digits (the full precision is in bench_detail.json)."""
  if isinstance(x, bool) or not isinstance(x, float):
    return x
  if x != x or x in (float("inf"), float("-inf")):
    return None
  return float("%.5g" % x)


def _text(s, n):
  return s if not isinstance(s, str) or len(s) <= n else s[: n - 3] + "..."


def _pick(d, keys, text_len):
  return {k: _text(_num(d[k]), text_len) for k in keys if k in d}


def _leg_of(node):
  """{ms, frac} of one measurement node (a dict with a time and / or a roofline), or None."""
  roof = node.get("roofline") if isinstance(node.get("roofline"), dict) else {}
  ms = next((node[k] for k in _MS_KEYS if isinstance(node.get(k), (int, float))), None)
  if ms is None:
    ms = next((roof[k] for k in ("ms_median", "avg_launch_ms") if isinstance(roof.get(k), (int, float))), None)
  frac = roof.get("frac")
  if frac is None and isinstance(node.get("filter_pass_tflops"), (int, float)):
    frac = node["filter_pass_tflops"] / 2500.0
  if ms is None and frac is None:
    return None
  leg = {}
  if ms is not None:
    leg["ms"] = _num(float(ms))
  if frac is not None:
    leg["frac"] = _num(float(frac))
  return leg


def _walk(node, path, legs):
  for key, val in node.items():
    if key in _SKIP or not isinstance(val, dict):
      continue
    name = key if not path else path + "." + key
    leg = _leg_of(val)
    if leg is not None:
      legs[name] = leg
    _walk(val, name, legs)


def compact(result: dict, limit: int = LIMIT_BYTES) -> dict:
  out = _pick(result, _TOP, 80)
  if isinstance(result.get("config"), dict):
    out["config"] = _pick(result["config"], _CONFIG, 120)
  if isinstance(result.get("roofline"), dict):
    out["roofline"] = _pick(result["roofline"], _ROOF, 100)
  if isinstance(result.get("cpu_baseline"), dict):
    out["cpu_baseline"] = _pick(result["cpu_baseline"], _CPU, 140)
  sec = result.get("secondary")
  if isinstance(sec, dict):
    out["secondary"] = _pick(sec, ("metric", "value", "unit", "ms_per_step"), 60)
    if isinstance(sec.get("roofline"), dict) and "frac" in sec["roofline"]:
      out["secondary"]["frac"] = _num(sec["roofline"]["frac"])
    if isinstance(sec.get("cpu_baseline"), dict):
      out["secondary"]["cpu_value"] = _num(sec["cpu_baseline"].get("value"))
  legs = {}
  _walk({k: v for k, v in result.items() if k not in ("secondary",)}, "", legs)
  if isinstance(sec, dict):
    _walk(sec, "train_step", legs)
  # shorter names for the nested ones
  short = {}
  for name, leg in legs.items():
    name = name.replace("config_legs.", "").replace("streaming.batches.", "streaming.").replace("robustness.", "rob.")
    short[name] = leg
  out["legs"] = short
  out["detail"] = result.get("detail_file", "gpurun_out/bench_detail.json")
  # never past the limit: drop the longest-named legs first, then shorten the texts (a worst case the unit test builds)
  while len(json.dumps(out, separators=(",", ":"))) > limit and out["legs"]:
    out["legs"].pop(max(out["legs"], key=len))
    out["legs_truncated"] = True
  if len(json.dumps(out, separators=(",", ":"))) > limit:
    for sect in ("cpu_baseline", "roofline", "config"):
      if sect in out:
        out[sect] = {k: _text(v, 40) for k, v in out[sect].items()}
  return out


def dumps(result: dict) -> str:
  return json.dumps(compact(result), separators=(",", ":"))
