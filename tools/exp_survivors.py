"""Experiment: threshold-pass plans of the fp16-prefiltered search (sampling stride, statistical
rank) on the BASELINE configs[1] batch.  Configurations are interleaved over several rounds and the
median per configuration is reported (the chip's clock state drifts by several % within a run)."""
import ctypes, os, sys, time, json, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd import _lib
from recommenders_amd.layers import factorized_top_k as ftk

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(42)
corpus = torch.randn((1_000_000, 64), generator=g, device=dev) / 8.0
queries = torch.randn((8192, 64), generator=g, device=dev) / 8.0
index = ftk.BruteForce(k=100).index(corpus)
lib = _lib.load()
KEYS = ("TFRS_TOPK_STAT", "TFRS_TOPK_SAMPLE", "TFRS_TOPK_SAMPLE_STAT", "TFRS_TOPK_STAT_PFAIL", "TFRS_SCAN16_DRAIN",
        "TFRS_SCAN16_DRAIN_EVERY", "TFRS_TOPK_WGS", "TFRS_SCAN16_SHAPE")
ENVS = [{}, {"TFRS_SCAN16_DRAIN": "8", "TFRS_SCAN16_DRAIN_EVERY": "1000000"}, {"TFRS_SCAN16_DRAIN_EVERY": "2"},
        {"TFRS_SCAN16_DRAIN_EVERY": "6"}, {"TFRS_TOPK_STAT": "0"}, {}]
if len(sys.argv) > 1:       # a JSON list of environments on the command line replaces the default sweep
  ENVS = json.loads(sys.argv[1])
ref = None
acc = [dict(step=[], filt=[], binmax=[], redo=[], same=True) for _ in ENVS]
for rnd in range(5):
  for ei, env in enumerate(ENVS):
    for k in KEYS:
      os.environ.pop(k, None)
    os.environ.update(env)
    for _ in range(2):
      out = index(queries)
    torch.cuda.synchronize()
    if ref is None:
      ref = (out[0].clone(), out[1].clone())
    acc[ei]["same"] &= bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]))
    lib.tfrs_profile_enable(1)
    steps = 10
    t0 = time.perf_counter()
    for _ in range(steps):
      index(queries)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    res = {}
    for kind in (1, 2):
      ms, n, fl = ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
      lib.tfrs_profile_read_kind(kind, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl))
      res[kind] = ms.value / steps
    lib.tfrs_profile_read(None, None, None)
    lib.tfrs_profile_enable(0)
    acc[ei]["step"].append(dt * 1e3); acc[ei]["filt"].append(res[1]); acc[ei]["binmax"].append(res[2])
    acc[ei]["redo"].append(index.last_redo_count())
for env, a in zip(ENVS, acc):
  print(json.dumps({"env": env, "step_ms": round(statistics.median(a["step"]), 4),
                    "filter_ms": round(statistics.median(a["filt"]), 4),
                    "binmax_ms": round(statistics.median(a["binmax"]), 4),
                    "redo_max": max(a["redo"]), "same_as_first": a["same"]}), flush=True)
