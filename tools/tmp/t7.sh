#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q -k "stream or topk or select or bruteforce or fuzz" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-train-step --no-gather --no-scale-workload --no-robustness --no-config-legs --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_quick.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_quick.json').read())
print("headline ms", d["ms_per_step"], "median", d.get("step_ms_median"), "filter ms", d["roofline"]["avg_launch_ms"])
print({k:v['ms'] for k,v in d["legs"].items() if k.startswith("streaming")})
PY
