#!/bin/bash
# HBM traffic of the kernels of the README train step (graph replay, tools/exp_trainstep_graph.py): FETCH_SIZE and
# WRITE_SIZE in separate --pmc passes (kernel trace only), per launch and per kernel -> gpurun_out/<tag>/trainstep_traffic.json
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/${1:-r05t}
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$O/pmc_fetch" -o b -- python $ROOT/tools/exp_trainstep_graph.py 50 > "$O/fetch.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$O/pmc_write" -o b -- python $ROOT/tools/exp_trainstep_graph.py 50 > "$O/write.log" 2>&1
python - "$O" <<'PY'
import csv, glob, json, re, sys, collections
o = sys.argv[1]
out = collections.defaultdict(dict)
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
  acc = collections.defaultdict(list)
  for f in glob.glob(o + "/" + sub + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
      if r["Counter_Name"] != name: continue
      k = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"]))
      if k.startswith("tfrs::"): acc[k].append(float(r["Counter_Value"]))
  for k, v in acc.items():
    out[k][name + "_KiB_per_launch"] = sum(v) / len(v)
    out[k]["launches_" + name] = len(v)
for k, v in out.items():
  v["fetch_bytes_corrected"] = 2.0 * 1024.0 * v.get("FETCH_SIZE_KiB_per_launch", 0.0)   # gfx950 x2 correction (MI355X_MICROARCH.md)
  v["write_bytes"] = 1024.0 * v.get("WRITE_SIZE_KiB_per_launch", 0.0)
json.dump({"source": "tools/run_trainstep_traffic.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/exp_trainstep_graph.py",
           "kernels": out}, open(o + "/trainstep_traffic.json", "w"), indent=1)
print(json.dumps({k: (round(v["fetch_bytes_corrected"]), round(v["write_bytes"])) for k, v in out.items()}))
PY
