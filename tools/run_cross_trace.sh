#!/bin/bash
# kernel trace of tools/exp_cross_trace.py: the dispatches of the LAST training pair in order
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/cross_trace
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- python $ROOT/tools/exp_cross_trace.py > "$OUT/log.txt" 2>&1
cd "$ROOT"
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "tfrs" in r["Kernel_Name"] or "g16" in r["Kernel_Name"]]
tail = rows[-40:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
  print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} us  {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f} us  {r["Kernel_Name"][:90]}')
PY
