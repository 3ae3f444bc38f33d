#!/bin/bash
# rocprofv3 evidence for gather_kernel: kernel trace + stats, then FETCH_SIZE
# and WRITE_SIZE in their own runs (--kernel-trace only).  Output under gpurun_out/$1.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-gather}
mkdir -p "$OUT"
python $ROOT/tools/exp_gather_c3.py 30 > "$OUT/events.jsonl" 2> "$OUT/events.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o g -- python $ROOT/tools/exp_gather_c3.py 30 > "$OUT/trace.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o g -- python $ROOT/tools/exp_gather_c3.py 3 > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write" -o g -- python $ROOT/tools/exp_gather_c3.py 3 > "$OUT/pmc_write.log" 2>&1
cd $ROOT
cat "$OUT/events.jsonl"
python tools/print_kernel_stats.py $(find "$OUT/trace" -name "*kernel_stats.csv" | head -1) 6
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for name in ("pmc_fetch", "pmc_write"):
  for f in glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
      if "gather_kernel" in r["Kernel_Name"]:
        acc[(r["Counter_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for (c, grid), v in acc.items():
      print(name, c, "grid", grid, "launches", len(v), "avg", sum(v) / len(v))
PY
