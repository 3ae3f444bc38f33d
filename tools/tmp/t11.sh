#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cp recommenders_amd/libtfrs_hip.so /tmp/lib_orig.so
for v in 0 4 1 2 3; do
  cp ab/lib_$v.so recommenders_amd/libtfrs_hip.so
  rm -rf /tmp/abl; (cd /tmp; TFRS_ALLOW_ABLATION=1 BATCH=1 ROWS=1000000 DIM=64 CALLS=30 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl -o p -- python $GRAFT_REPO_ROOT/tools/exp_bruteforce_small.py > /dev/null 2>&1)
  f=$(find /tmp/abl -name '*kernel_stats.csv' | head -1)
  echo -n "ablate $v: "; python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
  if 'list_topk16' in r['Name']: print("list_topk16 calls %s avg_us %.1f min %.1f" % (r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"])/1e3))
PY
done
cp /tmp/lib_orig.so recommenders_amd/libtfrs_hip.so
