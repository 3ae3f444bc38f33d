#!/bin/bash
cd $GRAFT_REPO_ROOT
cp recommenders_amd/libtfrs_hip.so /tmp/lib_orig.so
for rep in 1 2; do for v in ${VARIANTS}; do
  cp ab/lib_$v.so recommenders_amd/libtfrs_hip.so
  echo -n "ablate $v: "; TFRS_SOFTMAX_BWD_V=1 python tools/exp_sm16_ms.py 4096 64 300 2>&1 | tail -1
done; done
cp ab/lib_0.so recommenders_amd/libtfrs_hip.so
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/tools/exp_sm16_ms.py 4096 64 100 > /dev/null 2>&1; python - <<'PY'
import csv,glob
f=glob.glob('/tmp/tr/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
  print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
cp /tmp/lib_orig.so $GRAFT_REPO_ROOT/recommenders_amd/libtfrs_hip.so
