#!/bin/bash
cd $GRAFT_REPO_ROOT
export TFRS_ALLOW_ABLATION=1
cp recommenders_amd/libtfrs_hip.so /tmp/lib_orig.so
for rep in 1 2; do
for v in 0 1 3 4 7; do
  cp ab/lib_$v.so recommenders_amd/libtfrs_hip.so
  echo "== scatter nt $v rep $rep $(python tools/bench_scatter.py 2>&1 | grep '^{' | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms"],4))')"
done
done
cp /tmp/lib_orig.so recommenders_amd/libtfrs_hip.so
