#!/bin/bash
cd $GRAFT_REPO_ROOT
cp recommenders_amd/libtfrs_hip.so /tmp/lib_orig.so
for rep in 1 2 3; do
for v in 0 1; do
  cp ab/lib_$v.so recommenders_amd/libtfrs_hip.so
  echo "== global_as $v rep $rep"; NQS=1,64,128,257,512,1024 python tools/tmp/exp_w.py 2>&1 | grep "^{"
done
done
cp /tmp/lib_orig.so recommenders_amd/libtfrs_hip.so
