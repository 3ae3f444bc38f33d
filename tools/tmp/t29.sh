#!/bin/bash
cd $GRAFT_REPO_ROOT
export TFRS_ALLOW_ABLATION=1
cp recommenders_amd/libtfrs_hip.so /tmp/lib_orig.so
cp /tmp/lib_orig.so ab/lib_orig.so
for rep in 1 2 3; do
for v in orig 0 1; do
  cp ab/lib_$v.so recommenders_amd/libtfrs_hip.so
  echo "== pairs $v rep $rep"; NQS=33,64,96,128 python tools/tmp/exp_w.py 2>&1 | grep "^{"
  [ $rep = 1 ] && DIM=64 NQS=64,128,256 python tools/tmp/exp_w.py 2>&1 | grep "^{"
done
done
cp /tmp/lib_orig.so recommenders_amd/libtfrs_hip.so
