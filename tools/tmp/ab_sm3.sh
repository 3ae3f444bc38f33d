#!/bin/bash
cd $GRAFT_REPO_ROOT
cp recommenders_amd/libtfrs_hip.so /tmp/lib_orig.so
for rep in 1 2; do for v in vgpr agpr; do
  cp ab/lib_$v.so recommenders_amd/libtfrs_hip.so
  for bv in 1 0; do
  echo -n "$v BWD_V=$bv: "; TFRS_SOFTMAX_BWD_V=$bv python tools/exp_sm16_ms.py 4096 64 300 2>&1 | tail -1
  done
  echo -n "$v BWD_V=0 16k: "; TFRS_SOFTMAX_BWD_V=0 python tools/exp_sm16_ms.py 16384 64 50 2>&1 | tail -1
  echo -n "$v BWD_V=1 16k: "; TFRS_SOFTMAX_BWD_V=1 python tools/exp_sm16_ms.py 16384 64 50 2>&1 | tail -1
done; done
cp /tmp/lib_orig.so recommenders_amd/libtfrs_hip.so
