#!/bin/bash
cd $GRAFT_REPO_ROOT
cp recommenders_amd/libtfrs_hip.so /tmp/lib_orig.so
for rep in 1 2; do for lib in ab/lib_*.so; do
  cp $lib recommenders_amd/libtfrs_hip.so
  for wgs in 512 ${EXTRA_WGS}; do
    echo "== $lib rep $rep"; TFRS_SOFTMAX_WGS=$wgs python tools/exp_sm16_ms.py 4096 64 300 2>&1 | tail -1
  done
  echo -n "step: "; python tools/exp_trainstep_graph.py 2000 2>&1 | tail -1
done; done
cp /tmp/lib_orig.so recommenders_amd/libtfrs_hip.so
