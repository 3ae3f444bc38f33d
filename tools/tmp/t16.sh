#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp TFRS_ALLOW_ABLATION=1
cp recommenders_amd/libtfrs_hip.so /tmp/lib_orig.so
for v in 0 1 2 4; do
  cp ab/lib_$v.so recommenders_amd/libtfrs_hip.so
  for b in 1 128; do
    BATCH=$b CALLS=4 rocprofv3 --kernel-trace -d gpurun_out/l16_${v}_$b -o t --output-format csv -- python tools/exp_streaming_prof.py > /dev/null 2>&1
  done
done
cp /tmp/lib_orig.so recommenders_amd/libtfrs_hip.so
