#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for b in 1 128; do
  BATCH=$b CALLS=4 rocprofv3 --kernel-trace -d gpurun_out/st6_$b -o t --output-format csv -- python tools/exp_streaming_prof.py > gpurun_out/st6_$b.log 2>&1
done
python tools/exp_stream_small.py 2>&1 | tail -12
