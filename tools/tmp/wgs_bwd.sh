#!/bin/bash
cd $GRAFT_REPO_ROOT
for b in 2176 4096 8192 16384 32768; do for w in 256 512; do
  reps=200; [ $b -ge 16384 ] && reps=30
  echo -n "WGS_BWD=$w: "; TFRS_SOFTMAX_WGS_BWD=$w python tools/exp_sm16_ms.py $b 64 $reps 2>&1 | tail -1
done; done
for w in 256 512; do echo -n "WGS_BWD=$w step: "; TFRS_SOFTMAX_WGS_BWD=$w python tools/exp_trainstep_graph.py 3000 2>&1 | tail -1; done
for w in 256 512; do echo -n "WGS_BWD=$w step: "; TFRS_SOFTMAX_WGS_BWD=$w python tools/exp_trainstep_graph.py 3000 2>&1 | tail -1; done
