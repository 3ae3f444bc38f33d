#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for m in 33 1; do
  echo "== min_nq $m"; TFRS_STREAM_RAW16_MIN_NQ=$m DIM=64 NQS=1,16,32 python tools/tmp/exp_w.py 2>&1 | grep "^{"
  TFRS_STREAM_RAW16_MIN_NQ=$m DIM=32 NQS=1,32 python tools/tmp/exp_w.py 2>&1 | grep "^{"
done
done
