#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 0 1; do
  echo "BWD_V=$v"; TFRS_SOFTMAX_BWD_V=$v python tools/exp_sm16_ms.py 4096 64 300 2>&1 | tail -1
  TFRS_SOFTMAX_BWD_V=$v python tools/exp_sm16_ms.py 4000 64 300 2>&1 | tail -1
  TFRS_SOFTMAX_BWD_V=$v python tools/exp_sm16_ms.py 16384 64 50 2>&1 | tail -1
  TFRS_SOFTMAX_BWD_V=$v python tools/exp_sm16_ms.py 4096 32 300 2>&1 | tail -1
  echo -n "step: "; TFRS_SOFTMAX_BWD_V=$v python tools/exp_trainstep_graph.py 2000 2>&1 | tail -1
done
timeout 900 python -m pytest tests -m gpu -x -q -k "softmax or train_step or retrieval or fit" 2>&1 | tail -8
