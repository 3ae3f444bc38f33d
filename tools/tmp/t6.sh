#!/bin/bash
cd $GRAFT_REPO_ROOT
for m in x m; do echo -n "FINALIZE=$m: "; TFRS_SOFTMAX_FINALIZE=$m python tools/exp_sm16_ms.py 4096 64 300 2>&1 | tail -1; done
for m in x m; do for i in 1 2; do echo -n "FINALIZE=$m step: "; TFRS_SOFTMAX_FINALIZE=$m python tools/exp_trainstep_graph.py 3000 2>&1 | tail -1; done; done
timeout 900 python -m pytest tests -m gpu -x -q -k "softmax or train_step or retrieval or fit or quickstart" 2>&1 | tail -3
