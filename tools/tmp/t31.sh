#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_topk_gpu.py tests/test_fuzz_gpu.py tests/test_baseline_configs_gpu.py -x -q -k "stream" 2>&1 | tail -2
for s in 41 42; do timeout 280 python tools/fuzz_streaming.py $s 25 2>&1 | tail -1; done
NQS=64,65,100,128 python tools/tmp/exp_half.py 2>&1 | grep "^{"
