#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/tmp/exp_wgs.py 2>&1 | grep "^{"
timeout 900 python -m pytest tests/test_topk_gpu.py tests/test_fuzz_gpu.py tests/test_baseline_configs_gpu.py -x -q -k "stream" 2>&1 | tail -2
timeout 250 python tools/fuzz_streaming.py 71 25 2>&1 | tail -1
