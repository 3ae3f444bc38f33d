#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/exp_bf_nt.py 2>&1 | grep "^{"
timeout 1200 python -m pytest tests/test_topk_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -2
