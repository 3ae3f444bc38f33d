#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/exp_stream_nt.py 2>&1 | grep "^{"
DIM=64 NQS=1,32,33,64,128,256 python tools/exp_stream_nt.py 2>&1 | grep "^{"
timeout 900 python -m pytest tests/test_topk_gpu.py tests/test_fuzz_gpu.py -x -q -k "stream" 2>&1 | tail -2
