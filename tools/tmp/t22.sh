#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/tmp/exp_wf.py 2>&1 | grep "^{"
DIM=64 NQS=64,128,129,192,256,257 python tools/tmp/exp_wf.py 2>&1 | grep "^{"
DIM=32 NQS=128,256,257 python tools/tmp/exp_wf.py 2>&1 | grep "^{"
