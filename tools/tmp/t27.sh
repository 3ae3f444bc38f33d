#!/bin/bash
cd $GRAFT_REPO_ROOT
NQS=512,513,640,768,896,1024,1025,1536 python tools/tmp/exp_w.py 2>&1 | grep "^{"
DIM=64 NQS=512,513,768,1024,1025 python tools/tmp/exp_w.py 2>&1 | grep "^{"
