#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "copy_multi or fit or train_step or graph" 2>&1 | tail -4
timeout 300 python tools/exp_fit.py 2>&1 | grep -E "replay|fit epoch|graph.replay|captured" | head -8
