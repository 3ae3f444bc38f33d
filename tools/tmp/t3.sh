#!/bin/bash
cd $GRAFT_REPO_ROOT
for m in 0 1 2; do for i in 1 2; do echo -n "BRANCHES=$m: "; TFRS_STEP_BRANCHES=$m python tools/exp_trainstep_graph.py 3000 2>&1 | tail -1; done; done
