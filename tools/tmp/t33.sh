#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in 1 0; do echo "scatter_nt=$v $(TFRS_SCATTER_NT=$v python tools/bench_scatter.py 2>&1 | grep '^{' | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms"],4))')"; done; done
timeout 900 python -m pytest tests -m gpu -x -q -k "embedding or adagrad or scatter or config" 2>&1 | tail -2
timeout 280 python tools/fuzz_embedding.py 51 25 2>&1 | tail -1
