#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q -k "stream or topk or select or bruteforce or fuzz" 2>&1 | tail -3
for m in 1 4 8 1 4; do
echo -n "DENSE_KP=$m: "; TFRS_SELECT_DENSE_KP=$m python bench.py --steps 10 --warmup 3 --no-train-step --no-gather --no-scale-workload --no-robustness --no-config-legs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print({k.replace('streaming.batch_','B'):round(v['ms'],4) for k,v in d['legs'].items() if k.startswith('streaming.')})"
done
