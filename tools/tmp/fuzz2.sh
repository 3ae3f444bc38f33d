#!/bin/bash
cd $GRAFT_REPO_ROOT
for s in 21 22 23 24 25 26 27 28; do
  for f in fuzz_topk fuzz_streaming fuzz_gemm fuzz_embedding; do
    echo "== $f seed $s"; timeout 280 python tools/$f.py $s 25 2>&1 | tail -1
  done
done > gpurun_out/fuzz2.log 2>&1
grep -c "==" gpurun_out/fuzz2.log; grep -v "==" gpurun_out/fuzz2.log | sort | uniq -c | sort -rn | head -20
