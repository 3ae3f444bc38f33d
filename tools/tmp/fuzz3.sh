#!/bin/bash
cd $GRAFT_REPO_ROOT
for s in 61 62 63 64 65; do
  for f in fuzz_topk fuzz_streaming fuzz_embedding fuzz_gemm; do
    echo "== $f seed $s"; timeout 200 python tools/$f.py $s 20 2>&1 | tail -1
  done
done > gpurun_out/fuzz3.log 2>&1
grep -c "MISMATCHES 0" gpurun_out/fuzz3.log; grep "MISMATCHES" gpurun_out/fuzz3.log | sort | uniq -c
