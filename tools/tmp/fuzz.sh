for s in 11 12 13 14 15 16; do
  for f in fuzz_topk fuzz_streaming fuzz_gemm fuzz_embedding; do
    echo "== $f seed $s"; timeout 280 python tools/$f.py $s 25 2>&1 | tail -1
  done
done
