python -m pytest tests/test_ops_gpu.py tests/test_baseline_configs_gpu.py tests/test_ranking.py -m gpu -x -q -k "split_fp16 or skinny or mlp or dense or ranking" 2>&1 | tail -2
WHICH=dlrm bash tools/tmp/run10.sh | head -14
for s in 0 1 2; do timeout 250 python tools/fuzz_gemm.py $s 25 2>&1 | tail -2; done
for s in 0 1; do timeout 250 python tools/fuzz_streaming.py $s 25 2>&1 | tail -2; done
