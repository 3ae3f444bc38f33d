set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_topk_gpu.py tests/test_baseline_configs_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/k_trace -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-train-step --no-gather --no-scale-workload --no-robustness > $GRAFT_REPO_ROOT/gpurun_out/k_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/print_kernel_stats.py $(find gpurun_out/k_trace -name "*kernel_stats.csv" | head -1) 8
python bench.py --no-cpu-baseline --no-train-step --no-gather --no-scale-workload --no-robustness 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['step_ms_median'], r['roofline']['frac'])"
