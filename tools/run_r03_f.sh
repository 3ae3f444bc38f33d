set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "factorized or retrieval_golden or model_train or graph" 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ts_trace -o ts -- python $GRAFT_REPO_ROOT/tools/exp_trainstep_graph.py 300 > $GRAFT_REPO_ROOT/gpurun_out/ts_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/print_kernel_stats.py $(find gpurun_out/ts_trace -name "*kernel_stats.csv" | head -1) 10
grep graphed gpurun_out/ts_trace.log
bash tools/run_gather_evidence.sh r03_gather
