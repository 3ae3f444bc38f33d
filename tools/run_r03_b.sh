set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ts_trace -o ts -- python $GRAFT_REPO_ROOT/tools/exp_trainstep_graph.py 300 > $GRAFT_REPO_ROOT/gpurun_out/ts_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/print_kernel_stats.py $(find gpurun_out/ts_trace -name "*kernel_stats.csv" | head -1) 30
tail -2 gpurun_out/ts_trace.log
