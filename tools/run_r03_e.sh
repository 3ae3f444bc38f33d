set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ts_trace -o ts -- python $GRAFT_REPO_ROOT/tools/exp_trainstep_graph.py 300 > $GRAFT_REPO_ROOT/gpurun_out/ts_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/print_kernel_stats.py $(find gpurun_out/ts_trace -name "*kernel_stats.csv" | head -1) 12
grep graphed gpurun_out/ts_trace.log
python bench.py --no-scale-workload > gpurun_out/bench_r03_e.json 2> gpurun_out/bench_r03_e.err
python - <<'PY'
import json
r=json.load(open('gpurun_out/bench_r03_e.json'))
print({k:r[k] for k in ('value','ms_per_step','step_ms_median')})
print(r['roofline']['frac'], r['roofline']['avg_launch_ms'])
print({k:(v['ms_per_step'],v['vs_iid_step'],v['redo_queries_last_step']) for k,v in r['robustness'].items()})
s=r['secondary']; print(s['value'], s['ms_per_step'], s['eager_ms_per_step'], s['metrics_off']['ms_per_step'])
PY
