set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
python bench.py --no-scale-workload --no-robustness --no-train-step --no-gather --no-cpu-baseline > gpurun_out/r03b/bench_short.json 2>/dev/null
bash tools/profile_bench.sh r03b/prof > gpurun_out/r03b/profile.log 2>&1
python tools/summarize_profile.py gpurun_out/r03b/prof gpurun_out/r03b/bench_topk.md gpurun_out/r03b/bench_short.json > gpurun_out/r03b/summ.log 2>&1
ls gpurun_out/r03b
