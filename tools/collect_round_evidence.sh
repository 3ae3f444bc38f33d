#!/bin/bash
# Collects everything a round's profiles/ entries are made from (run on the GPU box through gpurun):
#   tools/collect_round_evidence.sh <tag>      ->  gpurun_out/<tag>/...
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-evidence}
O=$ROOT/gpurun_out/$TAG
mkdir -p "$O"
cd "$ROOT"
python -m pytest tests -m gpu -q > "$O/pytest_gpu.log" 2>&1; echo "rc=$?" >> "$O/pytest_gpu.log"
bash tools/profile_bench.sh "$TAG/prof" > "$O/profile.log" 2>&1
python bench.py > "$O/bench.json" 2> "$O/bench.err"
python tools/summarize_profile.py "$O/prof" "$O/bench_topk.md" "$O/bench.json" > "$O/summ.log" 2>&1
python tools/bench_interaction.py > "$O/interaction.jsonl" 2> "$O/interaction.err"
python tools/bench_scatter.py > "$O/scatter.jsonl" 2>&1
python tools/bench_streaming.py > "$O/streaming.jsonl" 2>&1
python tools/bench_ranking.py > "$O/ranking.jsonl" 2>&1
python tools/bench_clustered.py > "$O/clustered.jsonl" 2>&1
python tools/bench_ops.py > "$O/bench_ops.jsonl" 2>&1
python tools/bench_batch_sweep.py > "$O/batch_sweep.jsonl" 2>&1
python tools/bench_frontends.py > "$O/frontends.jsonl" 2>&1
python tools/exp_fit.py > "$O/fit.txt" 2>&1
# the N > 1 code paths on the one GPU of the box (two gloo ranks on cuda:0; one-rank RCCL group): NOT scaling numbers
TFRS_BENCH_ONE_GPU=1 TFRS_BENCH_ROWS=4000000 python bench.py --gpus 2 --steps 3 --warmup 1 2>/dev/null | grep '^{' > "$O/two_rank.json"
TFRS_BENCH_ONE_GPU=1 TFRS_BENCH_ROWS=4000000 python bench.py --gpus 2 --workload streaming128 --steps 3 --warmup 1 2>/dev/null | grep '^{' >> "$O/two_rank.json"
TFRS_BENCH_ONE_GPU=1 TFRS_BENCH_TABLE_ROWS=20000000 TFRS_BENCH_BATCH=8192 python bench.py --gpus 2 --workload dlrm_embedding --steps 3 --warmup 1 2>/dev/null | grep '^{' >> "$O/two_rank.json"
TFRS_BENCH_FORCE_DIST=1 TFRS_FORCE_EXCHANGE=1 TFRS_BENCH_ROWS=12500000 python bench.py --gpus 1 --workload streaming128 --steps 3 --warmup 1 2>/dev/null | grep '^{' > "$O/rccl_one_rank.json"
TFRS_BENCH_FORCE_DIST=1 TFRS_FORCE_EXCHANGE=1 TFRS_BENCH_TABLE_ROWS=100000000 python bench.py --gpus 1 --workload dlrm_embedding --steps 5 --warmup 2 2>/dev/null | grep '^{' >> "$O/rccl_one_rank.json"
python tools/exp_power.py > "$O/power.jsonl" 2>&1
for m in random zeros; do ./tools/ubench/mfma_rate $m; done > "$O/mfma_rate.txt" 2>&1
bash tools/pmc_generic.sh "$TAG/pmc_inter" tools/exp_interaction_prof.py > /dev/null 2>&1
python tools/pmc_summary.py "$O/pmc_inter" > "$O/pmc_inter.txt" 2>&1
python tools/print_kernel_stats.py "$O/pmc_inter/trace/bench_kernel_stats.csv" 30 > "$O/inter_kernel_stats.txt" 2>&1
bash tools/pmc_generic.sh "$TAG/pmc_train" tools/exp_trainstep_graph.py > /dev/null 2>&1
python tools/pmc_summary.py "$O/pmc_train" > "$O/pmc_train.txt" 2>&1
python tools/print_kernel_stats.py "$O/pmc_train/trace/bench_kernel_stats.csv" 30 > "$O/train_kernel_stats.txt" 2>&1
tail -2 "$O/pytest_gpu.log"
