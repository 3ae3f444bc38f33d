#!/bin/bash
# Copies what tools/collect_r05.sh left under gpurun_out/<tag>/ into the tracked profiles/<round>_* files:
#   tools/publish_evidence_r05.sh <tag> <round>      (e.g. r05a r05)
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
S=$ROOT/gpurun_out/$1
R=$ROOT/profiles/$2
cp "$S/bench.json" "${R}_bench.json"
cp "$S/bench_topk.md" "${R}_bench_topk.md"
cp "$S/prof/trace/bench_kernel_stats.csv" "${R}_bench_kernel_stats.csv"
cp "$S/bench_topk_traffic.json" "${R}_bench_topk_traffic.json"
cp "$S/bench_topk_traffic.json" "$ROOT/profiles/latest_traffic.json"
# (absent after a QUICK=1 collection: the files of the round's earlier collection stay)
[ -f "$S/two_rank.json" ] && grep '^{' "$S/two_rank.json" > "${R}_two_rank_one_gpu_dryrun.json"
[ -f "$S/rccl_one_rank.json" ] && grep '^{' "$S/rccl_one_rank.json" > "${R}_rccl_one_rank_dryrun.json"
[ -f "$S/mfma_peak.txt" ] && cp "$S/mfma_peak.txt" "${R}_mfma_peak.txt"
tail -4 "$S/pytest_gpu.log" > "${R}_pytest_gpu_tail.txt"
python "$ROOT/tools/summarize_errors.py" "$ROOT/gpurun_out/observed_errors.jsonl" > "${R}_observed_errors.md" 2>/dev/null || true
echo "published $1 -> profiles/$2_*"
