#!/bin/bash
# Copies the summaries tools/collect_round_evidence.sh left under gpurun_out/<tag>/ into the
# tracked profiles/<round>_* files:   tools/publish_evidence.sh <tag> <round>   (e.g. r02f2 r02)
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
S=$ROOT/gpurun_out/$1
R=$ROOT/profiles/$2
cp "$S/bench.json" "${R}_bench.json"
cp "$S/bench_topk.md" "${R}_bench_topk.md"
cp "$S/prof/trace/bench_kernel_stats.csv" "${R}_bench_kernel_stats.csv"
cp "$S/bench_topk_traffic.json" "${R}_bench_topk_traffic.json"
cp "$S/bench_topk_traffic.json" "$ROOT/profiles/latest_traffic.json"
grep '^{' "$S/interaction.jsonl" > "${R}_interaction.jsonl"
grep '^{' "$S/scatter.jsonl" > "${R}_scatter.jsonl"
grep '^{' "$S/streaming.jsonl" > "${R}_streaming_c3_shard.jsonl"
grep '^{' "$S/ranking.jsonl" > "${R}_ranking_models.jsonl"
grep '^{' "$S/clustered.jsonl" > "${R}_clustered.jsonl"
grep '^{' "$S/bench_ops.jsonl" > "${R}_bench_ops.jsonl"
grep '^{' "$S/batch_sweep.jsonl" > "${R}_batch_sweep.jsonl"
grep '^{' "$S/power.jsonl" > "${R}_power.jsonl"
cp "$S/mfma_rate.txt" "${R}_mfma_rate.txt"
[ -f "$S/two_rank.json" ] && grep '^{' "$S/two_rank.json" > "${R}_two_rank_one_gpu_dryrun.json" || true
[ -f "$S/rccl_one_rank.json" ] && grep '^{' "$S/rccl_one_rank.json" > "${R}_rccl_one_rank_dryrun.json" || true
[ -f "$S/frontends.jsonl" ] && grep '^{' "$S/frontends.jsonl" > "${R}_frontends.jsonl" || true
[ -f "$S/fit.txt" ] && grep -v "^ \|ncalls\|Ordered\|List reduced\|function calls\|amdgpu.ids" "$S/fit.txt" | grep -v '^$' > "${R}_fit.txt" || true
{
  echo "# rocprofv3 summary: Cross (configs[3]) and DotInteraction (configs[4]) kernels"
  echo
  echo "Collected by \`tools/pmc_generic.sh\` on \`tools/exp_interaction_prof.py\` (kernel trace + stats; SQ counter passes in separate runs)."
  echo
  echo "## kernel stats"; echo '```'; cat "$S/inter_kernel_stats.txt"; echo '```'
  echo; echo "## counters"; echo '```'; cat "$S/pmc_inter.txt"; echo '```'
} > "${R}_interaction_kernels.md"
{
  echo "# rocprofv3 summary: the C1 two-tower train step replayed from a HIP graph"
  echo
  echo "Collected by \`tools/pmc_generic.sh\` on \`tools/exp_trainstep_graph.py\` (200 replays)."
  echo
  echo "## kernel stats"; echo '```'; cat "$S/train_kernel_stats.txt"; echo '```'
  echo; echo "## counters"; echo '```'; cat "$S/pmc_train.txt"; echo '```'
} > "${R}_trainstep_kernels.md"
echo "published $1 -> profiles/$2_*"
