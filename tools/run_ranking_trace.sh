#!/bin/bash
# kernel trace + stats of tools/bench_ranking.py (DCN-v2 at configs[3], DLRM shard at configs[4])
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ranking_trace
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python $ROOT/tools/bench_ranking.py > "$OUT/log.txt" 2>&1
cd "$ROOT"
python tools/print_kernel_stats.py $(find "$OUT" -name "*kernel_stats.csv" | head -1) 28
