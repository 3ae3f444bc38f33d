#!/bin/bash
# Round-4 evidence for the grouped Streaming path (run on the GPU box through gpurun):
#   tools/collect_r04_streaming.sh <tag>  ->  gpurun_out/<tag>/{stream.jsonl, b<B>/..., *.txt}
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/${1:-r04s}
mkdir -p "$O"
cd "$ROOT"
BATCHES=8192,1024,256,128,64,32,1 python tools/bench_streaming.py > "$O/stream.jsonl" 2> "$O/stream.err"
cd /tmp && export TMPDIR=/tmp
for B in 1 64 128 8192; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/b$B/trace" -o bench -- env BATCH=$B python $ROOT/tools/exp_streaming_prof.py > "$O/b$B.trace.log" 2>&1
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$O/b$B/pmc_fetch" -o bench -- env BATCH=$B CALLS=3 python $ROOT/tools/exp_streaming_prof.py > "$O/b$B.fetch.log" 2>&1
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$O/b$B/pmc_write" -o bench -- env BATCH=$B CALLS=3 python $ROOT/tools/exp_streaming_prof.py > "$O/b$B.write.log" 2>&1
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d "$O/b$B/pmc_sq" -o bench -- env BATCH=$B CALLS=3 python $ROOT/tools/exp_streaming_prof.py > "$O/b$B.sq.log" 2>&1
  python $ROOT/tools/print_kernel_stats.py "$O/b$B/trace/bench_kernel_stats.csv" 16 > "$O/b$B.kernel_stats.txt" 2>&1
  python $ROOT/tools/pmc_summary.py "$O/b$B" > "$O/b$B.pmc.txt" 2>&1
done
cd "$ROOT"
cat "$O/stream.jsonl" | cut -c1-300
