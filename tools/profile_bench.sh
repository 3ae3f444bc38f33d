#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box:
#   1. kernel trace + stats (CSV) of bench.py's default protocol (10 warm-up + 50 timed steps)  -> $OUT/trace
#   2. PMC pass A: clocks / MFMA busy / waits  -> $OUT/pmc_a
#   3. PMC pass B: FETCH_SIZE                  -> $OUT/pmc_fetch
#   4. PMC pass C: WRITE_SIZE                  -> $OUT/pmc_write
# Counter passes are separate runs with --kernel-trace only (no sys/hip/hsa tracing).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-prof}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-step --no-gather --no-scale-workload --no-robustness --no-streaming --no-config-legs"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python $ROOT/bench.py --no-cpu-baseline --no-train-step --no-gather --no-scale-workload --no-robustness --no-streaming --no-config-legs > "$OUT/trace.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 -d "$OUT/pmc_a" -o bench -- $BENCH > "$OUT/pmc_a.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d "$OUT/pmc_b" -o bench -- $BENCH > "$OUT/pmc_b.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o bench -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write" -o bench -- $BENCH > "$OUT/pmc_write.log" 2>&1
cd "$ROOT"
find "$OUT" -name "*.csv" | head -30
