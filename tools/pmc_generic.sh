#!/bin/bash
# usage: tools/pmc_generic.sh <outdir-under-gpurun_out> <python script> : kernel trace + 2 SQ PMC passes
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1
SCRIPT=$ROOT/$2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python $SCRIPT > "$OUT/trace.log" 2>&1
timeout 100 rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d "$OUT/pmc_a" -o bench -- python $SCRIPT > "$OUT/pmc_a.log" 2>&1
timeout 100 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d "$OUT/pmc_b" -o bench -- python $SCRIPT > "$OUT/pmc_b.log" 2>&1
timeout 100 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM -d "$OUT/pmc_c" -o bench -- python $SCRIPT > "$OUT/pmc_c.log" 2>&1
cd "$ROOT"
