#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/pmc_generic.sh sm_pmc "tools/exp_sm16_ms.py 4096 64 20"
python tools/pmc_summary.py gpurun_out/sm_pmc sm16 > gpurun_out/sm_pmc/summary.txt 2>&1
python tools/pmc_summary.py gpurun_out/sm_pmc rank >> gpurun_out/sm_pmc/summary.txt 2>&1
cat gpurun_out/sm_pmc/summary.txt
