set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03z; mkdir -p $O
timeout 420 bash tools/pmc_generic.sh "r03z/pmc_inter" tools/exp_interaction_prof.py > /dev/null 2>&1
python tools/pmc_summary.py "$O/pmc_inter" > "$O/pmc_inter.txt" 2>&1
python tools/print_kernel_stats.py "$O/pmc_inter/trace/bench_kernel_stats.csv" 30 > "$O/inter_kernel_stats.txt" 2>&1
head -8 "$O/inter_kernel_stats.txt"
