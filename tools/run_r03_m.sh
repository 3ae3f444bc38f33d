set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03z; mkdir -p $O
python tools/bench_interaction.py > "$O/interaction.jsonl" 2> "$O/interaction.err"
python tools/bench_ranking.py > "$O/ranking.jsonl" 2>&1
bash tools/pmc_generic.sh "r03z/pmc_inter" tools/exp_interaction_prof.py > /dev/null 2>&1
python tools/pmc_summary.py "$O/pmc_inter" > "$O/pmc_inter.txt" 2>&1
python tools/print_kernel_stats.py "$O/pmc_inter/trace/bench_kernel_stats.csv" 30 > "$O/inter_kernel_stats.txt" 2>&1
python tools/exp_dotfwd.py > "$O/exp_dotfwd.jsonl" 2>&1
python tools/exp_dotbwd.py > "$O/exp_dotbwd.jsonl" 2>&1
grep cross "$O/interaction.jsonl" | cut -c1-150; cut -c1-220 "$O/ranking.jsonl" | grep config
