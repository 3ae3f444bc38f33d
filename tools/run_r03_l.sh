set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
{ echo "# tools/exp_zipf.py: Zipf-duplicated corpus, distinct-row index off (TFRS_EXP_DEDUP=0) and on"; TFRS_EXP_DEDUP=0 python tools/exp_zipf.py 2>&1 | grep zipf; echo "# with the distinct-row index (default of BruteForce.index)"; TFRS_EXP_DEDUP=1 python tools/exp_zipf.py 2>&1 | grep zipf; echo "# tools/exp_neardup.py: near-duplicate clusters re-scored inside the list kernel"; python tools/exp_neardup.py 2>&1 | grep cluster; } > $O/robustness.txt
cat $O/robustness.txt
python tools/exp_survivors.py '[{}, {"TFRS_TOPK_STAT_PFAIL":"1e-5"}, {"TFRS_TOPK_STAT_PFAIL":"1e-3"}, {"TFRS_TOPK_STAT_PFAIL":"1e-2"}, {"TFRS_TOPK_STAT":"0"}, {}]' 2>&1 | grep env > $O/threshold_plans.jsonl; cat $O/threshold_plans.jsonl
python tools/exp_rank_count.py 2>&1 | tail -1 > $O/rank_count.txt; cat $O/rank_count.txt
