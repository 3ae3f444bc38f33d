B="python bench.py --no-cpu-baseline --no-train-step --no-gather --no-scale-workload --no-robustness --no-streaming"
P='import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%s ms_step %.4f kernel %.4f frac %.4f thr %.4f redo %s" % (sys.argv[1], r["ms_per_step"], r["roofline"]["avg_launch_ms"], r["roofline"]["frac"], r["roofline"]["f16_threshold_pass_ms_per_step"], r["redo_queries_last_step"]))'
$B 2>/dev/null | python -c "$P" default
TFRS_TOPK_STAT_PFAIL=1e-5 $B 2>/dev/null | python -c "$P" pfail1e-5
TFRS_TOPK_STAT_PFAIL=1e-3 $B 2>/dev/null | python -c "$P" pfail1e-3
TFRS_TOPK_STAT_PFAIL=1e-2 $B 2>/dev/null | python -c "$P" pfail1e-2
TFRS_TOPK_SAMPLE_STAT=8 $B 2>/dev/null | python -c "$P" stride8
TFRS_TOPK_SAMPLE_STAT=6 $B 2>/dev/null | python -c "$P" stride6
TFRS_TOPK_WGS=768 $B 2>/dev/null | python -c "$P" wgs768
TFRS_TOPK_WGS=1024 $B 2>/dev/null | python -c "$P" wgs1024
TFRS_SCAN16_DRAIN_EVERY=8 $B 2>/dev/null | python -c "$P" drain8
TFRS_SCAN16_DRAIN_EVERY=2 $B 2>/dev/null | python -c "$P" drain2
