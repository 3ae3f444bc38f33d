for rep in 1 2; do
for cfg in "3 0" "2 0" "2 384" "2 256" "3 384"; do set -- $cfg
for m in 0 1; do
echo -n "nb=$1 bwd_wgs=$2 metrics=$m: "; TFRS_SOFTMAX_BWD_NB=$1 TFRS_SOFTMAX_BWD_WGS=$2 TFRS_EXP_METRICS=$m python tools/exp_trainstep_graph.py 2000 2>&1 | grep graphed
done; done; done
python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "softmax or retrieval or quickstart or train" 2>&1 | tail -3
