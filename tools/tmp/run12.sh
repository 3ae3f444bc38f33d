python -m pytest tests/test_ops_gpu.py tests/test_baseline_configs_gpu.py tests/test_ranking.py -m gpu -x -q -k "cross or ranking or adagrad or trajectory or quickstart or train" 2>&1 | tail -5
python tools/exp_dlrm_prof.py 2>&1 | grep "step ms"
python tools/exp_dcn_prof.py 2>&1 | grep "step ms"
