set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_topk_gpu.py tests/test_baseline_configs_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x 2>&1 | tail -4
python tools/exp_neardup.py 2>&1 | tail -5
python tools/fuzz_topk.py 401 16 2>&1 | tail -1
