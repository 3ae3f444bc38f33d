set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_topk_gpu.py -m gpu -q -x -k "duplicate or dedup or beyond or exclu or identif or full_size" 2>&1 | tail -6
python tools/exp_zipf.py 2>&1 | tail -4
