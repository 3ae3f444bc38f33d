set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -8
python tools/exp_host_overhead.py 2>&1 | tail -32
python tools/exp_redo.py 2>&1 | tail -5
