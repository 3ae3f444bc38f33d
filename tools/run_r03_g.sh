set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -4
python tools/fuzz_topk.py 301 24 2>&1 | tail -3
python tools/fuzz_streaming.py 302 12 2>&1 | tail -3
python tools/fuzz_embedding.py 303 12 2>&1 | tail -2
python tools/exp_rank_count.py 2>&1 | tail -1
python tools/bench_sharded_embedding.py > gpurun_out/r03_sharded_embedding.jsonl 2> gpurun_out/r03_sharded_embedding.err; cat gpurun_out/r03_sharded_embedding.jsonl; tail -3 gpurun_out/r03_sharded_embedding.err
