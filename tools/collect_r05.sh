#!/bin/bash
# Round-5 evidence (run on the GPU box through gpurun; ~8 min):   tools/collect_r05.sh <tag>  ->  gpurun_out/<tag>/...
# GPU suite, rocprofv3 kernel trace + PMC passes of bench.py's headline step, the full bench line, the N > 1 code paths on
# the one GPU of the box (NOT scaling numbers), the saturating MFMA loop.  tools/publish_evidence_r05.sh copies the
# summaries into profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}
O=$ROOT/gpurun_out/$TAG
mkdir -p "$O"
cd "$ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > "$O/pytest_gpu.log" 2>&1; echo "rc=$?" >> "$O/pytest_gpu.log"
timeout 900 bash tools/profile_bench.sh "$TAG/prof" > "$O/profile.log" 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > "$O/bench.json" 2> "$O/bench.err"
python tools/summarize_profile.py "$O/prof" "$O/bench_topk.md" "$O/bench.json" > "$O/summ.log" 2>&1
if [ -z "${QUICK:-}" ]; then   # (QUICK=1: a later collection of the same round -- these three do not depend on the kernels changed since)
TFRS_BENCH_ONE_GPU=1 TFRS_BENCH_ROWS=4000000 timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 2>/dev/null | grep '^{' > "$O/two_rank.json"
TFRS_BENCH_ONE_GPU=1 TFRS_BENCH_ROWS=4000000 timeout 300 python bench.py --gpus 2 --workload streaming128 --steps 3 --warmup 1 2>/dev/null | grep '^{' >> "$O/two_rank.json"
TFRS_BENCH_FORCE_DIST=1 TFRS_FORCE_EXCHANGE=1 TFRS_BENCH_ROWS=12500000 timeout 300 python bench.py --gpus 1 --workload streaming128 --steps 3 --warmup 1 2>/dev/null | grep '^{' > "$O/rccl_one_rank.json"
timeout 120 ./tools/ubench/mfma_peak > "$O/mfma_peak.txt" 2>&1
fi
tail -3 "$O/pytest_gpu.log"
head -c 600 "$O/bench.json"; echo
timeout 300 python __graft_entry__.py smoke > "$O/smoke.log" 2>&1; tail -1 "$O/smoke.log"
