#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cp recommenders_amd/libtfrs_hip.so /tmp/lib_orig.so
for rep in 1 2; do
for v in 24 20 18; do
  cp ab/lib_$v.so recommenders_amd/libtfrs_hip.so
  rm -rf /tmp/hp; CALLS=30 rocprofv3 --kernel-trace --stats -d /tmp/hp -o t --output-format csv -- python tools/exp_headline_prof.py > /dev/null 2>&1
  echo "== bits $v rep $rep"; python tools/print_kernel_stats.py $(find /tmp/hp -name "*kernel_stats.csv") 6 | grep -E "list_topk16|bin_threshold|scan16f"
done
done
cp /tmp/lib_orig.so recommenders_amd/libtfrs_hip.so
# fused merge: tests + timing
timeout 900 python -m pytest tests/test_topk_gpu.py -x -q -k "stream" 2>&1 | tail -3
for f in 1 0; do echo "== fused $f"; TFRS_STREAM_FUSED_MERGE=$f python tools/exp_stream_small.py 2>&1 | grep nq; done
for f in 1 0; do echo "== fused $f"; TFRS_STREAM_FUSED_MERGE=$f python tools/exp_stream_small.py 2>&1 | grep nq; done
