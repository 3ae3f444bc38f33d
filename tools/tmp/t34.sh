#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp TFRS_ALLOW_ABLATION=1
cp recommenders_amd/libtfrs_hip.so /tmp/lib_orig.so
for rep in 1 2; do
for v in 0 1; do
  cp ab/lib_$v.so recommenders_amd/libtfrs_hip.so
  rm -rf /tmp/hp; CALLS=30 rocprofv3 --kernel-trace --stats -d /tmp/hp -o t --output-format csv -- python tools/exp_headline_prof.py > /dev/null 2>&1
  echo "== rownt $v rep $rep"; python tools/print_kernel_stats.py $(find /tmp/hp -name "*kernel_stats.csv") 6 | grep -E "list_topk16"
done
done
cp /tmp/lib_orig.so recommenders_amd/libtfrs_hip.so
