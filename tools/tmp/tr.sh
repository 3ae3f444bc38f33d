#!/bin/bash
cd $GRAFT_REPO_ROOT
cp recommenders_amd/libtfrs_hip.so /tmp/lib_orig.so
cp ab/lib_256.so recommenders_amd/libtfrs_hip.so
for v in 1 0; do
echo "== BWD_V=$v WGS=256 (16 steps)"
TFRS_SOFTMAX_BWD_V=$v TFRS_SOFTMAX_WGS=256 python tools/exp_sm16_trace.py 16 2>&1 | tail -35 | head -12
TFRS_SOFTMAX_BWD_V=$v TFRS_SOFTMAX_WGS=256 python tools/exp_sm16_ms.py 4096 64 300 | tail -1
echo "== BWD_V=$v WGS=512 (8 steps)"
TFRS_SOFTMAX_BWD_V=$v TFRS_SOFTMAX_WGS=512 python tools/exp_sm16_trace.py 8 2>&1 | tail -35 | head -8
done
cp /tmp/lib_orig.so recommenders_amd/libtfrs_hip.so
