#!/bin/bash
cd $GRAFT_REPO_ROOT
for nw in 4 8; do for wgs in 128 256 512 1024; do
  TFRS_SOFTMAX_NW=$nw TFRS_SOFTMAX_WGS=$wgs timeout 120 python tools/exp_sm16_ms.py 4096 64 300 2>&1 | tail -1
done; done
for nw in 4 8; do for wgs in 256 512; do
  echo "NW=$nw WGS=$wgs:"; TFRS_SOFTMAX_NW=$nw TFRS_SOFTMAX_WGS=$wgs timeout 120 python tools/exp_trainstep_graph.py 2000 2>&1 | tail -1
done; done
