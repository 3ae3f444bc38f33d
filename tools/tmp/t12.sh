#!/bin/bash
cd $GRAFT_REPO_ROOT
cp recommenders_amd/libtfrs_hip.so /tmp/lib_orig.so
for rep in 1 2; do
for lib in a_old b_new; do
  cp ab/lib_$lib.so recommenders_amd/libtfrs_hip.so
  echo -n "== $lib rep $rep streaming: "
  python bench.py --steps 10 --warmup 3 --no-train-step --no-gather --no-scale-workload --no-robustness --no-config-legs --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print({k.replace('streaming.batch_','B'):round(v['ms'],4) for k,v in d['legs'].items() if k.startswith('streaming.')})"
done
for lib in c_oldtrain b_new; do
  cp ab/lib_$lib.so recommenders_amd/libtfrs_hip.so
  echo -n "== $lib rep $rep train step: "; python tools/exp_trainstep_graph.py 3000 2>&1 | tail -1
  echo -n "   $lib sm16: "; python tools/exp_sm16_ms.py 4096 64 300 2>&1 | tail -1
  python tools/exp_fit.py 2>&1 | grep -E "cycling|fit epoch   " | head -2
done
done
cp /tmp/lib_orig.so recommenders_amd/libtfrs_hip.so
