#!/bin/bash
cd /tmp; export TMPDIR=/tmp
for m in dlrm dcn; do
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rk_$m -o t -- python $GRAFT_REPO_ROOT/tools/exp_${m}_prof.py > $GRAFT_REPO_ROOT/gpurun_out/rk_$m.log 2>&1
done
cd $GRAFT_REPO_ROOT
for m in dlrm dcn; do echo "== $m"; python tools/step_kernels.py gpurun_out/rk_$m sort_init_kernel 3 40; done
