#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/exp_sm16_ms.py 4096 64 300 2>&1 | tail -1
for i in 1 2; do echo -n "step: "; python tools/exp_trainstep_graph.py 3000 2>&1 | tail -1; done
echo -n "step (no metrics): "; TFRS_EXP_METRICS=0 python tools/exp_trainstep_graph.py 3000 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -x -q -k "softmax or train_step or retrieval or fit or scatter or adagrad or embedding or rowscan or quickstart" 2>&1 | tail -4
cd /tmp; rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ts_trace2 -o t -- python $GRAFT_REPO_ROOT/tools/exp_trainstep_graph.py 50 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/step_kernels.py gpurun_out/ts_trace2 sm16_prep 10 30
