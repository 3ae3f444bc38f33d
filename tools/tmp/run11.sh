ROOT=$PWD
OUT=$ROOT/gpurun_out/head_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- python $ROOT/tools/exp_headline_prof.py > "$OUT/log.txt" 2>&1
cd $ROOT
python tools/trace_overlap.py $OUT 30
OUT=$ROOT/gpurun_out/st8k_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp
BATCH=8192 CALLS=3 timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- python $ROOT/tools/exp_streaming_prof.py > "$OUT/log.txt" 2>&1
cd $ROOT
python tools/trace_overlap.py $OUT 45
