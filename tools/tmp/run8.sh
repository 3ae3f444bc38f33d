ROOT=$PWD
OUT=$ROOT/gpurun_out/dlrm_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python $ROOT/tools/exp_dlrm_prof.py > "$OUT/log.txt" 2>&1
cd $ROOT
grep "step ms" $OUT/log.txt
python tools/print_kernel_stats.py $(find "$OUT" -name "*kernel_stats.csv" | head -1) 40
