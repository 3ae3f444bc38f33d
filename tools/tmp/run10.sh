ROOT=$PWD
for w in ${WHICH:-dcn}; do
OUT=$ROOT/gpurun_out/${w}_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- python $ROOT/tools/exp_${w}_prof.py > "$OUT/log.txt" 2>&1
cd $ROOT
echo "== $w"; grep "step ms" $OUT/log.txt
python tools/step_kernels.py $OUT sort_init_kernel 3 32
done
