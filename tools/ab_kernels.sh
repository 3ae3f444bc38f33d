# Same-box per-kernel comparison of the libraries under ab/ on the headline batch: for each ab/lib_*.so (twice,
# alternating) the bench line's step time and the rocprofv3 --kernel-trace --stats averages of its kernels.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
cp recommenders_amd/libtfrs_hip.so /tmp/lib_orig.so
for rep in $(seq 1 ${REPS:-2}); do
  for lib in ab/lib_*.so; do
    cp $lib recommenders_amd/libtfrs_hip.so
    echo "== $lib (rep $rep)"
    out=/tmp/abk_$(basename $lib .so)_$rep
    rm -rf $out
    rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python bench.py --no-cpu-baseline --no-train-step --no-gather --no-scale-workload --no-robustness --no-streaming --no-config-legs --steps 30 --warmup 5 2>/dev/null | grep '^{' | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step_ms', round(d['ms_per_step'],4), 'filter_ms', round(d['roofline']['avg_launch_ms'],4))"
    f=$(find $out -name '*kernel_stats.csv' | head -1)
    python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:6]:
  print("  %-52s calls %5s avg_us %9.1f" % (r["Name"][:52], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
done
cp /tmp/lib_orig.so recommenders_amd/libtfrs_hip.so
