# Same-box run of the TFRS_SCAN16_PEEL builds (ab/lib_<v>.so from tools/ab_variants.sh build): per library the
# filter-pass time, what differs from the f32 path on the 1 M x 64 batch (tools/exp_peel_dbg.py) and the whole-batch
# stress test (f16 == f32 on random batches, every shape).
cd "$(dirname "$0")/.."
cp recommenders_amd/libtfrs_hip.so /tmp/lib_orig.so
for lib in ab/lib_*.so; do
  cp $lib recommenders_amd/libtfrs_hip.so
  echo "== $lib"
  for shape in 8x2 16x2s2; do
    python tools/exp_filter_ms.py TFRS_SCAN16_SHAPE=$shape 2>&1 | grep "^{" | sed -n 2p
  done
  python tools/exp_peel_dbg.py 8x2 16x2 4x4 8x4 2>&1 | grep "^{" | cut -c1-420
  [ -n "$STRESS" ] && timeout 600 python -m pytest tests/test_fuzz_gpu.py -q -x -k "keep_every_survivor" -p no:cacheprovider 2>&1 | tail -1
done
cp /tmp/lib_orig.so recommenders_amd/libtfrs_hip.so
