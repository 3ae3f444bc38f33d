# Builds libtfrs_hip variants that differ in ONE compile-time define of ONE source file and times the fp16 filter pass
# of each on the same box (run under gpurun).  Usage:
#   tools/ab_variants.sh build topk_scan16.hip TFRS_SCAN16_ABLATE 0 1 2 ...   (here, CPU: cross-compiles ab/lib_<v>.so)
#   tools/ab_variants.sh run [exp_filter_ms.py args]                           (on the GPU box)
set -e
cd "$(dirname "$0")/.."
OBJ=recommenders_amd/csrc/_obj
if [ "$1" = build ]; then
  src=$2; def=$3; shift 3
  mkdir -p ab
  extra=""
  { [ "$src" = topk_scan16.hip ] || [ "$src" = topk_raw.hip ]; } && extra="-fno-honor-nans"
  [ "$src" = softmax16.hip ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
  base=$(basename $src .hip)
  for v in "$@"; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra -DTFRS_ALLOW_ABLATION -D$def=$v -x hip -c recommenders_amd/csrc/$src -o ab/${base}_$v.o &&
      objs=$(ls $OBJ/*.o | grep -v "/${base}.o") &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/lib_$v.so $objs ab/${base}_$v.o && echo built ab/lib_$v.so ) &
  done
  wait
  exit 0
fi
shift
cp recommenders_amd/libtfrs_hip.so /tmp/lib_orig.so
for rep in $(seq 1 ${REPS:-2}); do
  for lib in ab/lib_*.so; do
    cp $lib recommenders_amd/libtfrs_hip.so
    echo "== $lib (rep $rep)"
    python ${EXP:-tools/exp_filter_ms.py} "$@" 2>&1 | grep "^{" | tail -${TAILN:-1}
  done
done
cp /tmp/lib_orig.so recommenders_amd/libtfrs_hip.so
