# same-box A/B of two builds of the library on the configs[1] batch: the in-tree build against
# recommenders_amd/libtfrs_hip_head.so (a build of another revision, made by hand).  Evidence tool.
set -e
cd "$(dirname "$0")/.."
cp recommenders_amd/libtfrs_hip.so /tmp/lib_new.so
cp recommenders_amd/libtfrs_hip_head.so /tmp/lib_head.so
for rep in 1 2 3; do
  for v in head new; do
    cp /tmp/lib_$v.so recommenders_amd/libtfrs_hip.so
    echo "== $v (rep $rep)"
    python tools/exp_filter_ms.py 2>&1 | tail -1
  done
done
cp /tmp/lib_new.so recommenders_amd/libtfrs_hip.so
