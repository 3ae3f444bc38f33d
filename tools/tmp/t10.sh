#!/bin/bash
cd /tmp; export TMPDIR=/tmp
for cfg in "1 1000000 64" "64 1000000 64" "1 12500000 128"; do
set -- $cfg
BATCH=$1 ROWS=$2 DIM=$3 CALLS=6 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/bf_small -o t -- python $GRAFT_REPO_ROOT/tools/exp_bruteforce_small.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - "$cfg" <<'PY'
import csv,re,glob,sys
f=glob.glob('gpurun_out/bf_small/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
marks=[i for i,r in enumerate(rows) if 'query_kappa' in r['Kernel_Name']]
lo,hi=marks[-2],marks[-1]
t0=int(rows[lo]['Start_Timestamp'])
print("== BATCH ROWS DIM =",sys.argv[1])
for r in rows[lo:hi]:
  n=re.sub(r'\(.*','',r['Kernel_Name']).replace('void ','')[:50]
  print("%8.1f %7.1f  %-50s wgs %s"%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,n,int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X']))))
PY
rm -rf gpurun_out/bf_small; cd /tmp
done
