#!/bin/bash
cd /tmp; export TMPDIR=/tmp
for b in 1; do
BATCH=$b CALLS=6 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/st_trace_b$b -o t -- python $GRAFT_REPO_ROOT/tools/exp_streaming_prof.py > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,re,glob
for b in (1,):
  f=glob.glob('gpurun_out/st_trace_b%d/**/*kernel_trace.csv'%b,recursive=True)[0]
  rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
  marks=[i for i,r in enumerate(rows) if 'nonfinite_flag' in r['Kernel_Name']]
  lo,hi=marks[-2]+1,marks[-1]+1
  t0=int(rows[lo]['Start_Timestamp'])
  for r in rows[lo:hi]:
    n=re.sub(r'\(.*','',r['Kernel_Name']).replace('void ','')[:50]
    print("%8.1f %7.1f  %-50s wgs %s"%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,n,int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X']))))
PY
