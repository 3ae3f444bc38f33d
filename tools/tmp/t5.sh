#!/bin/bash
cd /tmp; export TMPDIR=/tmp
for b in 1 128; do
BATCH=$b CALLS=6 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/st_trace_b$b -o t -- python $GRAFT_REPO_ROOT/tools/exp_streaming_prof.py > $GRAFT_REPO_ROOT/gpurun_out/st_prof_$b.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,re,glob
for b in (1,128):
  f=glob.glob('gpurun_out/st_trace_b%d/**/*kernel_trace.csv'%b,recursive=True)[0]
  rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
  # last call: find the last occurrence of the first kernel name of a call: use 'query_kappa' or 'nonfinite'
  marks=[i for i,r in enumerate(rows) if 'note_nonfinite' in r['Kernel_Name'] or 'nonfinite_flag' in r['Kernel_Name']]
  lo=marks[-1]
  t0=int(rows[lo]['Start_Timestamp'])
  print("== BATCH",b, "kernels in last call:", len(rows)-lo)
  for r in rows[lo:]:
    n=re.sub(r'\(.*','',r['Kernel_Name']).replace('void ','')[:56]
    print("%8.1f %8.1f  %-56s grid %s"%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-t0)/1e3,n,int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X']))))
PY
tail -5 gpurun_out/st_prof_1.log
