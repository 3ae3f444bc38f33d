import csv,sys,glob
for v in (0,1,2,4):
  for b in (1,128):
    f=glob.glob(f'gpurun_out/l16_{v}_{b}/**/*kernel_trace.csv',recursive=True)
    if not f: continue
    rows=list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if 'list_topk16' in r['Kernel_Name']]
    # 4 per call; show last call's
    print(v,b,[round(x,1) for x in d[-4:]], 'first-of-call:',[round(x,1) for x in d[0::4]])
