import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find last nonfinite_flag start
idx=[i for i,r in enumerate(rows) if 'nonfinite_flag' in r['Kernel_Name']]
tail=rows[idx[-2]:idx[-1]]
t0=int(tail[0]['Start_Timestamp'])
for r in tail:
  s=int(r['Start_Timestamp']);e=int(r['End_Timestamp'])
  print(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} wg={int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X'])):6d} {r['Kernel_Name'][:70]}")
