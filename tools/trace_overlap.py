"""Prints the last N kernels of a rocprofv3 --kernel-trace CSV with start times relative to the first one printed
(us), durations and queue ids: shows what overlaps and where the gaps are.  python tools/trace_overlap.py DIR [N]"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 48
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
tail = rows[-n:]
t0 = int(tail[0]['Start_Timestamp'])
prev_end = t0
for r in tail:
  s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
  print("%-56s start %8.1f dur %7.1f gap %6.1f q %s" % (r['Kernel_Name'].replace('void ', '')[:56], (s - t0) / 1e3, (e - s) / 1e3,
                                                         (s - prev_end) / 1e3, r.get('Queue_Id', '?')))
  prev_end = max(prev_end, e)
