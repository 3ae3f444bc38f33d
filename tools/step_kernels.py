"""Per-kernel totals of the LAST `nsteps` train steps of a rocprofv3 --kernel-trace CSV, with workgroup counts: finds the
launches that leave the chip idle (few workgroups, long duration).  python tools/step_kernels.py DIR ANCHOR [nsteps]
ANCHOR = substring of a kernel that runs exactly once per step (e.g. sort_init_kernel)."""
import csv, sys, glob, re, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
anchor = sys.argv[2]
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
lo, hi = marks[-nsteps - 1], marks[-1]
sel = rows[lo:hi]
span = (int(sel[-1]['End_Timestamp']) - int(sel[0]['Start_Timestamp'])) / 1e6 / nsteps
agg = collections.defaultdict(lambda: [0.0, 0, 0, 1 << 60])
for r in sel:
  name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')[:64]
  wgs = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1) // max(1, int(r['Workgroup_Size_X']) * int(r.get('Workgroup_Size_Y', 1) or 1))
  a = agg[name]
  a[0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
  a[1] += 1
  a[2] = max(a[2], wgs)
  a[3] = min(a[3], wgs)
busy = sum(a[0] for a in agg.values()) / nsteps
print("step span %.3f ms, kernel time %.3f ms per step" % (span, busy / 1e3))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[4]) if len(sys.argv) > 4 else 45]:
  print("%-66s n/step %5.1f  us/step %9.1f  wgs %d..%d" % (k, a[1] / nsteps, a[0] / nsteps, a[3], a[2]))
