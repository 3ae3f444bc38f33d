"""Per-kernel means of the counters in rocprofv3 *_counter_collection.csv files under a dir."""
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
  for r in csv.DictReader(open(f)):
    name = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"]))[:48]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(acc):
  if pat and pat not in k: continue
  print(k)
  for cn in sorted(acc[k]):
    v = acc[k][cn]
    print("   %-28s %14.0f  (n=%d)" % (cn, sum(v) / len(v), len(v)))
