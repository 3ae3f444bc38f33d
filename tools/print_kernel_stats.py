"""Prints the top rows of a rocprofv3 *_kernel_stats.csv with shortened kernel names."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in rows[:n]:
  name = re.sub(r"^void ", "", r["Name"])
  name = re.sub(r"\(.*", "", name)[:72]
  print("%-72s calls=%5s avg_us=%8.1f pct=%s" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
