"""Condenses a tools/profile_bench.sh output directory (rocprofv3 CSVs under gpurun_out/)
into one small markdown file under profiles/ that can be committed and cited.

    python tools/summarize_profile.py gpurun_out/r01a/prof profiles/r01_bench_topk.md [bench.json]

Only kernels of this library (namespace tfrs) are listed by name; everything else is lumped
into "other".  PMC values are averaged per launch.  FETCH_SIZE/WRITE_SIZE are reported in
rocprofv3's unit (KiB) and FETCH_SIZE is additionally shown doubled, the gfx950 correction
for wide coalesced reads given in MI355X_MICROARCH.md (HBM section).
"""

import collections
import csv
import json
import os
import re
import sys


def short(name: str) -> str:
  name = re.sub(r"^void ", "", name)
  m = re.match(r"(tfrs::[A-Za-z0-9_]+(<[^>]*>)?)", name)
  return m.group(1) if m else "other"


def main() -> None:
  src, dst = sys.argv[1], sys.argv[2]
  bench = sys.argv[3] if len(sys.argv) > 3 else None
  out = ["# rocprofv3 summary: %s" % os.path.basename(dst), "",
         "Source directory (scratch, not committed): `%s`; collected by `tools/profile_bench.sh`" % src,
         "(kernel trace + stats in one run; each PMC group in its own run with --kernel-trace only).", ""]
  if bench and os.path.exists(bench):
    out += ["## bench.py line of the same build", "", "```json", open(bench).read().strip(), "```", ""]

  stats = os.path.join(src, "trace", "bench_kernel_stats.csv")
  if os.path.exists(stats):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(stats)):
      k = short(r["Name"])
      a = agg.setdefault(k, [0, 0.0, float("inf"), 0.0])
      a[0] += int(r["Calls"])
      a[1] += float(r["TotalDurationNs"])
      a[2] = min(a[2], float(r["MinNs"]))
      a[3] = max(a[3], float(r["MaxNs"]))
    total = sum(a[1] for a in agg.values()) or 1.0
    out += ["## kernel stats (`rocprofv3 --kernel-trace --stats`, bench.py default protocol: 10 warm-up + 50 timed steps)", "",
            "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for k, a in agg.items():
      out.append("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.2f |" %
                 (k, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100 * a[1] / total))
    out.append("")

  trace = os.path.join(src, "trace", "bench_kernel_trace.csv")
  if os.path.exists(trace):
    res = collections.OrderedDict()
    for r in csv.DictReader(open(trace)):
      k = short(r["Kernel_Name"])
      if k == "other" or k in res:
        continue
      res[k] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"),
                r.get("LDS_Block_Size"), r.get("Workgroup_Size_X"), r.get("Grid_Size_X"))
    out += ["## launch resources (first launch of each kernel)", "",
            "| kernel | VGPR | AGPR | SGPR | LDS B | wg | grid |", "|---|---|---|---|---|---|---|"]
    for k, v in res.items():
      out.append("| `%s` | %s | %s | %s | %s | %s | %s |" % ((k,) + v))
    out.append("")

  pmc = collections.OrderedDict()
  for sub in sorted(os.listdir(src)):
    p = os.path.join(src, sub, "bench_counter_collection.csv")
    if not sub.startswith("pmc") or not os.path.exists(p):
      continue
    for r in csv.DictReader(open(p)):
      k = short(r["Kernel_Name"])
      if k == "other":
        continue
      d = pmc.setdefault(k, collections.OrderedDict())
      v = d.setdefault(r["Counter_Name"], [0, 0.0])
      v[0] += 1
      v[1] += float(r["Counter_Value"])
  if pmc:
    out += ["## PMC counters, average per launch (separate runs, bench.py --steps 3 --warmup 1)", ""]
    for k, d in pmc.items():
      out.append("### `%s`" % k)
      out.append("")
      out.append("| counter | launches | avg per launch |")
      out.append("|---|---|---|")
      for c, (n, s) in d.items():
        out.append("| %s | %d | %.6g |" % (c, n, s / n))
      g = {c: s / n for c, (n, s) in d.items()}
      notes = []
      if "SQ_VALU_MFMA_BUSY_CYCLES" in g and g.get("GRBM_GUI_ACTIVE"):
        # BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
        util = (g["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (g["GRBM_GUI_ACTIVE"] / 8.0)
        notes.append("MFMA pipe utilisation = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / "
                     "(GRBM_GUI_ACTIVE / 8 XCDs) = **%.1f %%**" % (100 * util))
      if "FETCH_SIZE" in g:
        notes.append("FETCH_SIZE = %.1f MiB per launch as reported; x2 gfx950 wide-read correction = "
                     "**%.1f MiB**" % (g["FETCH_SIZE"] / 1024, 2 * g["FETCH_SIZE"] / 1024))
      if "WRITE_SIZE" in g:
        notes.append("WRITE_SIZE = %.1f MiB per launch (uncalibrated unit: KiB)" % (g["WRITE_SIZE"] / 1024))
      if "SQ_LDS_BANK_CONFLICT" in g and g.get("SQ_LDS_IDX_ACTIVE"):
        notes.append("LDS bank-conflict cycles / LDS active cycles = %.2f %%" %
                     (100 * g["SQ_LDS_BANK_CONFLICT"] / g["SQ_LDS_IDX_ACTIVE"]))
      if all(c in g for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")):
        t = g["SQ_WAIT_ANY"] + g["SQ_WAIT_INST_ANY"] + g["SQ_ACTIVE_INST_ANY"]
        notes.append("wave-cycle split: parked (s_waitcnt/barrier) %.0f %% / issue-stalled (MFMA pipe, RAW) "
                     "%.0f %% / issuing %.0f %%" % (100 * g["SQ_WAIT_ANY"] / t, 100 * g["SQ_WAIT_INST_ANY"] / t,
                                                    100 * g["SQ_ACTIVE_INST_ANY"] / t))
      out.append("")
      out += ["* " + n for n in notes]
      out.append("")
  # machine-readable HBM traffic per launch for bench.py's roofline.traffic
  traffic = {}
  for k, d in pmc.items():
    g = {c: s_ / n for c, (n, s_) in d.items()}
    if "FETCH_SIZE" in g:
      traffic[k] = {"fetch_bytes_corrected": 2 * g["FETCH_SIZE"] * 1024,
                    "write_bytes_uncalibrated": g.get("WRITE_SIZE", 0.0) * 1024,
                    "note": "rocprofv3 --pmc FETCH_SIZE (KiB) x2 per MI355X_MICROARCH.md; separate pass"}
  if traffic:
    with open(os.path.splitext(dst)[0] + "_traffic.json", "w") as f:
      json.dump(traffic, f, indent=1)
  os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
  with open(dst, "w") as f:
    f.write("\n".join(out) + "\n")
  print("wrote", dst)


if __name__ == "__main__":
  main()
