"""Cross-compiles kernel sources for gfx950 (no GPU needed) and lists, per kernel, how its global loads are
awaited: `loads`, `stores`, the number of `s_waitcnt vmcnt(0)` and of waits that leave loads in flight, and the
largest count left in flight.  A kernel whose source batches loads but whose ISA shows only vmcnt(0) has its
loads in branches (`if (i < n) v = p[i]`): gfx950 counts a wave's loads with one in-order counter, the compiler
cannot count loads issued under a condition and waits for every one ("Three code shapes").

  python tools/isa_audit.py                      # every source of recommenders_amd/csrc
  python tools/isa_audit.py gemm16.hip embedding.hip --min-loads 8
"""
import argparse, os, re, subprocess, sys, tempfile
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from recommenders_amd.csrc import build as csrc_build

ap = argparse.ArgumentParser()
ap.add_argument("sources", nargs="*")
ap.add_argument("--min-loads", type=int, default=6)
args = ap.parse_args()
csrc = os.path.dirname(csrc_build.__file__)
sources = args.sources or [s for s in csrc_build.SOURCES if s.endswith(".hip")]
tmp = tempfile.mkdtemp(prefix="tfrs_isa_")
for src in sources:
  out = os.path.join(tmp, os.path.basename(src) + ".s")
  cmd = [csrc_build.hipcc(), f"--offload-arch={csrc_build.ARCH}", "-O3", "-std=c++17",
         *csrc_build.EXTRA_FLAGS.get(src, []), "-S", "--cuda-device-only", "-o", out, os.path.join(csrc, src)]
  subprocess.run(cmd, check=True, capture_output=True, cwd=csrc)
  name, loads, stores, waits = None, 0, 0, Counter()
  for line in open(out):
    t = line.strip()
    if line.startswith("_Z") and ":" in line.split()[0]:
      name, loads, stores, waits = line.split(":")[0], 0, 0, Counter()
    if name is None:
      continue
    if t.startswith(("global_load", "buffer_load")):
      loads += 1
    elif t.startswith(("global_store", "buffer_store")):
      stores += 1
    elif t.startswith("s_waitcnt") and "vmcnt" in t:
      waits[int(re.search(r"vmcnt\((\d+)\)", t).group(1))] += 1
    elif t.startswith(".amdhsa_kernel"):
      if loads >= args.min_loads:
        pretty = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        nz = sum(v for k, v in waits.items() if k > 0)
        flag = "  <-- every wait drains the counter" if nz == 0 and waits.get(0, 0) > 2 else ""
        print(f"{src:18s} {pretty[:72]:72s} loads={loads:4d} stores={stores:4d} vmcnt(0)={waits.get(0, 0):3d} "
              f"vmcnt(>0)={nz:3d} max_in_flight={max(waits) if waits else 0:2d}{flag}")
      name = None
