"""Why did bench.py's wall time per step (1.93 ms) exceed its per-step HIP-event time (1.16 ms)?
Replays the timed loop of bench.py with and without the per-step torch events and the library's
per-launch profile events (development tool)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd import _lib
from recommenders_amd.layers import factorized_top_k as ftk
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(42)
corpus = torch.randn((1_000_000, 64), generator=g, device=dev) / 8.0
queries = torch.randn((8192, 64), generator=g, device=dev) / 8.0
index = ftk.BruteForce(k=100).index(corpus)
lib = _lib.load()
for _ in range(10):
  index(queries)
torch.cuda.synchronize()
for prof in (0, 1, 0, 1):
  for use_events in (False, True):
    lib.tfrs_profile_enable(prof)
    steps = 50
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in ev:
      if use_events: a.record()
      index(queries)
      if use_events: b.record()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms = sorted(a.elapsed_time(b) for a, b in ev) if use_events else [0.0]
    lib.tfrs_profile_read(None, None, None)
    lib.tfrs_profile_enable(0)
    print(f"profile={prof} events={use_events}: wall {el / steps * 1e3:.3f} ms/step, issue {t_issue / steps * 1e3:.3f}, "
          f"event median {ms[len(ms) // 2]:.3f} max {ms[-1]:.3f}", flush=True)
