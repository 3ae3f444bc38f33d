"""Headline benchmark: brute-force top-100 retrieval throughput (queries/sec).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): synthetic 1M-item x dim-64 corpus, batch of 8192
queries, exact top-100 (`BruteForce.call`, reference layers/factorized_top_k.py:586-607),
inputs resident in HBM before the timed region.  Returned scores and indices are those of
the float32 fma chain (bit-identical to the all-f32 path); the default path filters with
fp16 MFMA scores under a rigorous error bound and re-scores the survivors exactly
(DESIGN.md 4.1).  A "step" is one `BruteForce` call on the batch.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): the corpus is
row-sharded -- every rank owns a 1M-row shard (weak scaling: total corpus = N x 1M rows),
all ranks score the same 8192 queries against their shard, then all_gather the per-shard
top-100 (score, global row) lists over xGMI and merge them.  `value` counts
shard-queries: N * 8192 / step time.

Prints ONE JSON line on rank 0 with the contract fields plus `roofline` (the kernel the step
spends most time in: the fp16 filter pass, priced against the dense 16-bit MFMA peak with
its ALGORITHMIC flop 2*B*N*D), `cpu_baseline` and `secondary` (train steps/sec), N = 1 only.
"""

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

N_ROWS, DIM, BATCH, TOPK = 1_000_000, 64, 8192, 100
F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32-input MFMA, dense
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16/f16 MFMA, dense (no sparsity)


def synth(rows: int, seed: int, device) -> torch.Tensor:
  g = torch.Generator(device=device).manual_seed(seed)
  return torch.randn((rows, DIM), generator=g, device=device, dtype=torch.float32) / (DIM ** 0.5)


def hbm_traffic(kernel: str):
  """HBM bytes per launch of the dominant kernel, from the committed rocprofv3 PMC pass of this
  same command (tools/profile_bench.sh -> profiles/*_traffic.json; FETCH_SIZE with the gfx950
  x2 correction, reads only).  None when no profile of the current kernel is committed."""
  path = os.path.join(ROOT, "profiles", "latest_traffic.json")
  try:
    with open(path) as f:
      return json.load(f)[kernel]["fetch_bytes_corrected"]
  except (OSError, KeyError, ValueError):
    return None


def train_step_metric(dev) -> dict:
  """Second half of BASELINE.json's metric: train steps/sec of the in-batch-softmax two-tower
  step at the MovieLens-100K shapes of configs[0] (B=4096, D=64, 2k-row user/item tables):
  ``tfrs.Model.train_step`` of the quickstart two-tower model: embedding gather -> fused
  in-batch softmax loss (tasks/retrieval.py:172-210) -> backward -> sparse Adagrad on the
  looked-up rows (IndexedSlices semantics).  The arithmetic kernels are HIP; torch runs the
  autograd bookkeeping and the id sort."""
  import recommenders_amd as tfrs
  g = torch.Generator(device=dev).manual_seed(0)
  B, D, V = 4096, 64, 2000

  class TwoTower(tfrs.Model):          # the reference's quickstart model (README.md:58-82)
    def __init__(self):
      super().__init__()
      self.user_model = tfrs.layers.embedding.Embedding(V, D)
      self.item_model = tfrs.layers.embedding.Embedding(V, D)
      self.task = tfrs.tasks.Retrieval()

    def compute_loss(self, inputs, training=False):
      return self.task(self.user_model(inputs["user_id"]), self.item_model(inputs["movie_id"]),
                       compute_metrics=False)

  model = TwoTower()
  model.compile(optimizer=tfrs.optimizers.Adagrad(model.parameters(), learning_rate=0.5))
  batch = {"user_id": torch.randint(0, 943, (B,), generator=g, device=dev),
           "movie_id": torch.randint(0, 1682, (B,), generator=g, device=dev)}

  def step():
    model.train_step(batch)            # models/base.py:64-85

  for _ in range(5):
    step()
  torch.cuda.synchronize()
  iters = 100
  t0 = time.perf_counter()
  for _ in range(iters):
    step()
  torch.cuda.synchronize()
  dt_eager = (time.perf_counter() - t0) / iters
  # the same train_step captured once in a HIP graph and replayed (models/base.py
  # make_graphed_train_step): identical kernels and arithmetic, no per-launch host cost;
  # every replay includes the copy of the batch into the graph's static input buffers
  graphed = model.make_graphed_train_step(batch)
  for _ in range(5):
    graphed(batch)
  torch.cuda.synchronize()
  iters = 300
  t0 = time.perf_counter()
  for _ in range(iters):
    graphed(batch)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / iters
  return {"metric": "train steps/sec (in-batch softmax)", "value": 1.0 / dt, "unit": "steps/s",
          "ms_per_step": dt * 1e3, "dtype": "f32", "mode": "hipGraph replay of tfrs.Model.train_step",
          "eager_steps_per_s": 1.0 / dt_eager, "eager_ms_per_step": dt_eager * 1e3,
          "config": {"workload": "two-tower train step, MovieLens-100K shapes (BASELINE.json configs[0]): "
                                 "batch 4096, dim 64, 2k x 64 user + item tables, Adagrad lr 0.5, "
                                 "compute_metrics=False", "batch": B, "dim": D}}


def main() -> None:
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=20)
  ap.add_argument("--warmup", type=int, default=3)
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--cpu-budget", type=float, default=15.0)
  ap.add_argument("--no-train-step", action="store_true")
  args = ap.parse_args()

  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs a ROCm GPU: there is no CPU fallback for the product path")
  # Validation aid for boxes with ONE GPU (not used by the driver): TFRS_BENCH_ONE_GPU=1 runs the
  # N > 1 code path with every rank on cuda:0 and the single all-gather of the sharded path
  # staged through gloo/host, because RCCL refuses two ranks on one device.
  one_gpu = os.environ.get("TFRS_BENCH_ONE_GPU", "0") == "1"
  if one_gpu:
    local_rank = 0
  torch.cuda.set_device(local_rank)
  dev = torch.device("cuda", local_rank)
  if world > 1:
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if one_gpu:
      dist.init_process_group("gloo")
      real_gather = dist.all_gather_into_tensor

      def staged_gather(out, inp, group=None):
        host = torch.empty(out.shape, dtype=out.dtype)
        real_gather(host, inp.cpu(), group=group)
        out.copy_(host)

      dist.all_gather_into_tensor = staged_gather
    else:
      dist.init_process_group("nccl", device_id=dev)

  from recommenders_amd import _lib
  from recommenders_amd.layers import factorized_top_k as ftk

  corpus = synth(N_ROWS, seed=42 + rank, device=dev)       # this rank's shard
  queries = synth(BATCH, seed=7, device=dev)               # same queries on every rank
  if world > 1:
    index = ftk.ShardedBruteForce(k=TOPK).index(corpus, base_row=rank * N_ROWS)
  else:
    index = ftk.BruteForce(k=TOPK).index(corpus)
  lib = _lib.load()

  def step():
    return index(queries)

  for _ in range(args.warmup):
    step()
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  lib.tfrs_profile_enable(1)
  t0 = time.perf_counter()
  for _ in range(args.steps):
    out = step()
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  elapsed = time.perf_counter() - t0
  # per-launch HIP-event timings of the scan kernels (0: exact f32 scan, 1: fp16 filter pass over
  # all rows, 2: fp16 threshold pass over the sampled stages)
  kinds = {}
  for kind in (0, 1, 2):
    ms_k, n_k, fl_k = ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
    lib.tfrs_profile_read_kind(kind, ctypes.byref(ms_k), ctypes.byref(n_k), ctypes.byref(fl_k))
    kinds[kind] = (ms_k.value, n_k.value, fl_k.value)
  lib.tfrs_profile_read(None, None, None)   # reset
  lib.tfrs_profile_enable(0)
  dom = max(kinds, key=lambda kk: kinds[kk][0])          # the kernel the step spends most time in
  scan_ms, launches, flop = (ctypes.c_double(kinds[dom][0]), ctypes.c_int(kinds[dom][1]),
                             ctypes.c_double(kinds[dom][2]))

  t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
  if world > 1:
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
  elapsed = float(t.item())
  ms_per_step = elapsed / args.steps * 1e3
  value = world * BATCH / (elapsed / args.steps)

  if rank == 0:
    achieved = flop.value / (scan_ms.value * 1e-3) / 1e12 if scan_ms.value > 0 else 0.0
    peak = F16_MFMA_PEAK_TFLOPS if dom >= 1 else F32_MFMA_PEAK_TFLOPS
    result = {
        "metric": "queries/sec brute-force top-100",
        "value": value,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",   # returned scores: exact f32 fma chains (fp16 is used only to filter)
        "data": "synthetic",
        "config": {
            "workload": "BruteForce top-100, 1M-item x dim-64 corpus per GPU, batch 8192 "
                        "(BASELINE.json configs[1])",
            "corpus_rows_per_gpu": N_ROWS, "corpus_rows_total": N_ROWS * world, "dim": DIM,
            "batch": BATCH, "k": TOPK,
            "parallelism": ("single GPU" if world == 1 else
                            f"corpus row-sharded x{world}, RCCL all_gather of per-shard top-K + merge"),
        },
        "roofline": {
            "kernel": ("tfrs::scan16_kernel<64, FILTER> (fp16 MFMA prefilter scores of all rows + "
                       "fused top-K filter; survivors re-scored exactly in f32)" if dom >= 1 else
                       "tfrs::scan_kernel<64> (f32 MFMA scores + fused top-K filter)"),
            "bound": "mfma",
            "achieved": achieved,
            "peak": peak,
            "unit": "TFLOP/s",
            "frac": achieved / peak,
            "traffic": hbm_traffic("tfrs::scan16_kernel<64, 0>" if dom >= 1 else "tfrs::scan_kernel<64, false, true>"),
            "launches": launches.value,
            "avg_launch_ms": scan_ms.value / max(launches.value, 1),
            "algorithmic_flop_per_launch": flop.value / max(launches.value, 1),
            "algorithmic_flop_per_step": 2.0 * BATCH * N_ROWS * DIM,
            "scan_ms_per_step": scan_ms.value / args.steps,
            "f32_scan_ms_per_step": kinds[0][0] / args.steps,
            "f16_filter_pass_ms_per_step": kinds[1][0] / args.steps,
            "f16_threshold_pass_ms_per_step": kinds[2][0] / args.steps,
            "peak_note": "2.5 PFLOP/s = dense fp16/bf16 MFMA spec; tools/ubench/mfma_rate.hip "
                         "sustains 1.57 PFLOP/s on this chip with random operands (power-limited clock)",
        },
    }
    if world == 1 and not args.no_cpu_baseline:
      from oracle import cpu_path  # checker-side code: only this baseline leg uses it
      c_host = corpus.cpu().numpy()
      q_host = queries.cpu().numpy()
      base = cpu_path.time_brute_force(c_host, q_host, TOPK, budget_s=args.cpu_budget)
      # sanity: the timed CPU path agrees with the GPU result on its first block
      v, i = cpu_path.brute_force_topk(q_host[:64], c_host, TOPK)
      agree = float((i == out[1][:64].cpu().numpy()).mean())
      result["cpu_baseline"] = {
          "value": base["value"], "unit": "queries/s", "cores": base["threads"],
          "kind": "port",
          "sample": "%d queries x full 1M x 64 corpus in blocks of 256, %.1f s; torch-CPU "
                    "sgemm + topk restatement of BruteForce.call (not TensorFlow); index "
                    "agreement with the GPU on 64 queries: %.4f"
                    % (base["queries"], base["seconds"], agree),
      }
    if world == 1 and not args.no_train_step:
      result["secondary"] = train_step_metric(dev)
    print(json.dumps(result), flush=True)
  if world > 1:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
