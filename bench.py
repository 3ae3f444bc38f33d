"""Headline benchmark: brute-force top-100 retrieval throughput (queries/sec).

    python bench.py --gpus N --steps K --warmup W

N = 1 (BASELINE.json configs[1], the configuration the metric is quoted on): synthetic
1M-item x dim-64 corpus, batch of 8192 queries, exact top-100 (`BruteForce.call`, reference
layers/factorized_top_k.py:586-607), inputs resident in HBM before the timed region.  Returned
scores and indices are those of the float32 fma chain (bit-identical to the all-f32 path); the
default path FILTERS with fp16 MFMA scores under a rigorous error bound and re-scores the
survivors exactly.  A "step" is one `BruteForce` call on the batch.  The same
line also carries `scale_workload`: the single-GPU rate on the 100M x 64 corpus, the N = 1 point
of the strong-scaling configuration below.

N > 1 (north_star: ">= 6x top-K throughput at 8 GPUs vs 1 GPU on a 100M x 64 corpus"):
STRONG scaling -- the 100M x 64 corpus is row-sharded over the N ranks (one process per GPU,
RCCL), every rank scores the same 8192 queries against its 100M/N rows, the per-shard
(score, global row)[8192, 100] lists are exchanged by ONE all_gather over xGMI and merged
(`ShardedBruteForce`); collective + merge are inside the timed region; `value` = 8192 / t.
Started either by the driver's `python -m torch.distributed.run ... bench.py --gpus N` or as
plain `python bench.py --gpus N`, which re-executes itself under torch.distributed.run; it
fails loudly when fewer than N devices or ranks come up.  `--workload scale100m` forces the
100M x 64 workload at N = 1; TFRS_BENCH_ROWS shrinks it for dry runs.

Timing: W untimed warm-up steps, then K steps bracketed by barrier + synchronize on both
sides (max over ranks) -> `ms_per_step`, `value`; every step is additionally bracketed by HIP
events on the launch stream -> `step_ms_median/p10/p90`.  While the K steps run the library
brackets each scan launch with HIP events on its stream (`tfrs_profile_enable`; two event
records per launch inside the timed region -- conservative for `value`): `roofline` is the
kernel the step spends most time in, priced with its ALGORITHMIC flop 2*B*N*D against the
dense 16-bit MFMA peak.  N = 1 only: `cpu_baseline` (oracle restatement on the host cores),
`secondary` (train steps/sec of `Model.fit` with its own roofline + cpu_baseline), `gather` (embedding
gather GB/s at BASELINE configs[3] shapes), `streaming` (Streaming over a dataset of blocks, one GPU's
shard of configs[2]) and `config_legs` (bench_legs.py: fused Cross, DotInteraction, segment-sum, sparse
Adagrad and the DCN-v2 / DLRM-shard train steps of configs[3] / configs[4], each with roofline, parity
assert and cpu_baseline).  The CPU legs run after every GPU measurement.
"""

import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ROWS, DIM, BATCH, TOPK = 1_000_000, 64, 8192, 100
SCALE_ROWS = 100_000_000                 # north_star strong-scaling corpus (x DIM 64)
INGEST_BLOCK = 1_000_000                 # rows generated + packed per ingest step
F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32-input MFMA, dense
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16/f16 MFMA, dense (no sparsity)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured copy ceiling)
# the filter-pass instantiation a batch of >= 2 query tiles of dim <= 64 takes (csrc/topk_scan16.hip, round 5: 16 waves =
# two query tiles on one stage buffer, two stages per barrier period); rocprofv3 prints the same name
SCAN16F_KERNEL = "tfrs::scan16f_kernel<64, 16, 2, 2>"


def parse_args():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=50)
  ap.add_argument("--warmup", type=int, default=10)
  ap.add_argument("--workload", choices=("auto", "headline", "scale100m", "streaming128", "dlrm_embedding"),
                  default="auto")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--cpu-budget", type=float, default=12.0)
  ap.add_argument("--no-train-step", action="store_true")
  ap.add_argument("--no-gather", action="store_true")
  ap.add_argument("--no-scale-workload", action="store_true")
  ap.add_argument("--no-robustness", action="store_true")
  ap.add_argument("--no-streaming", action="store_true")
  ap.add_argument("--no-config-legs", action="store_true")
  ap.add_argument("--no-single-gpu-reference", action="store_true")
  return ap.parse_args()


def respawn_under_torchrun(args) -> None:
  """`python bench.py --gpus N` (N > 1) outside a launcher: start N ranks of this script."""
  port = 29500 + (os.getpid() % 2000)
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
         f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
         os.path.abspath(__file__)] + sys.argv[1:]
  env = dict(os.environ)
  env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  env.setdefault("OMP_NUM_THREADS", "8")
  raise SystemExit(subprocess.call(cmd, env=env))


args = parse_args()
if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
  respawn_under_torchrun(args)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def synth(rows: int, seed: int, device) -> "torch.Tensor":
  g = torch.Generator(device=device).manual_seed(seed)
  return torch.randn((rows, DIM), generator=g, device=device, dtype=torch.float32) / (DIM ** 0.5)


def corpus_blocks(row_begin: int, row_end: int, device):
  """Rows [row_begin, row_end) of the synthetic corpus as a re-iterable of <= 1M-row blocks;
  block b of the GLOBAL corpus is always generated from seed 1000 + b, so a shard's rows are
  the same rows whatever the number of ranks."""
  class Blocks:
    def __iter__(self):
      lo = row_begin
      while lo < row_end:
        b = lo // INGEST_BLOCK
        blk = synth(INGEST_BLOCK, 1000 + b, device)
        hi = min(row_end, (b + 1) * INGEST_BLOCK)
        yield blk[lo - b * INGEST_BLOCK: hi - b * INGEST_BLOCK]
        lo = hi
  return Blocks()


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


def gather_traffic() -> dict:
  """HBM bytes per launch of the configs[3] gather from the committed PMC passes, WITH provenance
  (profiles/gather_traffic.json, written by tools/run_gather_evidence.sh); never a constant in this
  file: `traffic` is null when the profile of the current kernel is missing."""
  path = os.path.join(ROOT, "profiles", "gather_traffic.json")
  try:
    with open(path) as f:
      rec = json.load(f)["tfrs::gather_kernel"]["c3"]
    return {"traffic": rec["fetch_bytes_corrected"] + rec["write_bytes"],
            "traffic_source": "profiles/gather_traffic.json: %s (FETCH_SIZE with the gfx950 x2 correction + "
                              "WRITE_SIZE, separate --pmc passes; not measured in this run)" % rec["source"]}
  except (OSError, KeyError, ValueError):
    return {"traffic": None}


def softmax_chain_traffic() -> dict:
  """HBM bytes per step of the in-batch softmax chain (tfrs::sm16_* kernels) of the README train step, from the
  committed PMC passes (profiles/trainstep_traffic.json, tools/run_trainstep_traffic.sh: FETCH_SIZE with the gfx950 x2
  correction + WRITE_SIZE, separate passes, graph-replayed step); null when the file is missing."""
  try:
    with open(os.path.join(ROOT, "profiles", "trainstep_traffic.json")) as f:
      rec = json.load(f)
    ks = {k: v for k, v in rec["kernels"].items() if k.startswith("tfrs::sm16_")}
    if not ks:
      return {"traffic": None}
    total = sum(v["fetch_bytes_corrected"] + v["write_bytes"] for v in ks.values())
    return {"traffic": total, "traffic_source": "profiles/trainstep_traffic.json (%s; sum over %s; not measured in this run)"
                                                % (rec["source"], ", ".join(sorted(ks)))}
  except (OSError, KeyError, ValueError):
    return {"traffic": None}


def emit(result: dict) -> None:
  """Rank 0's output: the full object to gpurun_out/bench_detail.json and to a PREFIXED stdout line (not a JSON
  line), then the compact line -- the LAST stdout line, a few KB (bench_compact.py; round 5's 23.5 KB line could not
  be parsed by the driver)."""
  import bench_compact
  detail = os.path.join(ROOT, "gpurun_out", "bench_detail.json")
  try:
    os.makedirs(os.path.dirname(detail), exist_ok=True)
    with open(detail, "w") as f:
      json.dump(result, f, indent=1)
    result["detail_file"] = "gpurun_out/bench_detail.json"
  except OSError as e:
    print(f"bench.py: could not write {detail}: {e}", file=sys.stderr, flush=True)
  print("bench_detail: " + json.dumps(result), flush=True)
  sys.stdout.flush()
  print(bench_compact.dumps(result), flush=True)


def percentiles(xs):
  xs = sorted(xs)
  n = len(xs)
  pick = lambda p: xs[min(n - 1, max(0, int(round(p * (n - 1)))))]
  return {"median": pick(0.5), "p10": pick(0.1), "p90": pick(0.9)}


def event_times_ms(fn, iters: int, warmup: int):
  """Per-iteration GPU times of fn() from HIP events on the current stream."""
  for _ in range(warmup):
    fn()
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        for _ in range(iters)]
  for a, b in ev:
    a.record()
    fn()
    b.record()
  torch.cuda.synchronize()
  return [a.elapsed_time(b) for a, b in ev]


def train_step_cpu_baseline() -> dict:
  """The secondary metric's `cpu_baseline` (run AFTER every GPU measurement of the line: the 128 host
  threads of a torch-CPU leg keep spinning for a while after it returns and slow the host side of
  whatever GPU work follows -- round 4 measured Model.fit at 0.37-0.52 ms per step right after the
  CPU legs against 0.137 ms without them)."""
  from oracle import cpu_path   # checker-side code: only this baseline leg uses it
  B, D, V = 4096, 64, 2000
  base = cpu_path.time_train_step(B, D, V, budget_s=4.0, with_metrics=True)
  base_off = cpu_path.time_train_step(B, D, V, budget_s=2.0, with_metrics=False)
  return {"value": base["value"], "unit": "steps/s", "cores": base["threads"], "kind": "port",
          "sample": "%d steps in %.1f s; torch-CPU restatement of the quickstart train "
                    "step WITH its FactorizedTopK update (lookup, sgemm logits, softmax CE, "
                    "14 candidate blocks of sgemm + top-k folded Streaming-style, in_top_k, "
                    "backward, Adagrad; not TensorFlow); without the metric update: %.1f "
                    "steps/s" % (base["steps"], base["seconds"], base_off["value"])}


def train_step_metric(dev) -> dict:
  """Second half of BASELINE.json's metric: train steps/sec of the in-batch-softmax two-tower
  step at the MovieLens-100K shapes of configs[0] (B=4096, D=64, 2k-row user/item tables), AS THE
  REFERENCE'S QUICKSTART RUNS IT (README.md:58-97): ``tfrs.Model.train_step`` of

      task = Retrieval(metrics=FactorizedTopK(candidates=movies.batch(128).map(item_model)))
      compute_loss = task(user_model(user_id), item_model(movie_id))     # compute_metrics=True

  i.e. embedding gather -> fused in-batch softmax loss (tasks/retrieval.py:172-210) ->
  FactorizedTopK.update_state over the 1682 candidates re-embedded through the item tower on every
  step (tasks/retrieval.py:216-226, metrics/factorized_top_k.py:91-194) -> backward -> sparse
  Adagrad on the looked-up rows -> the metrics dict (models/base.py:64-85).  `value` is that step
  under HIP-graph replay; the eager step and the same step with compute_metrics=False (round 2's
  figure) are reported beside it.  The arithmetic kernels are HIP; torch runs the autograd
  bookkeeping."""
  import recommenders_amd as tfrs
  from recommenders_amd.tasks import retrieval as rt
  g = torch.Generator(device=dev).manual_seed(0)
  B, D, V, ITEMS = 4096, 64, 2000, 1682

  class TwoTower(tfrs.Model):          # the reference's quickstart model (README.md:58-82)
    def __init__(self, with_metrics: bool):
      super().__init__()
      self.user_model = tfrs.layers.embedding.Embedding(V, D)
      self.item_model = tfrs.layers.embedding.Embedding(V, D)
      self._with_metrics = with_metrics
      if with_metrics:
        movies = tfrs.data.Dataset.from_tensor_slices(torch.arange(ITEMS, device=dev))
        self.task = tfrs.tasks.Retrieval(metrics=tfrs.metrics.FactorizedTopK(
            candidates=movies.batch(128).map(self.item_model)))
      else:
        self.task = tfrs.tasks.Retrieval()

    def compute_loss(self, inputs, training=False):
      return self.task(self.user_model(inputs["user_id"]), self.item_model(inputs["movie_id"]),
                       compute_metrics=self._with_metrics)

  batch = {"user_id": torch.randint(0, 943, (B,), generator=g, device=dev),
           "movie_id": torch.randint(0, ITEMS, (B,), generator=g, device=dev)}

  def timed(fn, iters):
    for _ in range(5):
      fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
      fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters

  def run(with_metrics: bool):
    model = TwoTower(with_metrics)
    model.compile(optimizer=tfrs.optimizers.Adagrad(model.parameters(), learning_rate=0.5))
    dt_eager = timed(lambda: model.train_step(batch), 100)    # models/base.py:64-85
    # the same train_step captured once in a HIP graph and replayed (models/base.py
    # make_graphed_train_step): identical kernels and arithmetic, no per-launch host cost;
    # every replay includes the copy of the batch into the graph's static input buffers
    graphed = model.make_graphed_train_step(batch)
    dt = timed(lambda: graphed(batch), 300)
    pct = percentiles(event_times_ms(lambda: graphed(batch), 100, 5))
    logs = graphed(batch)
    torch.cuda.synchronize()
    top100 = logs.get("factorized_top_k/top_100_categorical_accuracy")
    # ... and the call the README makes: model.fit(train.batch(4096)) over one MovieLens-100K epoch's
    # worth of batches (80 000 interactions: 19 x 4096 + one ragged 2176).  fit() captures a batch
    # shape the second time it sees it and replays it from then on (models/base.py); per epoch it
    # resets the metrics and reads the logs back (one host sync), as Keras does.
    sizes = [B] * 19 + [80_000 - 19 * B]
    epoch = [{"user_id": torch.randint(0, 943, (n,), generator=g, device=dev),
              "movie_id": torch.randint(0, ITEMS, (n,), generator=g, device=dev)} for n in sizes]
    fit_model = TwoTower(with_metrics)
    fit_model.compile(optimizer=tfrs.optimizers.Adagrad(fit_model.parameters(), learning_rate=0.5))
    # both shapes are captured by the end of epoch 2; 75 more untimed epochs (1500 graph launches, 0.2 s) take the
    # HIP runtime's one-off host stall of 50-85 ms -- it lands somewhere in a process's first ~1000 graph launches
    # (rounds 3 / 4: one epoch of 75-81 ms among fourteen of 2.6 ms) -- out of the timed epochs: it belongs to a
    # process's start-up, as the first-use costs of the search step's instrumentation do
    fit_model.fit(epoch, epochs=78)
    torch.cuda.synchronize()
    # 15 epochs, each timed on its own (fit() ends every epoch with a host read-back of the logs, so an
    # epoch is a closed interval).  The rate is taken from the MEDIAN epoch: the HIP runtime stalls the
    # host once per process for 50-75 ms somewhere in its first ~1000 graph launches (round 3 met the same
    # one-off behind the first instrumented search step; measured here: one epoch of 75.5 ms among
    # fourteen of 2.5-2.6 ms), which says nothing about the steady state of a training run.  The wall
    # clock over all 15 epochs and the slowest epoch are reported beside it.
    epochs = 15
    per_epoch = []
    hist = None
    for _ in range(epochs):
      t0 = time.perf_counter()
      hist = fit_model.fit(epoch, epochs=1)
      torch.cuda.synchronize()
      per_epoch.append(time.perf_counter() - t0)
    med = sorted(per_epoch)[len(per_epoch) // 2]
    dt_fit = med / len(sizes)
    captured = sum(callable(v) for v in fit_model.__dict__.get("_fit_graphs", {}).values())
    return {"fit_steps_per_s": 1.0 / dt_fit, "fit_ms_per_step": dt_fit * 1e3, "fit_captured_shapes": captured,
            "fit_wall_ms_per_step": sum(per_epoch) / (epochs * len(sizes)) * 1e3,
            "fit_slowest_epoch_ms": max(per_epoch) * 1e3, "fit_median_epoch_ms": med * 1e3,
            "fit_final_loss": hist["loss"][-1],
            "steps_per_s": 1.0 / dt, "ms_per_step": dt * 1e3, "step_ms_median": pct["median"],
            "step_ms_p10": pct["p10"], "step_ms_p90": pct["p90"],
            "eager_steps_per_s": 1.0 / dt_eager, "eager_ms_per_step": dt_eager * 1e3,
            "top_100_accuracy_running": None if top100 is None else float(top100)}

  on = run(True)
  off = run(False)

  # roofline of the step's dominant kernels: the fused in-batch softmax forward + backward
  # (prep/fwd/finalize + bwd/reduce launches of csrc/softmax16.hip), timed alone with HIP events
  # on the launch stream through the same autograd function the step uses
  q = synth(B, 11, dev)[:, :D].clone().requires_grad_(True)
  c = synth(B, 12, dev)[:, :D].clone().requires_grad_(True)
  one = torch.ones((), device=dev)

  def loss_fwd_bwd():
    loss = rt.in_batch_softmax_loss(q, c)
    torch.autograd.grad(loss, (q, c), grad_outputs=one)

  sm = percentiles(event_times_ms(loss_fwd_bwd, 50, 5))
  flop = 6.0 * B * B * D                     # fwd 2*B*Bc*D + bwd 4*B*Bc*D
  achieved = flop / (sm["median"] * 1e-3) / 1e12
  out = {"metric": "train steps/sec (in-batch softmax)", "value": on["fit_steps_per_s"], "unit": "steps/s",
         "ms_per_step": on["fit_ms_per_step"], "dtype": "f32 (split-fp16 MFMA)",
         "mode": "tfrs.Model.fit(batches) as the README calls it: 15 epochs x (19 x 4096 + 2176) batches after 78 "
                 "untimed epochs, captured-step replay per batch shape (default), metric reset + log read-back per "
                 "epoch; rate of the MEDIAN epoch, with the wall clock over all 15 epochs and the slowest epoch "
                 "beside it",
         "wall_ms_per_step_all_epochs": on["fit_wall_ms_per_step"], "slowest_epoch_ms": on["fit_slowest_epoch_ms"],
         "median_epoch_ms": on["fit_median_epoch_ms"],
         "fit_captured_shapes": on["fit_captured_shapes"],
         "graphed_step": {"note": "one captured train_step replayed on one fixed batch (what rounds 1-3 "
                                  "reported as the value)", "value": on["steps_per_s"], "unit": "steps/s",
                          "ms_per_step": on["ms_per_step"], "step_ms_median": on["step_ms_median"],
                          "step_ms_p10": on["step_ms_p10"], "step_ms_p90": on["step_ms_p90"]},
         "eager_steps_per_s": on["eager_steps_per_s"], "eager_ms_per_step": on["eager_ms_per_step"],
         "top_100_accuracy_running": on["top_100_accuracy_running"],
         "config": {"workload": "README-quickstart two-tower train step, MovieLens-100K shapes (BASELINE.json "
                                "configs[0]): batch 4096, dim 64, 2k x 64 user + item tables, Adagrad lr 0.5, "
                                "Retrieval(metrics=FactorizedTopK(candidates=movies.batch(128).map(item_model))) "
                                "called with compute_metrics=True: every step re-embeds the 1682 candidates and "
                                "updates top-1/5/10/50/100 accuracy", "batch": B, "dim": D,
                    "candidates": ITEMS, "candidate_batch": 128, "ks": [1, 5, 10, 50, 100]},
         "metrics_off": {"note": "the same step with compute_metrics=False (what round 2 reported)",
                         "value": off["fit_steps_per_s"], "unit": "steps/s", "ms_per_step": off["fit_ms_per_step"],
                         "graphed_step_steps_per_s": off["steps_per_s"],
                         "step_ms_median": off["step_ms_median"], "eager_steps_per_s": off["eager_steps_per_s"],
                         "eager_ms_per_step": off["eager_ms_per_step"]},
         "roofline": {"kernel": "in-batch softmax forward + backward (tfrs::sm16_* chain, split-fp16 MFMA: "
                                "3 fp16 products per f32 product), eager launches incl. autograd glue",
                      "bound": "mfma", "achieved": achieved, "peak": F16_MFMA_PEAK_TFLOPS,
                      "unit": "TFLOP/s", "frac": achieved / F16_MFMA_PEAK_TFLOPS, **softmax_chain_traffic(),
                      "algorithmic_flop": flop, "ms_median": sm["median"], "ms_p10": sm["p10"],
                      "ms_p90": sm["p90"],
                      "note": "4096^2 x 64 is 6.4 GFLOP: the chain is launch/latency bound at this "
                              "batch (DESIGN.md 4.5); the fraction is reported for completeness"}}
  return out


def gather_metric(dev) -> dict:
  """north_star evidence: embedding gather HBM GB/s at BASELINE configs[3] shapes (batch 65536 x
  26 categorical features, D = 128, 26 x 1M-row tables held as one 26M x 128 table = 13.3 GB),
  through the C ABI into a PRE-ALLOCATED output (nothing but the kernel between the HIP events;
  the rocprofv3 kernel row and the FETCH_SIZE / WRITE_SIZE counters of the same launches are in
  profiles/r03_gather_evidence.md), plus the dim-32 rows of configs[4]."""
  from recommenders_amd import _lib
  lib = _lib.load()
  out = {}
  g = torch.Generator(device=dev).manual_seed(3)
  for key, rows, d, n, what in (
      ("c3", 26_000_000, 128, 65536 * 26, "gather 65536 x 26 rows of dim 128 from 26 x 1M-row tables "
                                          "(BASELINE.json configs[3]), int64 ids uniform"),
      ("c4_rows", 100_000_000, 32, 131072 * 13, "gather 131072 x 13 rows of dim 32 from a 100M-row store "
                                                "(one GPU's share of BASELINE.json configs[4]), int64 ids uniform")):
    table = torch.empty((rows, d), dtype=torch.float32, device=dev).uniform_(-0.05, 0.05)
    ids = torch.randint(0, rows, (n,), generator=g, device=dev)
    dst = torch.empty((n, d), dtype=torch.float32, device=dev)
    stream = _lib.current_stream()

    def call():
      _lib.check(lib.tfrs_embedding_gather_fwd(_lib.ptr(table), rows, d, _lib.ptr(ids), 1, n, _lib.ptr(dst),
                                               None, stream))

    ts = percentiles(event_times_ms(call, 50, 5))
    # outside the timed region: the timed launches' output against the table through an independent
    # route (torch indexing), every row, plus rows on both sides of the 2^32-byte offset and the last row
    edge = (1 << 32) // (d * 4)
    probe = torch.tensor([0, edge - 1, edge, edge + 1, rows - 1], device=dev)
    ids[:5] = probe
    call()
    if not torch.equal(dst, table[ids]):
      raise SystemExit("bench.py: gather output differs from table[ids] (%s)" % key)
    nbytes = n * (2 * d * 4 + 8)            # rows * (D*4 read + D*4 write) + ids
    gbs = nbytes / (ts["median"] * 1e-3) / 1e9
    out[key] = {"value": gbs, "unit": "GB/s", "workload": what, "rows": n, "dim": d,
                "roofline": {"kernel": "tfrs::gather_kernel", "bound": "hbm", "achieved": gbs,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                             "traffic": None, "algorithmic_bytes": nbytes, "ms_median": ts["median"],
                             "ms_p10": ts["p10"], "ms_p90": ts["p90"]}}
    del table, ids, dst
    torch.cuda.empty_cache()
  c3 = out["c3"]
  return {"metric": "embedding gather", "value": c3["value"], "unit": "GB/s",
          "config": {"workload": c3["workload"], "rows": c3["rows"], "dim": c3["dim"]},
          "roofline": dict(c3["roofline"], **gather_traffic()),
          "configs4_rows_dim32": out["c4_rows"]}


def streaming_metric(dev) -> dict:
  """BASELINE.json configs[2], one GPU's shard: Streaming top-100 over a 12.5M x dim-128 candidate
  stream in blocks of 65536 rows handed over by a re-iterable dataset object (the layer keeps only a
  reference and re-reads it on every call, reference layers/factorized_top_k.py:384-390,:496-507; no
  packed copy of the stream is cached), at B = 8192 (MFMA-bound: one fp16 image of the group built
  straight from the blocks + fp16-prefiltered rounds, exact re-scoring from the blocks), B = 64 and
  B = 1 (HBM-bound: raw f32-MFMA scan of the blocks, every candidate byte read once).  Beside each:
  BruteForce over the same rows held as a resident index."""
  from recommenders_amd.layers import factorized_top_k as ftk
  n, d, k, bs = int(os.environ.get("TFRS_BENCH_STREAM_ROWS", 12_500_000)), 128, TOPK, 65536
  g = torch.Generator(device=dev).manual_seed(5)
  corpus = torch.randn((n, d), generator=g, device=dev) / (d ** 0.5)

  class Blocks:                     # a dataset object: iterated afresh by every call
    def __iter__(self):
      for lo in range(0, n, bs):
        yield corpus[lo:lo + bs]

  st = ftk.Streaming(k=k).index_from_dataset(Blocks())
  bf = ftk.BruteForce(k=k).index(corpus)
  out = {}
  for nq in (8192, 512, 128, 64, 1):
    q = torch.randn((nq, d), generator=g, device=dev) / (d ** 0.5)
    ts = percentiles(event_times_ms(lambda: st(q), 5, 2))
    tb = percentiles(event_times_ms(lambda: bf(q), 5, 2))
    a, b = st(q), bf(q)
    same = bool(torch.equal(a[0], b[0]) and torch.equal(a[1].long(), b[1].long()))
    if not same:
      raise SystemExit("bench.py: Streaming and BruteForce disagree at B = %d" % nq)
    ms = ts["median"]
    flop = 2.0 * nq * n * d
    if nq >= 1024:
      roof = {"kernel": "tfrs::scan16f_kernel<128, 8, 2> over the group's fp16 image (+ tfrs::pack16_raw_kernel)",
              "bound": "mfma", "achieved": flop / (ms * 1e-3) / 1e12, "peak": F16_MFMA_PEAK_TFLOPS,
              "unit": "TFLOP/s", "frac": flop / (ms * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS, "traffic": None,
              "algorithmic_flop": flop, "note": "whole call (packer, rounds, re-scoring, merges), not one launch"}
    else:
      nbytes = float(n) * d * 4 + nq * d * 4 + nq * k * 8     # N*D*s + B*D*s + B*K*8
      roof = {"kernel": ("tfrs::pack16_raw_kernel<128> + tfrs::scan16f_kernel<128, 8, 2> over the group's fp16 image "
                         "(between the regimes: the 1.9 ms packer dominates)" if nq > 256 else
                         "tfrs::rawscan16_kernel<128, %d> (fp16 filter fed by the f32 blocks; survivors re-scored "
                         "exactly)" % (1 if nq <= 32 else 2 if nq <= 64 else 4)), "bound": "hbm",
              "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
              "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
              "algorithmic_bytes": nbytes, "note": "whole call (exact dense round, 4 filtered ranges each with its list / re-score / merge "
                      "kernels), not one launch"}
    out["batch_%d" % nq] = {"value": nq / (ms * 1e-3), "unit": "queries/s", "ms_per_call": ms,
                            "ms_p10": ts["p10"], "ms_p90": ts["p90"],
                            "bruteforce_resident_index_ms": tb["median"], "vs_bruteforce": ms / tb["median"],
                            "equals_bruteforce": same, "roofline": roof}
  del st, bf, corpus
  torch.cuda.empty_cache()
  head = out["batch_8192"]
  return {"metric": "Streaming top-100 over a dataset of candidate blocks", "value": head["value"],
          "unit": "queries/s", "ms_per_call": head["ms_per_call"],
          "config": {"workload": "Streaming top-100, 12.5M x dim-128 stream (one GPU's shard of BASELINE.json "
                                 "configs[2]) in 65536-row blocks from a dataset object, batch 8192",
                     "rows": n, "dim": d, "block_rows": bs, "k": k},
          "roofline": head["roofline"], "batches": out}


def timed_region(fn, steps: int, warmup: int, world: int, dev):
  """W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize on both sides,
  maximum over ranks (the driver's contract); returns seconds."""
  for _ in range(warmup):
    fn()
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    fn()
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
  if world > 1:
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
  return float(t.item())


def run_streaming128(args, rank: int, world: int, dev, rccl_ranks: int):
  """BASELINE.json configs[2]: "Synthetic 100M-item x dim-128 corpus, Streaming top-100 sharded over 8
  MI355X via RCCL/xGMI".  STRONG scaling: the 100M x 128 stream is row-sharded over the N ranks (rank r
  streams rows [r * per, (r + 1) * per) in 65536-row blocks from a dataset object that is re-read on
  every call, reference layers/factorized_top_k.py:384-390,:404-509), every rank scores the same 8192
  queries (`ShardedStreaming`), one all_gather of the per-shard (score, row)[8192, 100] lists + merge
  inside the timed region; value = 8192 / t."""
  from recommenders_amd.layers import factorized_top_k as ftk
  total = int(os.environ.get("TFRS_BENCH_ROWS", SCALE_ROWS))
  d, bs = 128, 65536
  per = -(-total // world)
  lo, hi = rank * per, min(total, (rank + 1) * per)
  g = torch.Generator(device=dev).manual_seed(7)
  queries = torch.randn((BATCH, d), generator=g, device=dev) / (d ** 0.5)

  def shard_rows(a, b):
    # block k of the GLOBAL stream comes from seed 2000 + k: a shard's rows do not depend on N
    out = torch.empty((b - a, d), dtype=torch.float32, device=dev)
    k = a // INGEST_BLOCK
    pos = a
    while pos < b:
      gk = torch.Generator(device=dev).manual_seed(2000 + k)
      blk = torch.randn((INGEST_BLOCK, d), generator=gk, device=dev) / (d ** 0.5)
      end = min(b, (k + 1) * INGEST_BLOCK)
      out[pos - a:end - a] = blk[pos - k * INGEST_BLOCK:end - k * INGEST_BLOCK]
      pos, k = end, k + 1
    return out

  def dataset(rows):
    class Blocks:                       # re-iterated by every call: nothing is cached between calls
      def __iter__(self):
        for o in range(0, rows.shape[0], bs):
          yield rows[o:o + bs]
    return Blocks()

  rows = shard_rows(lo, hi)
  layer = ftk.ShardedStreaming(k=TOPK).index_from_dataset(dataset(rows), base_row=lo)
  elapsed = timed_region(lambda: layer(queries), args.steps, args.warmup, world, dev)
  value = BATCH / (elapsed / args.steps)
  if rank != 0:
    return None
  flop = 2.0 * BATCH * (hi - lo) * d
  ms = elapsed / args.steps * 1e3
  result = {"metric": "queries/sec streaming top-100", "value": value, "unit": "queries/s", "n_gpus": world,
            "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "filter_dtype": "f16", "data": "synthetic",
            "config": {"workload": "Streaming top-100, %dM-item x dim-128 stream row-sharded over %d GPU(s), "
                                   "65536-row blocks from a dataset object, batch 8192 (BASELINE.json configs[2])"
                                   % (total // 1_000_000, world),
                       "rows_total": total, "rows_per_gpu": hi - lo, "dim": d, "batch": BATCH, "k": TOPK,
                       "parallelism": ("single GPU" if world == 1 else
                                       f"stream row-sharded x{world}, one RCCL all_gather of per-shard top-K "
                                       "+ merge inside the timed region")},
            "roofline": {"kernel": "tfrs::scan16f_kernel<128, 8, 2> over the fp16 image of each group of blocks "
                                   "(+ tfrs::pack16_raw_kernel<128>)", "bound": "mfma",
                         "achieved": flop / (ms * 1e-3) / 1e12, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": flop / (ms * 1e-3) / 1e12 / F16_MFMA_PEAK_TFLOPS, "traffic": None,
                         "algorithmic_flop_per_step": flop, "note": "per-GPU flop over the whole call"},
            "exchange_bytes_per_rank": 2 * BATCH * TOPK * 4}
  if world > 1 and not args.no_single_gpu_reference:
    del layer, rows
    torch.cuda.empty_cache()
    allrows = shard_rows(0, total)
    single = ftk.Streaming(k=TOPK).index_from_dataset(dataset(allrows))
    for _ in range(2):
      single(queries)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
      single(queries)
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / 3
    result["single_gpu_same_workload"] = {"value": BATCH / one, "unit": "queries/s", "ms_per_step": one * 1e3,
                                          "steps": 3, "warmup": 2, "measured_on": "rank 0, after the timed region"}
    result["speedup_vs_single_gpu_same_workload"] = value / (BATCH / one)
  return result


def run_dlrm_embedding(args, rank: int, world: int, dev, rccl_ranks: int):
  """BASELINE.json configs[4]: "100 tables x 10M rows x dim-32, batch 131072, embedding-table row-sharded
  over 8 MI355X".  The 100 tables are held as ONE [10^9, 32] table whose rows are sharded over the N ranks
  (`ShardedEmbedding`: rank r owns rows [r * V/N, (r+1) * V/N)); the global batch of 131072 samples x 100
  features is split over the ranks (data parallel).  One step = lookup (owner bucketing on the device, two
  all-to-alls over xGMI) + backward (one all-to-all of gradient rows) + fused sparse Adagrad on the shard,
  with the split sizes of the NEXT batch staged while the current step runs (`ShardedEmbedding.stage`).
  STRONG scaling; value = samples / s of the global batch."""
  import recommenders_amd as tfrs
  from recommenders_amd.layers.sharded_embedding import ShardedEmbedding
  tables, per_table, d = 100, 10_000_000, 32
  gbatch = int(os.environ.get("TFRS_BENCH_BATCH", 131072))     # (shrunk for dry runs only)
  vocab = int(os.environ.get("TFRS_BENCH_TABLE_ROWS", tables * per_table))
  lbatch = gbatch // world
  layer = ShardedEmbedding(vocab, d, device=dev)
  opt = tfrs.optimizers.Adagrad(layer.parameters(), learning_rate=0.01)
  g = torch.Generator(device=dev).manual_seed(100 + rank)
  n_ids = lbatch * tables
  # feature f draws from table f's row range: ids = f * rows_per_table + uniform row
  rpt = vocab // tables
  offs = (torch.arange(tables, device=dev, dtype=torch.int64) * rpt)[None, :]
  batches = [torch.randint(0, rpt, (lbatch, tables), generator=g, device=dev) + offs for _ in range(4)]
  upstream = torch.randn((lbatch, tables, d), generator=g, device=dev)
  state = {"i": 0}
  layer.stage(batches[0])

  def step():
    ids = batches[state["i"] % len(batches)]
    state["i"] += 1
    opt.zero_grad(set_to_none=True)
    out = layer(ids)
    layer.stage(batches[state["i"] % len(batches)])      # the next batch's split sizes, one step ahead
    out.backward(upstream)
    opt.step()

  elapsed = timed_region(step, args.steps, args.warmup, world, dev)
  if rank != 0:
    return None
  ms = elapsed / args.steps * 1e3
  remote = (world - 1) / world
  # gather n * (D*4 read + D*4 write) + ids; scatter/Adagrad nnz*D*4 + 4*uniq*D*4 (uniq ~ nnz here)
  hbm = n_ids * (2 * d * 4 + 8) + n_ids * d * 4 * 5
  return {"metric": "DLRM embedding step (row-sharded lookup + backward + sparse Adagrad) samples/sec",
          "value": gbatch / (elapsed / args.steps), "unit": "samples/s", "n_gpus": world, "rccl_ranks": rccl_ranks,
          "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
          "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
          "config": {"workload": "100 tables x %d rows x dim-32 as one row-sharded table over %d GPU(s), global batch "
                                 "%d x 100 ids (BASELINE.json configs[4]): lookup + backward + fused sparse Adagrad"
                                 % (rpt, world, gbatch),
                     "table_rows_total": vocab, "table_rows_per_gpu": layer.row_range[1] - layer.row_range[0],
                     "dim": d, "global_batch": gbatch, "ids_per_gpu_per_step": n_ids,
                     "parallelism": ("single GPU" if world == 1 else
                                     f"rows sharded x{world}; per step and rank: all_to_all ids, all_to_all rows, "
                                     "all_to_all gradient rows")},
          "exchange_bytes_per_rank_per_step": int(n_ids * remote * (8 + 2 * d * 4)),
          "exchange_bytes_per_link_per_step": int(n_ids * (8 + 2 * d * 4) / world) if world > 1 else 0,
          "roofline": {"kernel": "tfrs::gather_kernel + tfrs::scatter_add_u32_kernel (fused Adagrad) on the shard",
                       "bound": "hbm", "achieved": hbm / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": hbm / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                       "algorithmic_bytes_per_gpu_per_step": hbm,
                       "note": "per-GPU HBM bytes of the local gather + scatter/Adagrad over the whole step "
                               "(the exchange is xGMI-bound: 7 links x ~153 GB/s)"}}


def robustness_block(dev, queries, ref_ms: float) -> dict:
  """The default (fp16-prefiltered) BruteForce path is data dependent: its thresholds come from
  sampled stages and its error margin from row norms.  Same shapes as the headline (1M x 64,
  batch 8192, top-100), four corpora that are NOT i.i.d. rows of equal norm: log-normal row norms
  (sigma 0.5 and 1.0 -- what trained embeddings look like), a cluster of 600 near-duplicates with a
  quarter of the queries aligned with it, and a Zipf-duplicated corpus (popular items repeated up to
  83 000 times: exact ties).  Reported: q/s, ms/step, the ratio to the
  i.i.d. step, and how many queries took the exact-redo path in the last step.  `BruteForce.index`
  detects bit-identical rows and indexes the distinct ones."""
  from recommenders_amd.layers import factorized_top_k as ftk
  g = torch.Generator(device=dev).manual_seed(99)
  out = {}

  def corpus(kind):
    base = torch.randn((N_ROWS, DIM), generator=g, device=dev) / (DIM ** 0.5)
    if kind.startswith("lognormal"):
      sigma = float(kind.split("_")[1])
      return base * torch.exp(sigma * torch.randn((N_ROWS, 1), generator=g, device=dev))
    if kind == "near_duplicates":
      return base
    # Zipf(1.0) popularity over 100k distinct items: item r has weight 1 / (r + 1)
    w = 1.0 / torch.arange(1, 100_001, device=dev, dtype=torch.float64)
    pick = torch.multinomial(w, N_ROWS, replacement=True, generator=g)
    return base[:100_000][pick].contiguous()

  for kind in ("lognormal_0.5", "lognormal_1.0", "near_duplicates", "zipf_duplicates"):
    c = corpus(kind)
    q_used = queries
    if kind == "near_duplicates":
      # a cluster of 600 rows within 2 eps of each other (distinct bit patterns) spread over the corpus,
      # and a quarter of the batch aligned with it: every such query's retained set exceeds K + band
      anchor = torch.randn((1, DIM), generator=g, device=dev) / (DIM ** 0.5)
      rows = torch.randperm(N_ROWS, generator=g, device=dev)[:600]
      c[rows] = anchor + 1e-7 * torch.randn((600, DIM), generator=g, device=dev)
      q_used = queries.clone()
      q_used[:BATCH // 4] = anchor * (1.0 + 3.0 * torch.rand((BATCH // 4, 1), generator=g, device=dev))
    index = ftk.BruteForce(k=TOPK).index(c)
    for _ in range(3):
      index(q_used)
    ts = percentiles(event_times_ms(lambda: index(q_used), 10, 0))
    dup = getattr(index, "_dup", None)
    out[kind] = {"value": BATCH / (ts["median"] * 1e-3), "unit": "queries/s", "ms_per_step": ts["median"],
                 "vs_iid_step": ts["median"] / ref_ms, "redo_queries_last_step": index.last_redo_count(),
                 "redo_reasons": index.last_redo_reasons(),
                 "distinct_rows_indexed": None if dup is None else dup.count}
    if kind == "zipf_duplicates":
      # (the same corpus with `dedup=False`: 82.9 ms per batch, 1 384 queries through the exact redo --
      # tools/exp_zipf.py; not re-measured here)
      out[kind]["without_dedup_ms_per_step_measured_by_tools_exp_zipf"] = 82.9
    del index, c
    torch.cuda.empty_cache()
  return out


def main() -> None:
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  if world != args.gpus:
    raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs a ROCm GPU: there is no CPU fallback for the product path")
  # Validation aid for boxes with ONE GPU (not used by the driver): TFRS_BENCH_ONE_GPU=1 runs the
  # N > 1 code path with every rank on cuda:0 and the single all-gather of the sharded path
  # staged through gloo/host, because RCCL refuses two ranks on one device.
  one_gpu = os.environ.get("TFRS_BENCH_ONE_GPU", "0") == "1"
  if not one_gpu and torch.cuda.device_count() < world:
    raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} device(s) are visible")
  if one_gpu:
    local_rank = 0
  torch.cuda.set_device(local_rank)
  dev = torch.device("cuda", local_rank)
  rccl_ranks = 1
  # TFRS_BENCH_FORCE_DIST=1 (validation aid, with TFRS_FORCE_EXCHANGE=1): a ONE-rank RCCL process group, so
  # that a single-GPU box executes the nccl branch of the sharded layers (all_gather / all_to_all of one rank)
  force_dist = world == 1 and os.environ.get("TFRS_BENCH_FORCE_DIST", "0") == "1"
  if force_dist:
    for key, val in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(29500 + os.getpid() % 2000)),
                     ("RANK", "0"), ("WORLD_SIZE", "1")):
      os.environ.setdefault(key, val)
  if world > 1 or force_dist:
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if one_gpu:
      dist.init_process_group("gloo")
      real_gather = dist.all_gather_into_tensor

      def staged_gather(out, inp, group=None):
        host = torch.empty(out.shape, dtype=out.dtype)
        real_gather(host, inp.cpu(), group=group)
        out.copy_(host)

      dist.all_gather_into_tensor = staged_gather
      rccl_ranks = 0
    else:
      dist.init_process_group("nccl", device_id=dev)
      # every rank must really be there: a one-element all_reduce over RCCL counts them
      ones = torch.ones((1,), device=dev)
      dist.all_reduce(ones)
      rccl_ranks = int(ones.item())
      if rccl_ranks != world:
        raise SystemExit(f"bench.py: {rccl_ranks} RCCL ranks answered, expected {world}")
      # librccl prints its version banner through C stdio at the first collective; when stdout is a pipe
      # it would otherwise sit in the buffer until exit and land AFTER the JSON line
      ctypes.CDLL(None).fflush(None)

  from recommenders_amd import _lib
  from recommenders_amd.layers import factorized_top_k as ftk
  lib = _lib.load()

  workload = args.workload
  if workload in ("streaming128", "dlrm_embedding"):
    fn = run_streaming128 if workload == "streaming128" else run_dlrm_embedding
    result = fn(args, rank, world, dev, rccl_ranks)
    if rank == 0:
      emit(result)
    if world > 1:
      try:
        dist.barrier()
        dist.destroy_process_group()
      except Exception as e:
        print(f"bench.py: process-group teardown: {e}", file=sys.stderr, flush=True)
    return
  if workload == "auto":
    workload = "headline" if world == 1 else "scale100m"
  if workload == "headline" and world > 1:
    raise SystemExit("bench.py: the 1M x 64 headline workload is the single-GPU configuration; "
                     "N > 1 runs the 100M x 64 strong-scaling workload")
  queries = synth(BATCH, seed=7, device=dev)               # same queries on every rank

  def build_index(total_rows: int):
    """Rank r indexes rows [r * per, min(total, (r + 1) * per)) of the total_rows corpus."""
    per = -(-total_rows // world)
    lo, hi = rank * per, min(total_rows, (rank + 1) * per)
    if world > 1:
      idx = ftk.ShardedBruteForce(k=TOPK)
      idx.index_from_dataset(corpus_blocks(lo, hi, dev), total_rows=hi - lo, base_row=lo)
    else:
      idx = ftk.BruteForce(k=TOPK).index_from_dataset(corpus_blocks(lo, hi, dev), total_rows=hi - lo)
    return idx, hi - lo

  def run_timed(index, steps: int, warmup: int):
    # The warm-up steps run with EXACTLY the instrumentation of the timed steps (per-step torch
    # events recorded, per-launch library events enabled): the first use of either costs tens of
    # milliseconds of one-off runtime set-up on the host (measured: 57 ms in the first instrumented
    # step), which belongs to the warm-up, not to the K timed steps.
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(steps)]
    lib.tfrs_profile_enable(1)
    for i in range(warmup):
      a, b = ev[i % steps]
      a.record()
      index(queries)
      b.record()
    torch.cuda.synchronize()
    lib.tfrs_profile_read(None, None, None)   # reset the per-launch timings (events stay created)
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()
    lib.tfrs_profile_enable(1)
    t0 = time.perf_counter()
    host_t = []
    for a, b in ev:
      a.record()
      out = index(queries)
      b.record()
      host_t.append(time.perf_counter())
    if os.environ.get("TFRS_BENCH_TRACE"):
      gaps = [round((y - x) * 1e3, 3) for x, y in zip([t0] + host_t[:-1], host_t)]
      print("host issue ms per step:", gaps, file=sys.stderr, flush=True)
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
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
      dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item()), kinds, [a.elapsed_time(b) for a, b in ev], out

  total_rows = N_ROWS if workload == "headline" else int(os.environ.get("TFRS_BENCH_ROWS", SCALE_ROWS))
  index, local_rows = build_index(total_rows)
  steps, warmup = args.steps, args.warmup
  elapsed, kinds, step_ms, out = run_timed(index, steps, warmup)
  ms_per_step = elapsed / steps * 1e3
  value = BATCH / (elapsed / steps)
  local = index._local if world > 1 else index
  redo = local.last_redo_count()

  if rank == 0:
    dom = max(kinds, key=lambda kk: kinds[kk][0])          # the kernel the step spends most time in
    scan_ms, launches, flop = kinds[dom]
    achieved = flop / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else 0.0
    peak = F16_MFMA_PEAK_TFLOPS if dom >= 1 else F32_MFMA_PEAK_TFLOPS
    pct = percentiles(step_ms)
    result = {
        "metric": "queries/sec brute-force top-100",
        "value": value,
        "unit": "queries/s",
        "n_gpus": world,
        "rccl_ranks": rccl_ranks,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": ms_per_step,
        "step_ms_median": pct["median"], "step_ms_p10": pct["p10"], "step_ms_p90": pct["p90"],
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",          # returned scores: exact f32 fma chains
        "filter_dtype": "f16",   # the prefilter (never returned): fp16 MFMA scores + proven error bound
        "redo_queries_last_step": redo,
        "data": "synthetic",
        "config": {
            "workload": ("BruteForce top-100, 1M-item x dim-64 corpus, batch 8192 (BASELINE.json configs[1])"
                         if workload == "headline" else
                         "BruteForce top-100, %dM-item x dim-64 corpus row-sharded over %d GPU(s), batch "
                         "8192 (north_star strong-scaling configuration)" % (total_rows // 1_000_000, world)),
            "corpus_rows_total": total_rows, "corpus_rows_per_gpu": local_rows, "dim": DIM,
            "batch": BATCH, "k": TOPK,
            "parallelism": ("single GPU" if world == 1 else
                            f"corpus row-sharded x{world}, one RCCL all_gather of per-shard top-K + merge "
                            "inside the timed region"),
        },
        "roofline": {
            "kernel": (SCAN16F_KERNEL + " (fp16 MFMA prefilter scores of all rows + fused top-K "
                       "filter; survivors re-scored exactly in f32)" if dom >= 1 else
                       "tfrs::scan_kernel<64> (f32 MFMA scores + fused top-K filter)"),
            "bound": "mfma",
            "achieved": achieved,
            "peak": peak,
            "unit": "TFLOP/s",
            "frac": achieved / peak,
            "traffic": hbm_traffic(SCAN16F_KERNEL if dom >= 1 else "tfrs::scan_kernel<64, false, true>"),
            "launches": launches,
            "avg_launch_ms": scan_ms / max(launches, 1),
            "algorithmic_flop_per_launch": flop / max(launches, 1),
            "algorithmic_flop_per_step": 2.0 * BATCH * local_rows * DIM,
            "f32_scan_ms_per_step": kinds[0][0] / steps,
            "f16_filter_pass_ms_per_step": kinds[1][0] / steps,
            "f16_threshold_pass_ms_per_step": kinds[2][0] / steps,
            "peak_note": "2.5 PFLOP/s = dense fp16/bf16 MFMA spec; tools/ubench/mfma_peak.hip (a saturating MFMA "
                         "loop: persistent accumulators, no memory traffic) reaches 2.49 PFLOP/s on all-zero operands at "
                         "2.4 GHz and sustains 1.61-1.63 PFLOP/s on random fp16 operands at ~1.6 GHz on this chip: "
                         "the clock follows the power the operand data draws (profiles/r05_mfma_peak.txt)",
        },
    }
    # this box's own ceilings, measured after the timed region (csrc/calibrate.hip through bench_legs.measure_ceilings)
    import bench_legs
    ceilings = bench_legs.measure_ceilings(dev)
    roof = result["roofline"]
    roof["shader_mhz"] = ceilings["shader_mhz"]
    roof["measured_copy_gbs"] = ceilings["copy_gbs"]
    if dom >= 1:
      roof["measured_ceiling"] = ceilings["mfma_f16_tflops"]
      roof["frac_of_measured_ceiling"] = achieved / ceilings["mfma_f16_tflops"]
    roof["measured_ceiling_note"] = ceilings["note"]
    if world == 1 and workload == "headline":
      cpu_inputs = None
      if not args.no_cpu_baseline:   # (inputs of the CPU leg, which runs after every GPU measurement;
        cpu_inputs = (index.candidates(), out[1][:64].clone())   # still on the device here)
      if not args.no_train_step:
        result["secondary"] = train_step_metric(dev)
      if not args.no_gather:
        result["gather"] = gather_metric(dev)
      if not args.no_streaming:
        result["streaming"] = streaming_metric(dev)
      if not args.no_robustness:
        result["robustness"] = robustness_block(dev, queries, pct["median"])
      if not args.no_config_legs:
        # BASELINE configs[3] / configs[4]: Cross, DotInteraction, segment-sum, sparse Adagrad and the two
        # ranking train steps, each with roofline + parity assert (bench_legs.py); CPU baselines further down
        result["config_legs"] = bench_legs.gpu_legs(dev, ceilings)
      if not args.no_scale_workload:
        # the N = 1 point of the strong-scaling configuration (what `value` at N > 1 compares with)
        del index, local
        torch.cuda.empty_cache()
        rows = int(os.environ.get("TFRS_BENCH_ROWS", SCALE_ROWS))
        big, _ = build_index(rows)
        el, kk, sms, _ = run_timed(big, 5, 2)
        p = percentiles(sms)
        result["scale_workload"] = {
            "workload": "BruteForce top-100, %dM-item x dim-64 corpus on ONE GPU, batch 8192 (N = 1 point "
                        "of the strong-scaling configuration that `bench.py --gpus N` runs for N > 1)"
                        % (rows // 1_000_000),
            "value": BATCH / (el / 5), "unit": "queries/s", "ms_per_step": el / 5 * 1e3,
            "step_ms_median": p["median"], "steps": 5, "warmup": 2,
            "filter_pass_tflops": kk[1][2] / max(kk[1][0] * 1e-3, 1e-12) / 1e12,
            "redo_queries_last_step": big.last_redo_count()}
        # What ONE GPU can show of the 8-GPU configuration (no multi-GPU node is available to this build; the
        # scaling curve itself is the driver's SCALE run): the per-shard search (rows [0, rows / 8) of the same
        # corpus) and the merge of eight [8192, 100] lists are MEASURED here; the all_gather is priced from the
        # guide's xGMI figure (every rank sends its 2 * 8192 * 100 * 4 bytes to 7 peers over 7 separate links).
        del big
        torch.cuda.empty_cache()
        shards = 8
        shard_rows = -(-rows // shards)
        shard = ftk.BruteForce(k=TOPK).index_from_dataset(corpus_blocks(0, shard_rows, dev), total_rows=shard_rows)
        sh_ms = percentiles(event_times_ms(lambda: shard(queries), 5, 2))
        s_sc, s_rows = shard(queries)
        gathered = torch.empty((shards, 2, BATCH, TOPK), dtype=torch.int32, device=dev)
        for r in range(shards):           # eight parts with distinct global rows (a real exchange's layout)
          gathered[r, 0].copy_(s_sc.contiguous().view(torch.int32))
          gathered[r, 1].copy_(s_rows.to(torch.int32) + r * shard_rows)
        out_s = torch.empty((BATCH, TOPK), dtype=torch.float32, device=dev)
        out_i = torch.empty((BATCH, TOPK), dtype=torch.int32, device=dev)
        base = gathered.view(-1)

        def merge():
          _lib.check(lib.tfrs_topk_merge_strided(base.data_ptr(), base.data_ptr() + BATCH * TOPK * 4, shards,
                                                 2 * BATCH * TOPK, BATCH, TOPK, TOPK, _lib.ptr(out_s),
                                                 _lib.ptr(out_i), _lib.current_stream()))

        mg_ms = percentiles(event_times_ms(merge, 10, 2))
        merge()
        torch.cuda.synchronize()
        # parity outside the timed region: every score occurs in all eight parts, so the (score desc, row asc)
        # rule decides most of the output; checked against a stable two-key sort of the concatenation
        all_s = gathered[:, 0].contiguous().view(torch.float32).permute(1, 0, 2).reshape(BATCH, shards * TOPK)
        all_r = gathered[:, 1].permute(1, 0, 2).reshape(BATCH, shards * TOPK)
        o1 = all_r.argsort(dim=1, stable=True)
        s1, r1 = all_s.gather(1, o1), all_r.gather(1, o1)
        s2, o2 = torch.sort(s1, dim=1, descending=True, stable=True)
        if not (torch.equal(out_s, s2[:, :TOPK]) and torch.equal(out_i, r1.gather(1, o2)[:, :TOPK])):
          raise SystemExit("bench.py: merge of eight parts differs from the (score desc, row asc) sort")
        link_gbs, link_latency_us = 153.0, 20.0      # MI355X_MICROARCH.md: 7 xGMI links x ~153 GB/s per GPU
        exch_ms = 2 * BATCH * TOPK * 4 / (link_gbs * 1e9) * 1e3 + link_latency_us * 1e-3
        t8 = sh_ms["median"] + exch_ms + mg_ms["median"]
        result["scale_workload"]["predicted_8gpu"] = {
            "per_shard_search_ms_measured": sh_ms["median"], "merge_8_parts_ms_measured": mg_ms["median"],
            "all_gather_ms_modelled": exch_ms,
            "all_gather_model": "2 * 8192 * 100 * 4 B per rank to each of 7 peers over 7 separate xGMI links at "
                                "%.0f GB/s per link + %.0f us latency (guide figures; NOT measured)" % (link_gbs, link_latency_us),
            "predicted_ms_per_step": t8, "predicted_value": BATCH / (t8 * 1e-3), "unit": "queries/s",
            "predicted_speedup_8gpu": (el / 5 * 1e3) / t8,
            "note": "a prediction from single-GPU measurements, not a scaling measurement: assumes all eight shards "
                    "take the time rank 0's shard takes here and ignores clock / power interactions between GPUs"}
        del shard, gathered
    if world == 1 and workload == "headline" and not args.no_cpu_baseline:
      # the CPU legs come LAST: their host threads would disturb the GPU measurements above
      from oracle import cpu_path  # checker-side code: only this baseline leg uses it
      corpus_host, q_host, gpu_idx64 = cpu_inputs[0].cpu().numpy(), queries.cpu().numpy(), cpu_inputs[1].cpu().numpy()
      base = cpu_path.time_brute_force(corpus_host, q_host, TOPK, budget_s=args.cpu_budget)
      # sanity: the timed CPU path agrees with the GPU result on its first block
      v, i = cpu_path.brute_force_topk(q_host[:64], corpus_host, TOPK)
      agree = float((i == gpu_idx64).mean())
      del corpus_host
      result["cpu_baseline"] = {
          "value": base["value"], "unit": "queries/s", "cores": base["threads"],
          "kind": "port",
          "sample": "%d queries x full 1M x 64 corpus in blocks of 256, %.1f s; torch-CPU "
                    "sgemm + topk restatement of BruteForce.call (not TensorFlow); index "
                    "agreement with the GPU on 64 queries: %.4f"
                    % (base["queries"], base["seconds"], agree),
      }
      if "secondary" in result:
        result["secondary"]["cpu_baseline"] = train_step_cpu_baseline()
      if "config_legs" in result:
        import bench_legs
        bench_legs.add_cpu_baselines(result["config_legs"])
    if world > 1 and not args.no_single_gpu_reference:
      # the SAME workload on one GPU, measured by rank 0 after the timed region (the other ranks
      # wait at the final barrier): makes the N > 1 line self-contained -- speedup = value / this --
      # because the N = 1 bench line is the 1M x 64 headline, not this corpus
      del index, local
      torch.cuda.empty_cache()
      single = ftk.BruteForce(k=TOPK).index_from_dataset(corpus_blocks(0, total_rows, dev),
                                                         total_rows=total_rows)
      for _ in range(2):
        single(queries)
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(3):
        single(queries)
      torch.cuda.synchronize()
      one = (time.perf_counter() - t0) / 3
      result["single_gpu_same_workload"] = {"value": BATCH / one, "unit": "queries/s", "ms_per_step": one * 1e3,
                                            "steps": 3, "warmup": 2, "measured_on": "rank 0, after the timed region"}
      result["speedup_vs_single_gpu_same_workload"] = value / (BATCH / one)
      del single
    emit(result)
  sys.stdout.flush()
  if world > 1:
    try:
      dist.barrier()
      dist.destroy_process_group()
    except Exception as e:   # the result line is out: a teardown hiccup must not fail the run
      print(f"bench.py: process-group teardown: {e}", file=sys.stderr, flush=True)


if __name__ == "__main__":
  main()
