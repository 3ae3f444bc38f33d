"""bench.py's legs for BASELINE.json configs[3] / configs[4] (round 5; VERDICT round 4 missing 2): the kernels
north_star names beside the top-K path -- the fused Cross layer, DotInteraction, embedding segment-sum, the
fused sparse Adagrad -- and the two ranking train steps they compose into, each with its own `roofline`
(ALGORITHMIC flop / bytes of SURVEY.md 8(d) over HIP-event medians on the launch stream), the same process's
copy rate for the HBM-bound ones, a parity assert OUTSIDE the timed region (float64 restatement of the reference
formula on sampled rows, computed by torch on the device -- an independent route, not the oracle), and a
`cpu_baseline` (oracle/cpu_path.py restatements on the host cores, bounded samples scaled per example; run by
bench.py after every GPU measurement of the line).

    gpu_legs(dev)            -> {"cross": {...}, "dot_interaction": {...}, "segment_sum": {...},
                                 "sparse_adagrad": {...}, "dcn_v2_step": {...}, "dlrm_shard_step": {...}}
    add_cpu_baselines(legs)   (imports oracle.cpu_path: checker-side code, only this leg uses it)
"""

import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

import torch  # noqa: E402

F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16/f16 MFMA, dense (no sparsity)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured copy ceiling)


def _pct(xs):
  xs = sorted(xs)
  n = len(xs)
  pick = lambda p: xs[min(n - 1, max(0, int(round(p * (n - 1)))))]
  return {"median": pick(0.5), "p10": pick(0.1), "p90": pick(0.9)}


def _event_ms(fn, iters, warmup):
  for _ in range(warmup):
    fn()
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
  for a, b in ev:
    a.record()
    fn()
    b.record()
  torch.cuda.synchronize()
  return _pct([a.elapsed_time(b) for a, b in ev])


def measure_ceilings(dev, mfma_iters: int = 4000, copy_bytes: int = 2 << 30) -> dict:
  """What THIS box sustains, measured by the library in this process (csrc/calibrate.hip): the fp16 MFMA rate of a
  saturating register-only loop on uniform random operands (+ the shader clock it ran at) and the rate of a float4
  read + write copy.  Fractions of the spec peaks are only comparable between boxes next to these (VERDICT round 5,
  next 1 / 5: the torch `copy_` this replaced ran at 4.56 TB/s on a box whose gather kernel sustained 6.3)."""
  import ctypes
  from recommenders_amd import _lib
  lib = _lib.load()
  st = _lib.current_stream()
  ws = torch.empty((max(copy_bytes, lib.tfrs_calibrate_workspace_bytes()),), dtype=torch.uint8, device=dev)
  tf, mhz, gbs = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
  _lib.check(lib.tfrs_calibrate_mfma_f16(_lib.ptr(ws), ws.numel(), mfma_iters, ctypes.byref(tf), ctypes.byref(mhz), st))
  _lib.check(lib.tfrs_calibrate_copy(_lib.ptr(ws), ws.numel(), 10, ctypes.byref(gbs), st))
  del ws
  return {"mfma_f16_tflops": tf.value, "shader_mhz": mhz.value, "copy_gbs": gbs.value,
          "note": "tfrs_calibrate_mfma_f16 (saturating v_mfma_f32_32x32x16_f16 loop, uniform random fp16 operands in "
                  "registers, 2 waves per SIMD) and tfrs_calibrate_copy (float4 copy, %d MiB read + %d MiB written per "
                  "launch), both in this process on this box" % (copy_bytes >> 21, copy_bytes >> 21)}


def _hbm_roof(kernel, nbytes, ts, copy_gbs, **extra):
  gbs = nbytes / (ts["median"] * 1e-3) / 1e9
  out = {"kernel": kernel, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": gbs / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes": nbytes, "ms_median": ts["median"],
         "ms_p10": ts["p10"], "ms_p90": ts["p90"], "measured_copy_gbs": copy_gbs,
         "frac_of_measured_ceiling": gbs / copy_gbs}
  out.update(extra)
  return out


_MFMA_CEILING = {"tflops": None}   # set by gpu_legs() from measure_ceilings()


def _mfma_roof(kernel, flop, ts, products=3, **extra):
  tf = flop / (ts["median"] * 1e-3) / 1e12
  out = {"kernel": kernel, "bound": "mfma", "achieved": tf, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
         "frac": tf / F16_MFMA_PEAK_TFLOPS, "traffic": None, "algorithmic_flop": flop, "ms_median": ts["median"],
         "ms_p10": ts["p10"], "ms_p90": ts["p90"],
         "fp16_products_per_f32_product": products,
         "frac_of_fp16_pipe_incl_split": products * tf / F16_MFMA_PEAK_TFLOPS,
         "note": "f32-grade results on the fp16 matrix cores: every f32 product is hi*hi + hi*lo + lo*hi (3 fp16 MFMA "
                 "products, f32 accumulation); `achieved` / `frac` count the ALGORITHMIC flop once, "
                 "`frac_of_fp16_pipe_incl_split` what the pipe executes"}
  if _MFMA_CEILING["tflops"]:
    out["measured_ceiling"] = _MFMA_CEILING["tflops"]
    out["frac_of_measured_ceiling"] = products * tf / _MFMA_CEILING["tflops"]
  out.update(extra)
  return out


def _rel_err(got, ref):
  return float((got.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


# ------------------------------------------------------------------------------------------------ Cross
def cross_leg(dev) -> dict:
  """The fused Cross layer at BASELINE configs[3]: B = 65536, d = 27 * 128 = 3456, full rank
  (reference layers/feature_interaction/dcn.py:151-186).  `forward` = inference (`tfrs_cross_fwd_f16`),
  `training_pair` = forward that also stores u = x W + b + diag x + the fused backward
  (`tfrs_cross_fwd_f16_train` + `tfrs_cross_bwd_f16_saved`: dx0, dx, dW, db)."""
  import recommenders_amd as tfrs
  from recommenders_amd import _lib
  from recommenders_amd.layers.feature_interaction import dcn
  lib = _lib.load()
  g = torch.Generator(device=dev).manual_seed(31)
  b, d = 65536, 3456
  x0 = torch.randn((b, d), generator=g, device=dev) * 0.5
  x = torch.randn((b, d), generator=g, device=dev) * 0.5
  dy = torch.randn((b, d), generator=g, device=dev)
  layer = tfrs.layers.feature_interaction.Cross()
  layer.build((b, d), dev)
  with torch.no_grad():
    layer.bias.uniform_(-0.1, 0.1, generator=g)
  w, bias = layer.kernel.detach(), layer.bias.detach()
  y, u = torch.empty_like(x0), torch.empty_like(x0)
  dx0, dx, dk, db = torch.empty_like(x0), torch.empty_like(x0), torch.empty_like(w), torch.empty_like(bias)
  ws = dcn._gemm_workspace(max(lib.tfrs_gemm_f16_workspace_bytes(b, d, d),
                               lib.tfrs_cross_bwd_workspace_bytes(b, d, 1)), dev)
  st = _lib.current_stream()

  def fwd():
    _lib.check(lib.tfrs_cross_fwd_f16(_lib.ptr(x0), _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), 0.0, b, d,
                                      _lib.ptr(y), _lib.ptr(ws), ws.numel(), st))

  def fwd_train():
    _lib.check(lib.tfrs_cross_fwd_f16_train(_lib.ptr(x0), _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), 0.0, b, d,
                                            _lib.ptr(y), _lib.ptr(u), _lib.ptr(ws), ws.numel(), st))

  def bwd_saved():
    _lib.check(lib.tfrs_cross_bwd_f16_saved(_lib.ptr(x0), _lib.ptr(x), _lib.ptr(u), _lib.ptr(w), 0.0, _lib.ptr(dy),
                                            b, d, _lib.ptr(dx0), _lib.ptr(dx), _lib.ptr(dk), _lib.ptr(db),
                                            _lib.ptr(ws), ws.numel(), st))

  def pair():
    fwd_train()
    bwd_saved()

  t_f = _event_ms(fwd, 10, 3)
  t_p = _event_ms(pair, 8, 2)
  t_b = _event_ms(bwd_saved, 8, 1)
  # parity, outside the timed region: sampled rows of y / dx0 / dx and a 64 x 64 corner of dW against float64
  fwd()
  rows = torch.tensor(list(range(16)) + list(range(b // 2, b // 2 + 16)) + list(range(b - 16, b)), device=dev)
  w64, b64 = w.double(), bias.double()
  xr, x0r, dyr = x[rows].double(), x0[rows].double(), dy[rows].double()
  z = xr @ w64 + b64
  errs = {"y": _rel_err(y[rows], x0r * z + xr)}
  pair()
  dz = dyr * x0r
  errs["y_train"] = _rel_err(y[rows], x0r * z + xr)
  errs["dx0"] = _rel_err(dx0[rows], dyr * z)
  errs["dx"] = _rel_err(dx[rows], dz @ w64.t() + dyr)
  corner = torch.zeros((64, 64), dtype=torch.float64, device=dev)
  dbc = torch.zeros((64,), dtype=torch.float64, device=dev)
  for lo in range(0, b, 16384):
    dzc = (dy[lo:lo + 16384, :64] * x0[lo:lo + 16384, :64]).double()
    corner += x[lo:lo + 16384, :64].double().t() @ dzc
    dbc += dzc.sum(dim=0)
  errs["dW[:64,:64]"] = _rel_err(dk[:64, :64], corner)
  errs["db[:64]"] = _rel_err(db[:64], dbc)
  if max(errs.values()) > 1e-5:
    raise SystemExit("bench.py: Cross kernels differ from the float64 restatement: %r" % errs)
  fl = 2.0 * b * d * d
  out = {"metric": "fused Cross layer (DCN-v2)", "value": b / (t_f["median"] * 1e-3), "unit": "examples/s",
         "config": {"workload": "Cross full rank, batch 65536, width 3456 = 27 x 128 (BASELINE.json configs[3])",
                    "batch": b, "width": d},
         "dtype": "f32 (split-fp16 MFMA)",
         "forward": {"ms": t_f["median"], "roofline": _mfma_roof(
             "tfrs::gemm16_big_kernel<cross epilogue> (tfrs_cross_fwd_f16)", fl, t_f)},
         "training_pair": {"ms": t_p["median"], "backward_ms": t_b["median"], "roofline": _mfma_roof(
             "tfrs_cross_fwd_f16_train + tfrs_cross_bwd_f16_saved (3 fused GEMMs: x W, dz W^T, x^T dz)",
             3.0 * fl, t_p)},
         "parity_max_rel_err_vs_float64": errs}
  out["roofline"] = out["forward"]["roofline"]
  del x0, x, dy, y, u, dx0, dx, dk, db, layer
  return out


# ------------------------------------------------------------------------------------------------ DotInteraction
def dot_interaction_leg(dev, copy_gbs) -> dict:
  """DotInteraction at BASELINE configs[4]: B = 131072, F = 101 vectors of dim 32, strict lower triangle
  (reference layers/feature_interaction/dot_interaction.py:53-104), forward and backward through the C ABI."""
  from recommenders_amd import _lib
  lib = _lib.load()
  g = torch.Generator(device=dev).manual_seed(32)
  b, f, d = 131072, 101, 32
  od = f * (f - 1) // 2
  x = torch.randn((b, f, d), generator=g, device=dev)
  out = torch.empty((b, od), device=dev)
  dout = torch.randn((b, od), generator=g, device=dev)
  dx = torch.empty_like(x)
  st = _lib.current_stream()
  fwd = lambda: _lib.check(lib.tfrs_dot_interaction_fwd(_lib.ptr(x), b, f, d, 0, 0, _lib.ptr(out), st))
  bwd = lambda: _lib.check(lib.tfrs_dot_interaction_bwd(_lib.ptr(x), _lib.ptr(dout), b, f, d, 0, 0, _lib.ptr(dx), st))
  t_f = _event_ms(fwd, 20, 3)
  t_b = _event_ms(bwd, 12, 3)
  rows = torch.tensor(list(range(8)) + list(range(b // 2, b // 2 + 8)) + list(range(b - 8, b)), device=dev)
  xr = x[rows].double()
  ii, jj = torch.tril_indices(f, f, -1, device=dev)
  gram = xr @ xr.transpose(1, 2)
  gm = torch.zeros((rows.numel(), f, f), dtype=torch.float64, device=dev)
  gm[:, ii, jj] = dout[rows].double()
  errs = {"fwd": _rel_err(out[rows], gram[:, ii, jj]),
          "bwd": _rel_err(dx[rows], (gm + gm.transpose(1, 2)) @ xr)}
  if max(errs.values()) > 1e-5:
    raise SystemExit("bench.py: DotInteraction kernels differ from the float64 restatement: %r" % errs)
  by_f = (b * f * d + b * od) * 4.0              # B*F*D*4 + B*out_dim*4
  by_b = (2.0 * b * f * d + b * od) * 4.0
  res = {"metric": "DotInteraction (DLRM)", "value": b / (t_f["median"] * 1e-3), "unit": "examples/s",
         "config": {"workload": "DotInteraction, batch 131072, 101 x dim 32 (BASELINE.json configs[4])",
                    "batch": b, "features": f, "dim": d},
         "dtype": "f32",
         "forward": {"ms": t_f["median"], "roofline": _hbm_roof("tfrs::dot_interaction_fwd_pc_kernel", by_f, t_f, copy_gbs)},
         "backward": {"ms": t_b["median"], "roofline": _hbm_roof("tfrs::dot_interaction_bwd_h16_kernel", by_b, t_b, copy_gbs)},
         "parity_max_rel_err_vs_float64": errs}
  res["roofline"] = res["forward"]["roofline"]
  del x, out, dout, dx
  return res


# ------------------------------------------------------------------------------------------------ embedding
def embedding_legs(dev, copy_gbs) -> dict:
  """Segment-sum lookup and the fused sparse Adagrad on the 26M x 128 table of configs[3] (reference
  tpu_embedding_layer.py:913-919 combiner lookup; models/base.py:77-78 + README.md:84 Adagrad on IndexedSlices)."""
  from recommenders_amd.layers import embedding as emb
  g = torch.Generator(device=dev).manual_seed(33)
  vocab, d, n, bag = 26_000_000, 128, 65536 * 26, 8
  table = torch.empty((vocab, d), device=dev).uniform_(-0.05, 0.05)
  ids = torch.randint(0, vocab, (n,), generator=g, device=dev)
  nb = n // bag
  splits = torch.arange(0, nb + 1, device=dev) * bag
  t_s = _event_ms(lambda: emb.embedding_lookup_sparse(table, ids, splits, combiner="sum"), 20, 3)
  got = emb.embedding_lookup_sparse(table, ids, splits, combiner="sum")
  pick = torch.tensor([0, 1, nb // 2, nb - 1], device=dev)
  want = torch.stack([table[ids[int(p) * bag:(int(p) + 1) * bag]].double().sum(dim=0) for p in pick])
  err_s = _rel_err(got[pick], want)
  if err_s > 1e-6:
    raise SystemExit("bench.py: segment-sum differs from the float64 restatement: %g" % err_s)
  by_s = n * d * 4.0 + nb * d * 4.0 + n * 8.0 + (nb + 1) * 8.0     # nnz*D*4 + B*D*4 + ids + splits
  seg = {"metric": "embedding segment-sum lookup", "value": n / (t_s["median"] * 1e-3), "unit": "ids/s",
         "config": {"workload": "sum-combiner lookup, 1.7M ids in bags of 8 from a 26M x 128 table (configs[3] tables)",
                    "nnz": n, "bags": nb, "dim": d, "vocab": vocab}, "dtype": "f32",
         "roofline": _hbm_roof("tfrs::segment_reduce_kernel", by_s, t_s, copy_gbs), "parity_max_rel_err_vs_float64": err_s}
  del got
  # fused sparse Adagrad: own radix sort + segmented sum of duplicate rows + update of the touched rows
  acc = torch.full_like(table, 0.1)
  go = torch.randn((n, d), generator=g, device=dev)
  lr = 0.5
  t_a = _event_ms(lambda: emb.adagrad_sparse_update_(table, acc, go, ids, lr), 10, 2)
  # outside the timed region: one more update, 256 touched rows against the float64 restatement --
  # duplicates summed, acc += g^2, row -= lr g / sqrt(acc + eps)
  up = torch.unique(ids[:4096])[:256]
  before_t, before_a = table[up].double(), acc[up].double()
  emb.adagrad_sparse_update_(table, acc, go, ids, lr)
  gsum = torch.stack([go[ids == u].double().sum(dim=0) for u in up])
  a64 = before_a + gsum * gsum
  t64 = before_t - lr * gsum / torch.sqrt(a64 + 1e-7)
  err_a = max(_rel_err(table[up], t64), _rel_err(acc[up], a64))
  if err_a > 1e-5:
    raise SystemExit("bench.py: sparse Adagrad differs from the float64 restatement: %g" % err_a)
  uniq = int(torch.unique(ids).numel())
  by_a = n * d * 4.0 + 4.0 * uniq * d * 4 + n * 8.0               # nnz*D*4 + uniq*D*4 (r+w) x (table, acc) + ids
  ada = {"metric": "fused sparse Adagrad (scatter-add backward)", "value": n / (t_a["median"] * 1e-3), "unit": "ids/s",
         "config": {"workload": "Adagrad on an IndexedSlices gradient of 1.7M rows of dim 128 into a 26M x 128 table "
                                "(configs[3]): own radix sort + duplicate sum + row update", "nnz": n, "unique": uniq,
                    "dim": d, "vocab": vocab}, "dtype": "f32",
         "roofline": _hbm_roof("tfrs::sort_*_kernel + tfrs::scatter_add_u32_kernel (fused Adagrad)", by_a, t_a, copy_gbs),
         "parity_max_rel_err_vs_float64": err_a}
  del table, acc, go, ids
  torch.cuda.empty_cache()
  return {"segment_sum": seg, "sparse_adagrad": ada}


# ------------------------------------------------------------------------------------------------ ranking steps
def ranking_step_leg(dev, kind: str) -> dict:
  """One `tfrs.Model.train_step` of `experimental.models.Ranking` (reference
  experimental/models/ranking.py:135-236 under models/base.py:64-85) with this package's Adagrad:
  kind "dcn_v2" = configs[3] (26 x 1M x 128 tables, 13 dense, 3 Cross layers of width 3456, batch 65536);
  kind "dlrm_shard" = one GPU's share of configs[4] (100 tables x 1.25M rows x 32 -- the 1/8 row shard --
  DotInteraction over 101 vectors, batch 131072)."""
  import recommenders_amd as tfrs
  from recommenders_amd.experimental.models import ranking as rk
  g = torch.Generator(device=dev).manual_seed(34)
  if kind == "dcn_v2":
    n_tables, vocab, dim, batch = 26, 1_000_000, 128, 65536
    fi = rk.ConcatCross(num_layers=3)
  else:
    n_tables, vocab, dim, batch = 100, 1_250_000, 32, 131072
    fi = tfrs.layers.feature_interaction.DotInteraction()
  emb = rk.EmbeddingDict({str(i): vocab for i in range(n_tables)}, dim)
  bottom = tfrs.layers.blocks.MLP(units=[512, 256, dim], final_activation="relu")
  top = tfrs.layers.blocks.MLP(units=[1024, 512, 1], final_activation="sigmoid")
  model = rk.Ranking(emb, bottom_stack=bottom, feature_interaction=fi, top_stack=top,
                     task=tfrs.tasks.Ranking(loss=tfrs.losses.BinaryCrossentropy(reduction="none")))
  feats = {"dense_features": torch.rand((batch, 13), generator=g, device=dev),
           "sparse_features": {str(i): torch.randint(0, vocab, (batch,), generator=g, device=dev)
                               for i in range(n_tables)}}
  labels = torch.randint(0, 2, (batch,), generator=g, device=dev)
  with torch.no_grad():
    pred = model(feats)
  model.compile(optimizer=tfrs.optimizers.Adagrad(model.parameters(), learning_rate=0.01))
  # parity outside the timed region: the loss train_step reports against float64 BCE of the model's own
  # predictions (tasks/ranking.py:92-115), predictions in (0, 1)
  logs = model.train_step((feats, labels))
  p64 = pred.double().clamp(1e-7, 1 - 1e-7)
  want = float((-(labels.double() * p64.log() + (1 - labels.double()) * (1 - p64).log())).mean())
  if not (abs(float(logs["loss"]) - want) <= 1e-5 * abs(want) and 0.0 < float(pred.min()) and float(pred.max()) < 1.0):
    raise SystemExit("bench.py: %s step: loss %r vs float64 %r" % (kind, float(logs["loss"]), want))
  step = lambda: model.train_step((feats, labels))
  ts = _event_ms(step, 5, 2)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(5):
    step()
  torch.cuda.synchronize()
  wall = (time.perf_counter() - t0) / 5 * 1e3
  f = n_tables + 1
  mlp_flop = 2.0 * batch * (13 * 512 + 512 * 256 + 256 * dim)
  if kind == "dcn_v2":
    dw = f * dim
    top_in = dim + dw
    flop = 3 * 6.0 * batch * dw * dw + 3 * (mlp_flop + 2.0 * batch * (top_in * 1024 + 1024 * 512 + 512))
    roof = _mfma_roof("3 x (tfrs_cross_fwd_f16_train + tfrs_cross_bwd_f16_saved) + MLP GEMMs (tfrs_dense_*)", flop, ts,
                      algorithmic_flop_cross_layers=3 * 6.0 * batch * dw * dw)
  else:
    od = f * (f - 1) // 2
    top_in = dim + od
    flop = 3 * (mlp_flop + 2.0 * batch * (top_in * 1024 + 1024 * 512 + 512))
    n_ids = batch * n_tables
    nbytes = (n_ids * (2 * dim * 4 + 8.0)                     # gather
              + (batch * f * dim + batch * od) * 4.0          # DotInteraction forward
              + (2.0 * batch * f * dim + batch * od) * 4      # ... backward
              + n_ids * dim * 4.0 * 5 + n_ids * 8.0)          # scatter + Adagrad (uniq ~ nnz)
    gbs = nbytes / (ts["median"] * 1e-3) / 1e9
    roof = {"kernel": "tfrs::gather_kernel + dot_interaction_fwd_pc / bwd_h16 + sort / scatter_add_u32 (Adagrad) "
                      "[HBM-bound part] + MLP GEMMs [split-fp16 MFMA part]", "bound": "hbm", "achieved": gbs,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes": nbytes, "algorithmic_flop_mlp": flop,
            "mlp_tflops_if_alone": flop / (ts["median"] * 1e-3) / 1e12, "ms_median": ts["median"],
            "note": "mixed step: the embedding + interaction kernels are HBM-bound, the top MLP (5082 -> 1024 -> 512 -> 1, "
                    "4.1 TFLOP per step) runs on the split-fp16 GEMM; bytes and flop are both over the WHOLE step time"}
  out = {"metric": "ranking train step (%s)" % kind, "value": batch / (ts["median"] * 1e-3), "unit": "examples/s",
         "ms_per_step": ts["median"], "ms_p10": ts["p10"], "ms_p90": ts["p90"], "wall_ms_per_step": wall,
         "dtype": "f32 (split-fp16 MFMA for the large GEMMs)",
         "config": {"workload": ("DCN-v2 train step: 26 x 1M x 128 tables + 13 dense, 3 Cross layers (width 3456), batch "
                                 "65536, Adagrad (BASELINE.json configs[3])" if kind == "dcn_v2" else
                                 "DLRM train step, one GPU's 1/8 row shard: 100 tables x 1.25M x 32, DotInteraction over "
                                 "101 vectors, batch 131072, Adagrad (BASELINE.json configs[4])"),
                    "tables": n_tables, "rows_per_table": vocab, "dim": dim, "batch": batch},
         "loss": float(logs["loss"]), "loss_float64_of_own_predictions": want, "roofline": roof}
  del model, emb, feats
  torch.cuda.empty_cache()
  return out


def gpu_legs(dev, ceilings=None) -> dict:
  ceilings = ceilings or measure_ceilings(dev)
  copy_gbs = ceilings["copy_gbs"]
  _MFMA_CEILING["tflops"] = ceilings["mfma_f16_tflops"]
  legs = {"cross": cross_leg(dev)}
  torch.cuda.empty_cache()
  legs["dot_interaction"] = dot_interaction_leg(dev, copy_gbs)
  torch.cuda.empty_cache()
  legs.update(embedding_legs(dev, copy_gbs))
  legs["dcn_v2_step"] = ranking_step_leg(dev, "dcn_v2")
  legs["dlrm_shard_step"] = ranking_step_leg(dev, "dlrm_shard")
  return legs


def add_cpu_baselines(legs: dict, budget_s: float = 2.5) -> None:
  """`cpu_baseline` of every leg: oracle/cpu_path.py restatements on the host cores over a bounded sample
  (a slice of the batch), scaled linearly to the leg's unit -- stated in `sample`."""
  from oracle import cpu_path   # checker-side code: only the cpu_baseline legs use it
  def rec(value, unit, r, sample):
    return {"value": value, "unit": unit, "cores": r["threads"], "kind": "port", "sample": sample}
  r = cpu_path.time_cross(1024, 3456, budget_s)
  legs["cross"]["cpu_baseline"] = rec(r["rows"] / r["seconds_per_call"], "examples/s", r,
      "forward on %d of the 65536 examples, %d calls; torch-CPU sgemm + element-wise restatement of Cross.call (not "
      "TensorFlow); per-example cost is batch independent" % (r["rows"], r["calls"]))
  rt = cpu_path.time_cross(1024, 3456, budget_s, train=True)
  legs["cross"]["training_pair"]["cpu_baseline"] = rec(rt["rows"] / rt["seconds_per_call"], "examples/s", rt,
      "forward + backward (autograd) on %d examples, %d calls" % (rt["rows"], rt["calls"]))
  r = cpu_path.time_dot_interaction(4096, 101, 32, budget_s)
  legs["dot_interaction"]["cpu_baseline"] = rec(r["rows"] / r["seconds_per_call"], "examples/s", r,
      "forward on %d of the 131072 examples, %d calls; torch-CPU bmm + lower-triangle gather (not TensorFlow)"
      % (r["rows"], r["calls"]))
  rb = cpu_path.time_dot_interaction(4096, 101, 32, budget_s, backward=True)
  legs["dot_interaction"]["backward"]["cpu_baseline"] = rec(rb["rows"] / rb["seconds_per_call"], "examples/s", rb,
      "forward + backward (autograd) on %d examples, %d calls" % (rb["rows"], rb["calls"]))
  r = cpu_path.time_segment_sum(4_000_000, 128, 65536, 8, budget_s)
  legs["segment_sum"]["cpu_baseline"] = rec(r["nnz"] / r["seconds_per_call"], "ids/s", r,
      "%d ids in bags of 8 from a 4M x 128 host table (2 GB; the 13.3 GB table of the GPU leg is not duplicated on the "
      "host), %d calls; torch embedding_bag(sum)" % (r["nnz"], r["calls"]))
  r = cpu_path.time_sparse_adagrad(4_000_000, 128, 262144, 0.5, budget_s)
  legs["sparse_adagrad"]["cpu_baseline"] = rec(r["n_ids"] / r["seconds_per_call"], "ids/s", r,
      "%d gradient rows into a 4M x 128 host table, %d calls; torch unique + index_add_ + row update" % (r["n_ids"], r["calls"]))
  r = cpu_path.time_ranking_step("dcn", 1024, 26, 100_000, 128, budget_s)
  legs["dcn_v2_step"]["cpu_baseline"] = rec(r["rows"] / r["seconds_per_call"], "examples/s", r,
      "train step on %d of the 65536 examples with 26 x 100k-row host tables, %d calls; torch-CPU autograd restatement of "
      "the DCN-v2 ranking step (not TensorFlow)" % (r["rows"], r["calls"]))
  r = cpu_path.time_ranking_step("dlrm", 2048, 100, 100_000, 32, budget_s)
  legs["dlrm_shard_step"]["cpu_baseline"] = rec(r["rows"] / r["seconds_per_call"], "examples/s", r,
      "train step on %d of the 131072 examples with 100 x 100k-row host tables, %d calls; torch-CPU autograd restatement "
      "of the DLRM ranking step (not TensorFlow)" % (r["rows"], r["calls"]))
