"""CPU reference path for timing (test infrastructure only; used by bench.py's
``cpu_baseline`` leg).

TensorFlow cannot be installed here, so the "reference CPU path" is this restatement of
``BruteForce.call`` (``layers/factorized_top_k.py:586-607``) executed with torch-CPU:
oneDNN/MKL ``sgemm`` for ``tf.matmul`` (:333/:603) and ``torch.topk`` for
``tf.math.top_k`` (:605) -- the same class of multithreaded kernels TF-CPU dispatches
to.  Queries are processed in blocks so the ``[block, N]`` score matrix fits in RAM.
Labelled "CPU reference restatement (not TensorFlow)" wherever it is reported.
"""

import time
from typing import Tuple

import numpy as np
import torch


def brute_force_topk(queries: np.ndarray, candidates: np.ndarray, k: int,
                     block: int = 256) -> Tuple[np.ndarray, np.ndarray]:
  q = torch.from_numpy(np.ascontiguousarray(queries, dtype=np.float32))
  c = torch.from_numpy(np.ascontiguousarray(candidates, dtype=np.float32))
  vals, idx = [], []
  with torch.no_grad():
    for lo in range(0, q.shape[0], block):
      s = q[lo:lo + block] @ c.t()            # :603 (scores :333)
      v, i = torch.topk(s, k, dim=1)          # :605
      vals.append(v)
      idx.append(i)
  return torch.cat(vals).numpy(), torch.cat(idx).numpy()


def time_brute_force(candidates: np.ndarray, queries: np.ndarray, k: int,
                     budget_s: float = 15.0, block: int = 256) -> dict:
  """Times the CPU path on as many query blocks as fit in ``budget_s`` seconds (at
  least one block after one warm-up block); returns queries/s and what was sampled."""
  q = torch.from_numpy(np.ascontiguousarray(queries, dtype=np.float32))
  c = torch.from_numpy(np.ascontiguousarray(candidates, dtype=np.float32))
  ct = c.t()
  with torch.no_grad():
    torch.topk(q[:block] @ ct, k, dim=1)      # warm-up (thread pool, page faults)
    done, t0 = 0, time.perf_counter()
    while True:
      lo = done % max(q.shape[0] - block + 1, 1)
      torch.topk(q[lo:lo + block] @ ct, k, dim=1)
      done += min(block, q.shape[0])
      dt = time.perf_counter() - t0
      if dt >= budget_s or done >= 16 * block:
        break
  return {"value": done / dt, "seconds": dt, "queries": done,
          "threads": torch.get_num_threads()}


def time_train_step(batch: int = 4096, dim: int = 64, vocab: int = 2000, users: int = 943,
                    items: int = 1682, lr: float = 0.5, budget_s: float = 5.0, seed: int = 0,
                    with_metrics: bool = False, ks=(1, 5, 10, 50, 100), cand_batch: int = 128) -> dict:
  """CPU restatement of the quickstart two-tower train step (reference ``README.md:58-97``,
  ``models/base.py:64-85``): two ``Embedding`` lookups -> ``Retrieval`` loss
  (``tasks/retrieval.py:172-210``: scores = q @ c^T, labels = eye, Keras
  ``CategoricalCrossentropy(from_logits=True, reduction=SUM)``) -> backward -> Keras Adagrad
  (``acc += g*g ; var -= lr * g / sqrt(acc + eps)``, dense here: at 2k-row tables TF-CPU's
  sparse apply and the dense apply touch the same order of bytes).  torch-CPU autograd =
  oneDNN sgemm + vectorised elementwise, all host threads; "CPU reference restatement (not
  TensorFlow)".

  ``with_metrics`` adds what the reference's quickstart step also does on every call
  (``README.md:69-80`` builds ``Retrieval(metrics=FactorizedTopK(movies.batch(128).map(item_model)))``
  and calls the task with ``compute_metrics=True``, ``tasks/retrieval.py:216-226``):
  ``FactorizedTopK.update_state`` (``metrics/factorized_top_k.py:91-194``) over the ``items``
  candidates re-embedded in blocks of ``cand_batch`` -- the ``Streaming`` fold of
  ``layers/factorized_top_k.py:420-507`` (per block scores + top-k, concat with the state, top-k)
  and ``in_top_k`` on ``concat([positive, top_k])`` for every k, then the running means."""
  g = torch.Generator().manual_seed(seed)
  tables = [torch.empty(vocab, dim).uniform_(-0.05, 0.05, generator=g).requires_grad_(True)
            for _ in range(2)]
  accs = [torch.full((vocab, dim), 0.1) for _ in range(2)]
  uid = torch.randint(0, users, (batch,), generator=g)
  iid = torch.randint(0, items, (batch,), generator=g)
  labels = torch.arange(batch)
  movie_ids = torch.arange(items)
  kmax = max(ks)
  totals = torch.zeros(len(ks))
  counts = torch.zeros(len(ks))

  def update_metrics(q, c):
    with torch.no_grad():
      pos = (q * c).sum(dim=1, keepdim=True)                          # :133-134
      state = None
      for lo in range(0, items, cand_batch):                           # Streaming.call :420-507
        block = torch.nn.functional.embedding(movie_ids[lo:lo + cand_batch], tables[1])
        v = torch.topk(q @ block.t(), min(kmax, block.shape[0]), dim=1).values
        state = v if state is None else torch.topk(torch.cat([state, v], dim=1),
                                                   min(kmax, state.shape[1] + v.shape[1]), dim=1).values
      pred = torch.cat([pos, state], dim=1)                            # :183
      greater = (pred > pos).sum(dim=1)                                # in_top_k(target 0) :187-190
      for i, k in enumerate(ks):
        hit = ((greater < k) & torch.isfinite(pos[:, 0])).float()
        totals[i] += hit.sum()
        counts[i] += hit.numel()

  def step():
    q = torch.nn.functional.embedding(uid, tables[0])
    c = torch.nn.functional.embedding(iid, tables[1])
    loss = torch.nn.functional.cross_entropy(q @ c.t(), labels, reduction="sum")
    if with_metrics:
      update_metrics(q.detach(), c.detach())
    grads = torch.autograd.grad(loss, tables)
    with torch.no_grad():
      for t, a, gr in zip(tables, accs, grads):
        a.addcmul_(gr, gr)
        t.addcdiv_(gr, torch.sqrt(a + 1e-7), value=-lr)
    return float(loss.detach())

  step()
  n, t0 = 0, time.perf_counter()
  while True:
    step()
    n += 1
    dt = time.perf_counter() - t0
    if dt >= budget_s or n >= 200:
      break
  return {"value": n / dt, "seconds": dt, "steps": n, "threads": torch.get_num_threads()}


# ------------------------------------------------------------------------------------------------
# CPU legs of bench.py's configs[3] / configs[4] blocks (round 5).  Same rules as above: torch-CPU
# restatements of the reference formulas on ALL host threads, on a bounded SAMPLE of the workload
# (a slice of the batch; the per-example cost of every one of these ops does not depend on the
# batch size), reported per unit of work so that bench.py can scale to the full batch and say so.
# ------------------------------------------------------------------------------------------------
def _timed(fn, budget_s: float, max_iters: int = 50):
  fn()                                   # warm-up (thread pool, page faults of freshly allocated tables:
  fn()                                   # the first two calls of the update legs run 5-20x slower)
  n, t0 = 0, time.perf_counter()
  while True:
    fn()
    n += 1
    dt = time.perf_counter() - t0
    if dt >= budget_s or n >= max_iters:
      return dt / n, n


def time_cross(rows: int, d: int, budget_s: float = 3.0, train: bool = False) -> dict:
  """``Cross.call`` (dcn.py:151-186) ``y = x0 * (x @ W + b) + x`` on ``rows`` examples of width ``d``
  (forward; ``train``: forward + backward through torch autograd = the reference's three sgemms + the
  element-wise passes)."""
  g = torch.Generator().manual_seed(0)
  x0 = torch.randn((rows, d), generator=g)
  x = torch.randn((rows, d), generator=g).requires_grad_(train)
  w = (torch.randn((d, d), generator=g) * 0.05).requires_grad_(train)
  b = torch.zeros((d,)).requires_grad_(train)
  dy = torch.randn((rows, d), generator=g)

  def fwd():
    with torch.no_grad():
      return x0 * (x @ w + b) + x

  def pair():
    y = x0 * (x @ w + b) + x
    torch.autograd.grad(y, (x, w, b), grad_outputs=dy)

  sec, n = _timed(pair if train else fwd, budget_s)
  return {"seconds_per_call": sec, "rows": rows, "calls": n, "threads": torch.get_num_threads()}


def time_dot_interaction(rows: int, f: int, d: int, budget_s: float = 3.0, backward: bool = False) -> dict:
  """``DotInteraction.call`` (dot_interaction.py:69-104): batched Gram ``X X^T`` + the row-major strict
  lower triangle (``boolean_mask``), optionally with its backward."""
  g = torch.Generator().manual_seed(0)
  x = torch.randn((rows, f, d), generator=g).requires_grad_(backward)
  ii, jj = torch.tril_indices(f, f, -1)
  dy = torch.randn((rows, ii.numel()), generator=g)

  def fwd():
    with torch.no_grad():
      return torch.bmm(x, x.transpose(1, 2))[:, ii, jj]

  def bwd():
    out = torch.bmm(x, x.transpose(1, 2))[:, ii, jj]
    torch.autograd.grad(out, x, grad_outputs=dy)

  sec, n = _timed(bwd if backward else fwd, budget_s)
  return {"seconds_per_call": sec, "rows": rows, "calls": n, "threads": torch.get_num_threads()}


def time_segment_sum(vocab: int, d: int, bags: int, bag: int, budget_s: float = 3.0) -> dict:
  """Sum-combiner lookup of ``bags`` bags of ``bag`` ids (tpu_embedding_layer.py:913-919 CPU branch):
  ``torch.nn.functional.embedding_bag(mode="sum")``, the multithreaded gather + segment-sum TF-CPU's
  ``embedding_lookup_sparse`` amounts to."""
  g = torch.Generator().manual_seed(0)
  table = torch.empty((vocab, d)).uniform_(-0.05, 0.05, generator=g)
  ids = torch.randint(0, vocab, (bags * bag,), generator=g)
  offs = torch.arange(0, bags * bag, bag)
  sec, n = _timed(lambda: torch.nn.functional.embedding_bag(ids, table, offs, mode="sum"), budget_s)
  return {"seconds_per_call": sec, "nnz": bags * bag, "vocab": vocab, "calls": n,
          "threads": torch.get_num_threads()}


def time_sparse_adagrad(vocab: int, d: int, n_ids: int, lr: float = 0.5, budget_s: float = 3.0) -> dict:
  """Keras Adagrad on an ``IndexedSlices`` gradient (README.md:84, models/base.py:77-78): duplicate ids
  summed (``unique`` + ``index_add_``), then ``acc += g^2; row -= lr g / sqrt(acc + eps)`` on the touched
  rows only."""
  g = torch.Generator().manual_seed(0)
  table = torch.empty((vocab, d)).uniform_(-0.05, 0.05, generator=g)
  acc = torch.full((vocab, d), 0.1)
  ids = torch.randint(0, vocab, (n_ids,), generator=g)
  rows = torch.randn((n_ids, d), generator=g)

  def step():
    uniq, inv = torch.unique(ids, return_inverse=True)
    gsum = torch.zeros((uniq.numel(), d)).index_add_(0, inv, rows)
    a = acc[uniq] + gsum * gsum
    acc[uniq] = a
    table[uniq] -= lr * gsum / torch.sqrt(a + 1e-7)

  sec, n = _timed(step, budget_s)
  return {"seconds_per_call": sec, "n_ids": n_ids, "vocab": vocab, "calls": n,
          "threads": torch.get_num_threads()}


def time_ranking_step(kind: str, rows: int, n_tables: int, vocab: int, dim: int, budget_s: float = 4.0) -> dict:
  """One train step of ``experimental/models/ranking.py:135-236`` on ``rows`` examples, torch-CPU autograd:
  per-feature embedding lookups (sparse gradients), bottom MLP [512, 256, dim] on 13 dense features,
  ``kind`` = "dcn" (3 full-rank Cross layers on the concatenation) or "dlrm" (DotInteraction + concat with
  the bottom output), top MLP [1024, 512, 1] + sigmoid, mean binary cross-entropy, Adagrad (sparse on the
  tables).  ``vocab`` may be smaller than the configuration's (host RAM); say so where it is reported."""
  g = torch.Generator().manual_seed(0)
  emb = torch.nn.Embedding(n_tables * vocab, dim, sparse=True)
  f = n_tables + 1
  width = dim + (f * dim if kind == "dcn" else f * (f - 1) // 2)

  def mlp(sizes, final):
    layers = []
    for a, b in zip(sizes[:-1], sizes[1:]):
      layers += [torch.nn.Linear(a, b), torch.nn.ReLU()]
    layers[-1] = final
    return torch.nn.Sequential(*layers)

  bottom = mlp([13, 512, 256, dim], torch.nn.ReLU())
  top = mlp([width, 1024, 512, 1], torch.nn.Sigmoid())
  cross_w = [torch.nn.Parameter(torch.randn((f * dim, f * dim), generator=g) * 0.01) for _ in range(3)] \
      if kind == "dcn" else []
  cross_b = [torch.nn.Parameter(torch.zeros((f * dim,))) for _ in cross_w]
  dense_params = list(bottom.parameters()) + list(top.parameters()) + cross_w + cross_b
  opt_d = torch.optim.Adagrad(dense_params, lr=0.01, initial_accumulator_value=0.1, eps=1e-7)
  opt_s = torch.optim.Adagrad(emb.parameters(), lr=0.01, initial_accumulator_value=0.1, eps=1e-7)
  dense = torch.rand((rows, 13), generator=g)
  ids = torch.randint(0, vocab, (rows, n_tables), generator=g) + torch.arange(n_tables) * vocab
  labels = torch.randint(0, 2, (rows,), generator=g).float()
  ii, jj = torch.tril_indices(f, f, -1)

  def step():
    opt_d.zero_grad()
    opt_s.zero_grad()
    dv = bottom(dense)
    x = torch.cat([emb(ids), dv[:, None, :]], dim=1)              # [rows, F + 1, dim]
    if kind == "dcn":
      x0 = x.reshape(rows, -1)
      xl = x0
      for w, b in zip(cross_w, cross_b):
        xl = x0 * (xl @ w + b) + xl
      inter = xl
    else:
      inter = torch.bmm(x, x.transpose(1, 2))[:, ii, jj]
    p = top(torch.cat([dv, inter], dim=1)).reshape(-1)
    loss = torch.nn.functional.binary_cross_entropy(p, labels)
    loss.backward()
    opt_d.step()
    opt_s.step()

  sec, n = _timed(step, budget_s, max_iters=20)
  return {"seconds_per_call": sec, "rows": rows, "vocab": vocab, "calls": n, "threads": torch.get_num_threads()}
