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
