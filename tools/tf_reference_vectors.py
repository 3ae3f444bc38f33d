#!/usr/bin/env python
"""Reference-side vectors for the three numerics the reference's own tests leave unpinned (VERDICT
round 5, missing 3): the embedding combiners, the Adagrad update on IndexedSlices with duplicate ids, and one
README-quickstart train step.  Needs TensorFlow (>= 2.9) -- NOT installable in the build container, so this script is
the hand-over: a maintainer with TensorFlow runs

    pip install tensorflow tf-keras        # TF >= 2.16 needs tf-keras and TF_USE_LEGACY_KERAS=1, as the reference's
    TF_USE_LEGACY_KERAS=1 python tools/tf_reference_vectors.py      # tools/build_scripts/pip_install.sh does

and commits the three files it writes under tests/golden/ (tf_combiners.json, tf_adagrad.json, tf_train_step.json).
`pytest tests/test_tf_vectors.py` then holds the oracle (CPU) and the HIP kernels (GPU) to TensorFlow's own numbers,
and the "parity unpinned" notes of oracle/embedding.py, DESIGN.md section 2 and INTEGRATION.md can be dropped.  The
inputs are generated HERE from fixed seeds with NumPy only and stored in the files: the tests never need TensorFlow.

What is recorded, and the reference code it pins:
  * combiners  tf.nn.embedding_lookup_sparse(params, sp_ids, sp_weights, combiner) for sum / mean / sqrtn, with and
               without weights, ragged rows incl. an empty one -- the arithmetic behind TPUEmbedding's CPU branch
               (layers/embedding/tpu_embedding_layer.py:913-919 via tf.tpu.experimental.embedding.serving_embedding_lookup);
  * adagrad    three steps of tf.keras.optimizers.Adagrad(0.5) on a [50, 8] table with an IndexedSlices gradient that
               repeats ids (README.md:84 under models/base.py:77-78), for the optimizer class the installed TF gives
               (`formula`: "sqrt(acc+eps)" for the TF >= 2.11 / tf-keras optimizer, "sqrt(acc)+eps" for optimizer_v2),
               and, where available, for tf.keras.optimizers.legacy.Adagrad as well;
  * train step one tfrs-style two-tower step WITHOUT tensorflow_recommenders installed: embeddings -> in-batch
               softmax (tasks/retrieval.py:172-210 restated with tf ops: eye labels, CategoricalCrossentropy(from_logits,
               SUM)) -> tape.gradient -> Adagrad.apply_gradients; records loss, both tables and accumulators after
               each of three steps.  With tensorflow_recommenders importable the loss is cross-checked against
               tfrs.tasks.Retrieval()(q, c) and the file says so (`checked_against_tfrs`).
"""

import json
import os
import sys

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _l(a):
  return np.asarray(a).tolist()


def combiner_inputs():
  rng = np.random.default_rng(2024)
  table = rng.uniform(-0.5, 0.5, size=(40, 6)).astype(np.float32)
  row_splits = np.array([0, 3, 3, 4, 9, 12], dtype=np.int64)        # five rows, the second one empty
  ids = rng.integers(0, 40, size=12).astype(np.int64)
  ids[5] = ids[4]                                                     # a repeated id inside one row
  weights = rng.uniform(0.25, 2.0, size=12).astype(np.float32)
  return table, ids, row_splits, weights


def adagrad_inputs():
  rng = np.random.default_rng(2025)
  table = rng.uniform(-0.05, 0.05, size=(50, 8)).astype(np.float32)
  steps = []
  for _ in range(3):
    ids = rng.integers(0, 50, size=24).astype(np.int64)
    ids[3] = ids[0]; ids[7] = ids[0]; ids[11] = ids[10]                # duplicates: summed before the update
    grads = rng.normal(size=(24, 8)).astype(np.float32)
    steps.append((ids, grads))
  return table, steps


def train_step_inputs():
  rng = np.random.default_rng(2026)
  users = rng.uniform(-0.05, 0.05, size=(30, 16)).astype(np.float32)
  items = rng.uniform(-0.05, 0.05, size=(40, 16)).astype(np.float32)
  batches = [(rng.integers(0, 30, size=32).astype(np.int64), rng.integers(0, 40, size=32).astype(np.int64))
             for _ in range(3)]
  return users, items, batches


def main() -> None:
  os.environ.setdefault("TF_USE_LEGACY_KERAS", "1")
  import tensorflow as tf
  os.makedirs(OUT, exist_ok=True)
  meta = {"tensorflow": tf.__version__, "keras": getattr(tf.keras, "__version__", "?"),
          "generated_by": "tools/tf_reference_vectors.py"}

  # ---- combiners
  table, ids, row_splits, weights = combiner_inputs()
  nrows = len(row_splits) - 1
  seg = np.repeat(np.arange(nrows), np.diff(row_splits))
  pos = np.concatenate([np.arange(n) for n in np.diff(row_splits)]) if len(ids) else np.zeros((0,), np.int64)
  indices = np.stack([seg, pos], axis=1).astype(np.int64)
  dense_shape = [nrows, int(np.diff(row_splits).max())]
  sp_ids = tf.sparse.SparseTensor(indices, ids, dense_shape)
  sp_w = tf.sparse.SparseTensor(indices, weights, dense_shape)
  comb = {"table": _l(table), "ids": _l(ids), "row_splits": _l(row_splits), "weights": _l(weights), "results": {}}
  for combiner in ("sum", "mean", "sqrtn"):
    for name, w in (("unweighted", None), ("weighted", sp_w)):
      out = tf.nn.embedding_lookup_sparse(tf.constant(table), sp_ids, w, combiner=combiner)
      out = out.numpy()
      if out.shape[0] < nrows:                                         # (old TF drops trailing empty rows)
        out = np.concatenate([out, np.zeros((nrows - out.shape[0], out.shape[1]), np.float32)])
      comb["results"][f"{combiner}/{name}"] = _l(out)
  with open(os.path.join(OUT, "tf_combiners.json"), "w") as f:
    json.dump({"meta": meta, **comb}, f)

  # ---- Adagrad on IndexedSlices with duplicate ids
  table0, steps = adagrad_inputs()

  def run_adagrad(opt_cls):
    var = tf.Variable(table0)
    opt = opt_cls(learning_rate=0.5)
    outs = []
    for ids_t, grads_t in steps:
      slices = tf.IndexedSlices(tf.constant(grads_t), tf.constant(ids_t), dense_shape=tf.constant(table0.shape, tf.int64))
      opt.apply_gradients([(slices, var)])
      outs.append(_l(var.numpy()))
    return outs

  ada = {"table": _l(table0), "learning_rate": 0.5, "initial_accumulator_value": 0.1, "epsilon": 1e-7,
         "steps": [{"ids": _l(i), "grads": _l(g)} for i, g in steps], "variants": {}}
  default_cls = tf.keras.optimizers.Adagrad
  is_v2 = any(c.__name__ == "OptimizerV2" for c in default_cls.__mro__)
  ada["variants"]["default"] = {"class": default_cls.__module__ + "." + default_cls.__name__,
                                "formula": "sqrt(acc)+eps" if is_v2 else "sqrt(acc+eps)", "tables": run_adagrad(default_cls)}
  legacy = getattr(getattr(tf.keras.optimizers, "legacy", None), "Adagrad", None)
  if legacy is not None and legacy is not default_cls:
    ada["variants"]["legacy"] = {"class": legacy.__module__ + "." + legacy.__name__, "formula": "sqrt(acc)+eps",
                                 "tables": run_adagrad(legacy)}
  with open(os.path.join(OUT, "tf_adagrad.json"), "w") as f:
    json.dump({"meta": meta, **ada}, f)

  # ---- one two-tower train step, three times
  users0, items0, batches = train_step_inputs()
  u, v = tf.Variable(users0), tf.Variable(items0)
  opt = tf.keras.optimizers.Adagrad(0.5)
  loss_fn = tf.keras.losses.CategoricalCrossentropy(from_logits=True, reduction=tf.keras.losses.Reduction.SUM)
  try:
    import tensorflow_recommenders as tfrs
    task = tfrs.tasks.Retrieval()
  except Exception:     # noqa: BLE001 -- the restatement below stands on its own
    task = None
  rec = {"users": _l(users0), "items": _l(items0), "learning_rate": 0.5, "steps": [],
         "optimizer_formula": ada["variants"]["default"]["formula"], "checked_against_tfrs": task is not None}
  for uid, iid in batches:
    with tf.GradientTape() as tape:
      q, c = tf.gather(u, uid), tf.gather(v, iid)
      scores = tf.linalg.matmul(q, c, transpose_b=True)                 # tasks/retrieval.py:172-180
      labels = tf.eye(tf.shape(scores)[0], tf.shape(scores)[1])         # :185
      loss = loss_fn(labels, scores)                                    # :86-87, :205-210
    if task is not None:
      want = float(task(q, c, compute_metrics=False))
      assert abs(float(loss) - want) <= 1e-5 * abs(want), (float(loss), want)
    grads = tape.gradient(loss, [u, v])                                 # models/base.py:77
    opt.apply_gradients(zip(grads, [u, v]))                             # :78
    rec["steps"].append({"user_ids": _l(uid), "item_ids": _l(iid), "loss": float(loss), "users": _l(u.numpy()),
                         "items": _l(v.numpy())})
  with open(os.path.join(OUT, "tf_train_step.json"), "w") as f:
    json.dump({"meta": meta, **rec}, f)
  print("wrote tf_combiners.json, tf_adagrad.json, tf_train_step.json under", OUT)


if __name__ == "__main__":
  if "--inputs-only" in sys.argv:       # what the container CAN do: show that the seeded inputs are reproducible
    t, i, r, w = combiner_inputs()
    print(json.dumps({"combiner_ids": _l(i), "adagrad_first_ids": _l(adagrad_inputs()[1][0][0])}))
  else:
    main()
