"""Regenerates tests/golden/*.json -- the known-answer vectors that pin the oracle.

The reference (tensorflow/recommenders) cannot be imported here: TensorFlow is
not installable in this environment.  So instead of running the reference, this
script re-evaluates, with NumPy only, the *expected-value formulas written in the
reference's own test files* on the same seeded inputs those tests use, and stores
inputs + expected outputs.  Every block cites the reference test it restates
(paths relative to /root/reference/tensorflow_recommenders).  Nothing here uses
the oracle or the HIP path -- the fixtures are independent of both.

Run:  python tests/golden/make_golden.py      (NumPy >= 1.17; output is
deterministic: RandomState / default_rng streams are stable across versions).
"""

import itertools
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def dump(name, obj):
  with open(os.path.join(HERE, name), "w") as f:
    json.dump(obj, f, separators=(",", ":"))
    f.write("\n")


def topk_grid():
  """layers/factorized_top_k_test.py:31-66 (grid) and :85-147 (run_top_k_test).

  Inputs are regenerated from the seed by the consumer (RandomState(42): candidates
  normal(N,4), query normal(Q,4), exclude randint(0,N,(Q,5)) -- in that order,
  :96-103); only the expected outputs are stored."""
  cases = []
  for k, bs, nq, nc, dtype, excl in itertools.product(
      (5, 10), (3, 16), (3, 15, 16), (1024, 128), ("str", None), (True, False)):
    rng = np.random.RandomState(42)
    candidates = rng.normal(size=(nc, 4)).astype(np.float32)
    query = rng.normal(size=(nq, 4)).astype(np.float32)
    exclude = rng.randint(0, nc, size=(nq, 5))
    scores = np.dot(query, candidates.T)                      # :105
    adjusted = scores.copy()
    if excl:                                                  # :108-114
      for r, row in enumerate(exclude):
        for c in set(row):
          adjusted[r, c] -= 1000.0
    indices = np.argsort(-adjusted, axis=1)[:, :k]            # :117
    cases.append(dict(
        k=k, batch_size=bs, num_queries=nq, num_candidates=nc,
        indices_dtype=dtype, use_exclusions=excl,
        expected_indices=indices.tolist(),
        expected_scores=np.take_along_axis(scores, indices, 1).astype(float).tolist()))
  dump("topk_grid.json", dict(
      source="layers/factorized_top_k_test.py:31-66,85-147", seed=42, dim=4,
      score_atol=1e-4, cases=cases))


def metric():
  """metrics/factorized_top_k_test.py:39-86 (weighted, score/id based) and
  :93-131 (id-based, k=100)."""
  rng = np.random.RandomState(42)
  nc, nq, d = 100, 10, 4
  candidates = rng.normal(size=(nc, d)).astype(np.float32)
  query = rng.normal(size=(nq, d)).astype(np.float32)
  sample_weight = rng.uniform(size=(nq, 1)).astype(np.float32)
  true_idx = rng.randint(0, nc, size=nq)
  scores = query @ candidates.T
  ks = [1, 5, 10, 50]
  expected = []
  for k in ks:                                                # :74-84
    t = scores[np.arange(nq), true_idx]
    in_top_k = ((scores > t[:, None]).sum(1) < k).astype(np.float32)
    expected.append(float(np.average(in_top_k, weights=sample_weight[:, 0])))
  weighted = dict(source="metrics/factorized_top_k_test.py:39-86", seed=42,
                  num_candidates=nc, num_queries=nq, dim=d, ks=ks, batch=32,
                  true_candidate_indexes=true_idx.tolist(), expected=expected,
                  rtol=1e-6)

  rng = np.random.default_rng(42)                             # :96
  k, nc, nq, d = 100, 1280, 128, 128
  candidates = rng.normal(size=(nc, d)).astype(np.float32)
  queries = rng.normal(size=(nq, d)).astype(np.float32)
  true_idx = rng.integers(0, nc, size=nq).astype(np.int32)
  s = queries.astype(np.float64) @ candidates.astype(np.float64).T
  top = np.argsort(-s, axis=1, kind="stable")[:, :k]
  found = [(int(t) in row.tolist()) for t, row in zip(true_idx, top)]
  # margin between the 100th and 101st score, to show the expectation does not
  # hinge on float32 rounding of the scores
  srt = -np.sort(-s, axis=1)
  idbased = dict(source="metrics/factorized_top_k_test.py:93-131", seed=42, k=k,
                 num_candidates=nc, num_queries=nq, dim=d, batch=32,
                 true_candidate_indices=true_idx.tolist(),
                 expected_metric=float(np.mean(found)),
                 min_rank_margin=float((srt[:, k - 1] - srt[:, k]).min()))
  dump("metric.json", dict(weighted=weighted, id_based=idbased))


def _sigmoid(x):
  return 1.0 / (1 + np.exp(-x))


def retrieval():
  """tasks/retrieval_test.py:33-71,112-137 (2x2), :181-213 (extra negatives),
  :257-298 (multi-head maxsim)."""
  dump("retrieval.json", dict(cases=[
      dict(source="tasks/retrieval_test.py:33-71",
           query=[[1, 2, 3], [2, 3, 4]], candidate=[[1, 1, 1], [1, 1, 0]],
           corpus_zeros=[20, 3], corpus_batch=16, ks=[5], sample_weight=None,
           expected_loss=float(-np.log(_sigmoid(3.0)) - np.log(1 - _sigmoid(4.0))),
           expected_top5=1.0, expected_batch_top1=0.5),
      dict(source="tasks/retrieval_test.py:112-137",
           query=[[1, 2, 3], [2, 3, 4]], candidate=[[1, 1, 1], [1, 1, 0]],
           corpus_zeros=[20, 3], corpus_batch=16, ks=[5], sample_weight=[0.7, 0.3],
           expected_loss=float(-0.7 * np.log(_sigmoid(3.0))
                               - 0.3 * np.log(1 - _sigmoid(4.0))),
           expected_top5=1.0, expected_batch_top1=0.7),
      dict(source="tasks/retrieval_test.py:181-213",
           query=[[3, 2, 1], [2, 3, 4]],
           candidate=[[0, 1, 0], [0, 1, 1], [1, 1, 0]],
           corpus_zeros=[20, 3], corpus_batch=16, ks=[5], sample_weight=None,
           expected_loss=float(-np.log(1 / (1 + np.exp(1) + np.exp(3)))
                               - np.log(np.exp(4) / (1 + np.exp(4) + np.exp(2)))),
           expected_top5=1.0, expected_batch_top1=0.5),
      dict(source="tasks/retrieval_test.py:257-298",
           query=[[[3, 2, 1], [1, 2, 3]], [[2, 3, 4], [4, 3, 2]]],
           candidate=[[0, 1, 0], [0, 1, 1], [1, 1, 0]],
           corpus_zeros=[20, 3], corpus_batch=16, ks=[5], sample_weight=None,
           expected_loss=float(-np.log(1 / (1 + np.exp(3) + np.exp(3)))
                               - np.log(np.exp(5) / (np.exp(1) + np.exp(5) + np.exp(5)))),
           expected_top5=0.0, expected_batch_top1=0.5),
  ], loss_layer_seeds=[42, 123, 8391, 12390, 1230],
      loss_layer_source="layers/loss_test.py:29-130", rtol=1e-6))


def feature_interaction():
  """dcn_test.py:29-50,68-101; multi_layer_dcn_test.py:28-60;
  dot_interaction_test.py:25-64."""
  f1 = np.asarray([0.1, -4.3, 0.2, 1.1, 0.3], np.float32)
  f2 = np.asarray([2.0, 3.2, -1.0, 0.0, 1.0], np.float32)
  f3 = np.asarray([0.0, 1.0, -3.0, -2.2, -0.2], np.float32)
  f11, f12, f13 = float(f1 @ f1), float(f1 @ f2), float(f1 @ f3)
  f22, f23, f33 = float(f2 @ f2), float(f2 @ f3), float(f3 @ f3)
  dump("feature_interaction.json", dict(
      cross=[
          dict(source="dcn_test.py:29-35", x0=[[0.1, 0.2, 0.3]], x=[[0.4, 0.5, 0.6]],
               projection_dim=None, kernel="ones", bias="zeros", diag_scale=0.0,
               expected=[[0.55, 0.8, 1.05]]),
          dict(source="dcn_test.py:37-43", x0=[[0.1, 0.2, 0.3]], x=[[0.4, 0.5, 0.6]],
               projection_dim=1, kernel="ones", bias="zeros", diag_scale=0.0,
               expected=[[0.55, 0.8, 1.05]]),
          dict(source="dcn_test.py:45-50", x0=[[0.1, 0.2, 0.3]], x=None,
               projection_dim=None, kernel="ones", bias="zeros", diag_scale=0.0,
               expected=[[0.16, 0.32, 0.48]]),
          dict(source="dcn_test.py:68-75", x0=[[0.1, 0.2, 0.3]], x=[[0.4, 0.5, 0.6]],
               projection_dim=None, kernel="ones", bias="ones", diag_scale=0.0,
               expected=[[0.65, 1.0, 1.35]]),
          dict(source="dcn_test.py:83-90", x0=[[0.1, 0.2, 0.3]], x=[[0.4, 0.5, 0.6]],
               projection_dim=None, kernel="ones", bias="zeros", diag_scale=1.0,
               expected=[[0.59, 0.9, 1.23]]),
          dict(source="dcn_test.py:92-101", x0=[[0.1, 0.2, 0.3]], x=[[0.4, 0.5, 0.6]],
               projection_dim=None, kernel="truncated_normal", bias="zeros",
               diag_scale=0.0, preactivation="zeros_like",
               expected=[[0.4, 0.5, 0.6]]),
      ],
      multi_layer_dcn=[
          dict(source="multi_layer_dcn_test.py:28-38", x0=[[0.1, 0.2, 0.3]],
               projection_dim=3, num_layers=1, use_bias=False, kernel="ones",
               bias="zeros", expected=[[0.28, 0.56, 0.84]]),
          dict(source="multi_layer_dcn_test.py:40-50", x0=[[0.1, 0.2, 0.3]],
               projection_dim=1, num_layers=1, use_bias=False, kernel="ones",
               bias="zeros", expected=[[0.16, 0.32, 0.48]]),
          dict(source="multi_layer_dcn_test.py:52-59", x0=[[0.1, 0.2, 0.3]],
               projection_dim=1, num_layers=3, use_bias=True, kernel="ones",
               bias="ones", expected=[[0.9256, 1.8512, 2.7768]]),
      ],
      dot_interaction=dict(
          source="dot_interaction_test.py:25-64",
          features=[[f1.tolist()], [f2.tolist()], [f3.tolist()]],
          expected={
              "self_gather": [[f11, f12, f22, f13, f23, f33]],
              "self_skip": [[f11, 0, 0, f12, f22, 0, f13, f23, f33]],
              "noself_gather": [[f12, f13, f23]],
              "noself_skip": [[0, 0, 0, f12, 0, 0, f13, f23, 0]],
          }),
      rtol=1e-6, atol=1e-6))


def embedding():
  """Derived KAT (NOT asserted by the reference): fixture at
  layers/embedding/tpu_embedding_layer_test.py:51-111; expected values follow from
  the combiner definitions (sum / mean) -- SURVEY.md Appendix B last row."""
  dump("embedding.json", dict(
      source="tpu_embedding_layer_test.py:51-111 (fixture only; parity unpinned)",
      video_table=[[0, 1, 2, 3], [4, 5, 6, 7]], video_combiner="sum",
      user_table=[[0, 1], [2, 3], [4, 5], [6, 7]], user_combiner="mean",
      watched=dict(ids=[0, 0, 1, 0, 1, 1], row_splits=[0, 1, 3, 5, 6],
                   expected=[[0, 1, 2, 3], [4, 6, 8, 10], [4, 6, 8, 10], [4, 5, 6, 7]]),
      favorited=dict(ids=[0, 1, 1, 0, 0, 1], row_splits=[0, 2, 3, 4, 6],
                     expected=[[4, 6, 8, 10], [4, 5, 6, 7], [0, 1, 2, 3], [4, 6, 8, 10]]),
      friends=dict(ids=[3, 0, 1, 2, 3, 0, 1, 2], row_splits=[0, 1, 4, 5, 8],
                   expected=[[6, 7], [2, 3], [6, 7], [2, 3]])))


def hashing():
  """SipHash-2-4 known answers (Aumasson & Bernstein, "SipHash: a fast short-input PRF",
  Appendix A and the 64-entry vector table of the public reference implementation):
  key = bytes 00..0f, message = bytes 00..len-1, output as a little-endian 64-bit word.
  These pin the hash under ``Hashing(salt=...)`` of unified_embedding.py:155-159; the
  TensorFlow glue around it (decimal strings, modulo) is not asserted by the reference's
  tests (unified_embedding_test.py:73-150 check shapes only) and stays unpinned."""
  dump("hashing.json", dict(
      source="SipHash-2-4 reference vectors; key 000102..0f, message 00..len-1",
      key=[0x0706050403020100, 0x0F0E0D0C0B0A0908],
      vectors=[dict(len=n, hash=h) for n, h in (
          (0, "726fdb47dd0e0e31"), (1, "74f839c593dc67fd"), (2, "0d6c8009d9a94f5a"),
          (3, "85676696d7fb7e2d"), (4, "cf2794e0277187b7"), (5, "18765564cd99a68d"),
          (15, "a129ca6149be45e5"), (63, "958a324ceb064572"))]))


if __name__ == "__main__":
  hashing()
  topk_grid()
  metric()
  retrieval()
  feature_interaction()
  embedding()
  print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".json")))
