"""CPU oracle for the tensorflow/recommenders retrieval hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``recommenders_amd/`` imports this
package; only ``tests/``, ``bench.py``'s ``cpu_baseline`` leg and
``__graft_entry__.smoke()`` may, and only as the checker.

What it is: a restatement, in NumPy plus a small C library (``oracle/c``), of the
algorithms in the reference's Python files, each function citing the
``/root/reference/tensorflow_recommenders`` file:line it follows.  The arithmetic
itself lives in TensorFlow / tf-keras (``tensorflow>=2.9.0``, ``tf-keras``;
``requirements.txt:1-4``), which is not vendored in the reference and cannot be
installed here, so the TF op semantics used (``tf.math.top_k`` tie order,
``tf.math.in_top_k``, Keras ``CategoricalCrossentropy(from_logits, SUM)``,
Keras ``Dense``) are restated from their published behaviour.

How it is pinned: ``tests/test_oracle_golden.py`` checks every function here
against the known-answer vectors the reference's own tests hold
(``tests/golden/*.json``, produced by ``tests/golden/make_golden.py`` from the
reference test files).  Pieces the reference's tests do not pin numerically
(embedding combiners, Adagrad) say "parity unpinned" in their docstrings.
"""

from oracle import embedding, feature_interaction, metrics, retrieval, topk  # noqa: F401
