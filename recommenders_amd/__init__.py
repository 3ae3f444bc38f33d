"""recommenders_amd: the tensorflow/recommenders retrieval hot path on AMD MI355X.

Same class names and call signatures as ``tensorflow_recommenders`` for the path
that matters -- ``layers.factorized_top_k.{TopK, BruteForce, Streaming}``,
``metrics.FactorizedTopK``, ``tasks.Retrieval``,
``layers.feature_interaction.{Cross, DotInteraction, MultiLayerDCN}``,
``layers.embedding.Embedding`` and ``models.Model`` -- on ``torch`` tensors that
live in HBM.  All arithmetic runs in hand-written gfx950 HIP kernels behind the C ABI
of ``include/tfrs_hip.h`` (``libtfrs_hip.so``); torch only carries device memory,
streams and autograd bookkeeping.  There is no CPU fallback.
"""

from recommenders_amd import data, experimental, layers, losses, metrics, models, optimizers, tasks  # noqa: F401
from recommenders_amd.models import Model  # noqa: F401

__version__ = "0.1.0"
