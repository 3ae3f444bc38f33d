"""Kernel-trace target: the C1 two-tower tfrs.Model.train_step replayed from a HIP graph
(bench.py train_step_metric).  rocprofv3 --kernel-trace --stats shows the per-kernel GPU time
that bounds the replayed step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recommenders_amd as tfrs
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
B, D, V, ITEMS = 4096, 64, 2000, 1682
METRICS = os.environ.get("TFRS_EXP_METRICS", "1") == "1"   # README step: FactorizedTopK updated every step


class TwoTower(tfrs.Model):
  def __init__(self):
    super().__init__()
    self.user_model = tfrs.layers.embedding.Embedding(V, D)
    self.item_model = tfrs.layers.embedding.Embedding(V, D)
    if METRICS:
      movies = tfrs.data.Dataset.from_tensor_slices(torch.arange(ITEMS, device=dev))
      self.task = tfrs.tasks.Retrieval(metrics=tfrs.metrics.FactorizedTopK(
          candidates=movies.batch(128).map(self.item_model)))
    else:
      self.task = tfrs.tasks.Retrieval()

  def compute_loss(self, inputs, training=False):
    return self.task(self.user_model(inputs["user_id"]), self.item_model(inputs["movie_id"]),
                     compute_metrics=METRICS)


model = TwoTower()
model.compile(optimizer=tfrs.optimizers.Adagrad(model.parameters(), learning_rate=0.5))
batch = {"user_id": torch.randint(0, 943, (B,), generator=g, device=dev),
         "movie_id": torch.randint(0, 1682, (B,), generator=g, device=dev)}
step = model.make_graphed_train_step(batch)
for _ in range(5):
  step(batch)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
t0 = time.perf_counter()
for _ in range(n):
  step(batch)
torch.cuda.synchronize()
print("graphed ms/step", (time.perf_counter() - t0) / n * 1e3)
