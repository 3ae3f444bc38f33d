"""Where does Model.fit spend its time at the MovieLens-100K shapes?  (evidence tool)"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recommenders_amd as tfrs

dev = torch.device("cuda", 0)
B, D, V, ITEMS = 4096, 64, 2000, 1682
g = torch.Generator(device=dev).manual_seed(0)


class TwoTower(tfrs.Model):
  def __init__(self):
    super().__init__()
    self.user_model = tfrs.layers.embedding.Embedding(V, D)
    self.item_model = tfrs.layers.embedding.Embedding(V, D)
    movies = tfrs.data.Dataset.from_tensor_slices(torch.arange(ITEMS, device=dev))
    self.task = tfrs.tasks.Retrieval(metrics=tfrs.metrics.FactorizedTopK(candidates=movies.batch(128).map(self.item_model)))

  def compute_loss(self, inputs, training=False):
    return self.task(self.user_model(inputs["user_id"]), self.item_model(inputs["movie_id"]), compute_metrics=True)


sizes = [B] * 19 + [80_000 - 19 * B]
epoch = [{"user_id": torch.randint(0, 943, (n,), generator=g, device=dev),
          "movie_id": torch.randint(0, ITEMS, (n,), generator=g, device=dev)} for n in sizes]
m = TwoTower()
m.compile(optimizer=tfrs.optimizers.Adagrad(m.parameters(), learning_rate=0.5))
m.fit(epoch, epochs=3)
torch.cuda.synchronize()


def timed(fn, n):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(n):
    fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / n * 1e3


cache = m.__dict__["_fit_graphs"]
steps = {k: v for k, v in cache.items() if callable(v)}
print("captured:", [k[0][1] for k in steps])
big = [v for k, v in steps.items() if k[0][1] == (B,)][0]
print("replay, one batch        : %.3f ms" % timed(lambda: big(epoch[0]), 200))
print("replay, cycling 19 batches: %.3f ms/step" % (timed(lambda: [big(b) for b in epoch[:19]], 20) / 19))
print("graph.replay only         : %.3f ms" % timed(lambda: big.graph.replay(), 200))
print("fit epoch                 : %.3f ms/step" % (timed(lambda: m.fit(epoch, epochs=1), 10) / 20))
print("reset_states only         : %.3f ms" % timed(lambda: [x.reset_states() for x in m.metrics], 50))
pr = cProfile.Profile()
pr.enable()
m.fit(epoch, epochs=5)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

# the order bench.py uses: a first model is stepped eagerly, captured by hand and replayed, THEN a
# second model is fitted
print("---- second model after a hand-captured first one")
m1 = TwoTower()
m1.compile(optimizer=tfrs.optimizers.Adagrad(m1.parameters(), learning_rate=0.5))
print("eager step                : %.3f ms" % timed(lambda: m1.train_step(epoch[0]), 100))
graphed = m1.make_graphed_train_step(epoch[0])
print("hand-captured replay      : %.3f ms" % timed(lambda: graphed(epoch[0]), 300))
m2 = TwoTower()
m2.compile(optimizer=tfrs.optimizers.Adagrad(m2.parameters(), learning_rate=0.5))
m2.fit(epoch, epochs=3)
print("fit epoch (second model)  : %.3f ms/step" % (timed(lambda: m2.fit(epoch, epochs=1), 10) / 20))
big2 = [v for k, v in m2.__dict__["_fit_graphs"].items() if callable(v) and k[0][1] == (B,)][0]
print("its replay                : %.3f ms" % timed(lambda: big2(epoch[0]), 200))
print("errors:", m2.__dict__["_fit_graphs"].get("_errors"))

print("---- after a large device-to-host copy")
big_dev = torch.randn((1_000_000, 64), device=dev)
host = big_dev.cpu().numpy()
print("fit epoch (second model)  : %.3f ms/step" % (timed(lambda: m2.fit(epoch, epochs=1), 10) / 20))
print("its replay                : %.3f ms" % timed(lambda: big2(epoch[0]), 200))
print("replay cycling            : %.3f ms/step" % (timed(lambda: [big2(b) for b in epoch[:19]], 20) / 19))
import gc
del host
gc.collect()
print("fit epoch after freeing the host copy: %.3f ms/step" % (timed(lambda: m2.fit(epoch, epochs=1), 10) / 20))
m3 = TwoTower()
m3.compile(optimizer=tfrs.optimizers.Adagrad(m3.parameters(), learning_rate=0.5))
m3.fit(epoch, epochs=3)
print("fit epoch (third model, captured after the copy): %.3f ms/step" % (timed(lambda: m3.fit(epoch, epochs=1), 10) / 20))
pr = cProfile.Profile()
pr.enable()
m3.fit(epoch, epochs=5)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
