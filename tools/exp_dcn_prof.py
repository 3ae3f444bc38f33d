import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recommenders_amd as tfrs
from recommenders_amd.experimental.models import ranking as rk
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
n_tables, vocab, dim, batch = 26, 1000000, 128, 65536
emb = rk.EmbeddingDict({str(i): vocab for i in range(n_tables)}, dim)
bottom = tfrs.layers.blocks.MLP(units=[512, 256, dim], final_activation="relu")
fi = rk.ConcatCross(num_layers=3)
top = tfrs.layers.blocks.MLP(units=[1024, 512, 1], final_activation="sigmoid")
model = rk.Ranking(emb, bottom_stack=bottom, feature_interaction=fi, top_stack=top,
                   task=tfrs.tasks.Ranking(loss=tfrs.losses.BinaryCrossentropy(reduction="none")))
feats = {"dense_features": torch.rand((batch, 13), generator=g, device=dev),
         "sparse_features": {str(i): torch.randint(0, vocab, (batch,), generator=g, device=dev) for i in range(n_tables)}}
labels = torch.randint(0, 2, (batch,), generator=g, device=dev)
model(feats)
model.compile(optimizer=tfrs.optimizers.Adagrad(model.parameters(), learning_rate=0.01))
for _ in range(2): model.train_step((feats, labels))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): model.train_step((feats, labels))
torch.cuda.synchronize(); print("step ms", (time.perf_counter() - t0) / 5 * 1e3)
