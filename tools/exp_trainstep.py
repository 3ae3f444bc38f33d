"""Kernel-trace target: the C1 two-tower train step (see bench.py train_step_metric)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd.layers import embedding as emb
from recommenders_amd.tasks.retrieval import in_batch_softmax_loss
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
B, D, V = 4096, 64, 2000
user, item = emb.Embedding(V, D), emb.Embedding(V, D)
uid = torch.randint(0, 943, (B,), generator=g, device=dev)
iid = torch.randint(0, 1682, (B,), generator=g, device=dev)
opt = torch.optim.Adagrad(list(user.parameters()) + list(item.parameters()), lr=0.5,
                          initial_accumulator_value=0.1, eps=1e-7)
def step():
  opt.zero_grad(set_to_none=True)
  loss = in_batch_softmax_loss(user(uid), item(iid))
  loss.backward()
  opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 50 * 1e3)
