import sys, time, torch
sys.path.insert(0, "/root/repo")
from recommenders_amd.layers import embedding as emb
g = torch.Generator(device="cuda").manual_seed(0)
vocab, d, n = 1_000_000, 32, 1_500_000
table = torch.randn((vocab, d), generator=g, device="cuda"); acc = torch.full_like(table, 0.1)
grad = torch.randn((n, d), generator=g, device="cuda")
for hot in (vocab, 10000, 1000, 50, 1):
  ids = torch.randint(0, hot, (n,), generator=g, device="cuda")
  for name, fn in (("adagrad", lambda: emb.adagrad_sparse_update_(table, acc, grad, ids, 0.05, 1e-7)),
                   ("dense_scatter", lambda: emb.scatter_add_rows(grad, ids, vocab))):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    print(hot, name, round((time.perf_counter() - t0) * 1e3, 3), "ms", flush=True)
