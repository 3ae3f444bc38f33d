"""CPU-only checks of the boundary and the host logic (no GPU compute):
the C-ABI library loads and exports every symbol include/tfrs_hip.h declares, argument
validation returns the documented error codes, host classes raise the reference's
errors, the product path refuses to run without a GPU, and the multi-GPU exchange logic
works across 2 gloo ranks."""

import ctypes
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tfrs_hip.h")


@pytest.fixture(scope="module")
def lib():
  import __graft_entry__
  __graft_entry__.build()
  from recommenders_amd import _lib
  return _lib.load()


def _declared_symbols():
  text = open(HEADER).read()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return sorted(set(re.findall(r"\b(tfrs_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound(lib):
  from recommenders_amd import _lib
  declared = _declared_symbols()
  assert len(declared) >= 30
  so = ctypes.CDLL(_lib.LIB_PATH)
  missing = [s for s in declared if not hasattr(so, s)]
  assert not missing, f"declared in tfrs_hip.h but not exported: {missing}"
  unbound = [s for s in declared if s not in _lib.SIGNATURES]
  assert not unbound, f"declared but no ctypes prototype in _lib.SIGNATURES: {unbound}"
  extra = [s for s in _lib.SIGNATURES if s not in declared]
  assert not extra, f"bound but not declared in the header: {extra}"
  assert lib.tfrs_version() >= 100


def test_argument_validation_without_gpu(lib):
  """Validation happens before any HIP call, so it is testable on a CPU-only box."""
  from recommenders_amd import _lib
  rc = lib.tfrs_topk_merge(None, None, 0, 4, 10, 10, None, None, None, 0, None)
  assert rc == _lib.TFRS_EINVAL
  assert "topk_merge" in _lib.last_error()
  with pytest.raises(ValueError, match="topk_merge"):
    _lib.check(rc)
  rc = lib.tfrs_bruteforce_topk(None, None, 4, 10, None, None, None, 0, None)
  assert rc == _lib.TFRS_ESTATE
  with pytest.raises(ValueError, match="has not been built"):
    _lib.check(rc)
  rc = lib.tfrs_streaming_topk_update(None, 4, 4096, None, 8, 0, 5, None, None, 0, None, None, 0, None)
  assert rc == _lib.TFRS_ENOTIMPL
  with pytest.raises(NotImplementedError):
    _lib.check(rc)
  assert lib.tfrs_cross_fwd(None, None, None, None, ctypes.c_float(-1.0), 4, 4, None, None) == _lib.TFRS_EINVAL
  assert "non-negative" in _lib.last_error()
  assert lib.tfrs_embedding_segment_reduce_fwd(None, 10, 4, None, None, 1, None, 3, 7, None, None, None) == _lib.TFRS_EINVAL
  assert lib.tfrs_inbatch_softmax_ce_fwd(None, None, 8, 4, 16, None, ctypes.c_float(1.0), None, None, None,
                                         None, None, None, None, 0, None) == _lib.TFRS_EINVAL
  assert lib.tfrs_bruteforce_topk_workspace_bytes(8192, 1_000_000, 64, 100) > 8192 * 4096 * 4
  assert lib.tfrs_inbatch_softmax_workspace_bytes(4096, 4096, 64) > 0


def test_dot_interaction_strided_envelope(lib):
  """tfrs_dot_interaction_strided_supported is the single source of the fused-concat gate: the
  backward's 64 KB S tile ends at F = 122 for D = 32 although the forward covers F <= 128
  (a Ranking model with 123..128 features crashed in its first backward)."""
  q = lib.tfrs_dot_interaction_strided_supported
  assert q(131072, 101, 32, 0) == 1 and q(131072, 101, 32, 1) == 1        # BASELINE configs[4]
  assert q(2048, 27, 16, 0) == 1
  assert q(2048, 122, 32, 0) == 1
  for f in (123, 127, 128, 129):
    assert q(2048, f, 32, 0) == 0, f
  assert q(511, 27, 32, 0) == 0                                            # small batches: contiguous kernels
  assert q(2048, 27, 64, 0) == 0 and q(2048, 27, 24, 0) == 0


def test_adagrad_sparse_mode_belongs_to_the_latest_optimizer():
  """A second Adagrad built on the same tables before the first is collected must keep its sparse
  gradient mode when the first one dies (the old one's __del__ switched it off
  and training silently fell back to dense [vocab, d] gradients)."""
  import gc
  import recommenders_amd as tfrs
  from recommenders_amd.layers import embedding as emb
  layer = emb.Embedding(10, 4)
  first = tfrs.optimizers.Adagrad(layer.parameters(), 0.1)
  second = tfrs.optimizers.Adagrad(layer.parameters(), 0.1)
  layer.embeddings._tfrs_slices.append("pending")
  del first
  gc.collect()
  assert layer.embeddings._tfrs_sparse_grad is True
  assert layer.embeddings._tfrs_slices == ["pending"]          # pending slices survive too
  second.close()
  assert layer.embeddings._tfrs_sparse_grad is False


def test_streaming_cache_probe_never_iterates_lazy_datasets():
  """Only a list / tuple of resident tensors is eligible for Streaming's packed-block cache: a
  lazily mapped dataset (`candidates.map(item_model)`, re-embedded on every pass) must not even
  be iterated by the probe, let alone cached (ADVICE round 2, high + medium)."""
  from recommenders_amd.layers import factorized_top_k as ftk

  class Lazy:
    def __init__(self):
      self.passes = 0
    def __iter__(self):
      self.passes += 1
      yield np.zeros((4, 3), np.float32)

  ds = Lazy()
  layer = ftk.Streaming(k=2).index_from_dataset(ds)
  passes = ds.passes                      # (index_from_dataset may peek at the first element)
  assert layer._block_keys() is None
  assert ds.passes == passes
  assert ftk.Streaming(k=2).index_from_dataset([np.zeros((4, 3), np.float32)])._block_keys() is None   # host blocks
  import torch
  p = torch.nn.Parameter(torch.zeros((4, 3)))
  assert not ftk.Streaming._cacheable(p) and not ftk.Streaming._cacheable(p * 2.0)


def test_options_are_addressable_through_the_c_abi(lib, monkeypatch):
  """The TFRS_* switches are one configuration plane read through `tfrs::option`: a value set with
  tfrs_set_option wins over the environment, NULL removes the override (the
  switches were 19 scattered getenv calls)."""
  from recommenders_amd import _lib
  monkeypatch.delenv("TFRS_TOPK_STAT", raising=False)
  plan = (ctypes.c_int64 * 5)()
  assert _lib.get_option("TFRS_TOPK_STAT") is None
  assert lib.tfrs_debug_topk_plan(1_000_000, 100, 1, plan) == 0 and plan[4] == 1      # statistical plan
  _lib.set_option("TFRS_TOPK_STAT", "0")
  assert _lib.get_option("TFRS_TOPK_STAT") == "0"
  assert lib.tfrs_debug_topk_plan(1_000_000, 100, 1, plan) == 0 and plan[4] == 0 and plan[3] == 100
  monkeypatch.setenv("TFRS_TOPK_STAT", "1")                  # the override still wins
  assert lib.tfrs_debug_topk_plan(1_000_000, 100, 1, plan) == 0 and plan[4] == 0
  _lib.set_option("TFRS_TOPK_STAT", None)                    # back to the environment
  assert _lib.get_option("TFRS_TOPK_STAT") == "1"
  assert lib.tfrs_debug_topk_plan(1_000_000, 100, 1, plan) == 0 and plan[4] == 1
  with pytest.raises(ValueError, match="TFRS_"):
    _lib.set_option("PATH", "x")


def test_streaming_group_regimes_size_their_workspace(lib, monkeypatch):
  """tfrs_streaming_topk_blocks_workspace_bytes follows the regime a group will take (topk_api.hip:
  group_uses_raw16 / group_uses_f16): the block-fed fp16 filter needs one range's top-K + a norm word on top
  of the round workspace, the fp16-image regime the image of the whole group; the switches move the
  boundaries (host logic only -- nothing is launched)."""
  from recommenders_amd import _lib
  for name in ("TFRS_STREAM_RAW16_MIN_NQ", "TFRS_STREAM_RAW16_MAX_NQ", "TFRS_STREAM_RAW_MAX_NQ", "TFRS_TOPK_FILTER"):
    monkeypatch.delenv(name, raising=False)
  n, k = 12_500_000, 100
  size = lambda nq, d: int(lib.tfrs_streaming_topk_blocks_workspace_bytes(nq, n, d, k))
  image = n * (128 * 2 + 16)                                  # fp16 image of the group at dim 128
  assert size(8192, 128) > image                              # large batches: the image regime
  assert size(128, 128) < image and size(256, 128) < image    # block-fed filter up to 256 queries ...
  assert size(512, 128) < image and size(1024, 128) < image   # ... and, from dim 32 on, up to 1024 (rawscan16pc_kernel)
  assert size(1025, 128) > image                              # the image beyond ...
  assert size(512, 16) > n * (16 * 2 + 16) > size(256, 16)    # ... and below dim 32 beyond 256 queries
  try:
    _lib.set_option("TFRS_STREAM_RAW16_MAX_NQ", "0")          # off: 65+ queries go through the image
    assert size(128, 128) > image and size(64, 128) < image
    _lib.set_option("TFRS_STREAM_RAW16_MAX_NQ", None)
    # one query group below dim 128 stays on the exact scan unless asked: the filter's extra buffers
    # (one range's top-K: 2 * nq * k * 4 bytes, aligned, + the norm word) appear with the switch
    base = size(8, 64)
    _lib.set_option("TFRS_STREAM_RAW16_MIN_NQ", "1")
    assert size(8, 64) > base
    _lib.set_option("TFRS_STREAM_RAW16_MIN_NQ", None)
    assert size(8, 128) > size(8, 64)                         # dim 128 takes the filter from one query on
    _lib.set_option("TFRS_TOPK_FILTER", "f32")                # no fp16 anywhere: exact rounds for every batch
    assert size(8192, 128) < image
  finally:
    for name in ("TFRS_STREAM_RAW16_MIN_NQ", "TFRS_STREAM_RAW16_MAX_NQ", "TFRS_TOPK_FILTER"):
      _lib.set_option(name, None)


def test_host_classes_reference_errors():
  import recommenders_amd as tfrs
  ftk = tfrs.layers.factorized_top_k
  q = np.zeros((2, 4), np.float32)
  with pytest.raises(NotImplementedError, match="index_from_dataset"):
    ftk.Streaming().index(q)
  with pytest.raises(ValueError, match="must be called first"):
    ftk.Streaming()(q)
  with pytest.raises(ValueError, match="must be called first"):
    ftk.BruteForce()(q)
  with pytest.raises(ValueError, match="must be 2D"):
    ftk.BruteForce().index(np.zeros((3,), np.float32))
  with pytest.raises(ValueError, match="same number of"):
    ftk.BruteForce().index(np.zeros((3, 2), np.float32), np.arange(2))
  with pytest.raises(ValueError, match="tuples of"):
    ftk.BruteForce().index_from_dataset([(1, 2, 3)])
  with pytest.raises(ValueError, match="same batch dimension"):
    ftk.Streaming().index_from_dataset([(np.arange(3), np.zeros((2, 4), np.float32))])
  assert ftk.BruteForce().is_exact() and ftk.Streaming().is_exact()
  with pytest.raises(ValueError, match="should be non-negative"):
    tfrs.layers.feature_interaction.Cross(diag_scale=-1.0)
  with pytest.raises(ValueError, match="dimensions must be equal"):
    tfrs.layers.feature_interaction.DotInteraction()([torch.zeros(1, 3), torch.zeros(1, 2)])
  with pytest.raises(ValueError, match="candidate ids"):
    tfrs.tasks.Retrieval(remove_accidental_hits=True)(torch.zeros(2, 3), torch.zeros(2, 3))
  with pytest.raises(NotImplementedError, match="compute_loss"):
    tfrs.Model().compute_loss(None)
  m = tfrs.metrics.FactorizedTopK(candidates=[np.zeros((4, 2), np.float32)], ks=(1, 5))
  assert [x.name for x in m.metrics] == ["factorized_top_k/top_1_categorical_accuracy",
                                         "factorized_top_k/top_5_categorical_accuracy"]


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_product_path_fails_loudly_without_gpu():
  import recommenders_amd as tfrs
  with pytest.raises(RuntimeError, match="no CPU fallback"):
    tfrs.layers.factorized_top_k.BruteForce(k=2).index(np.zeros((8, 4), np.float32))
  with pytest.raises(RuntimeError, match="no CPU fallback"):
    tfrs.layers.embedding.gather_rows(torch.zeros(4, 4), torch.zeros(2, dtype=torch.long))


def test_product_package_never_imports_oracle():
  """The oracle is test infrastructure: nothing under recommenders_amd/ may reference it."""
  pkg = os.path.join(ROOT, "recommenders_amd")
  offenders = []
  for base, _, files in os.walk(pkg):
    for f in files:
      if f.endswith((".py", ".hip", ".cpp", ".h")):
        text = open(os.path.join(base, f), errors="ignore").read()
        if re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M) or "oracle/" in text and f.endswith(".py"):
          offenders.append(os.path.join(base, f))
  assert not offenders, offenders


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from oracle import topk as o_topk
from recommenders_amd.layers import factorized_top_k as ftk

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
rng = np.random.default_rng(123)
n, nq, d, k = 4000, 37, 16, 25
cand = rng.integers(-3, 4, size=(n, d)).astype(np.float32)      # integer valued: many ties
qry = rng.integers(-3, 4, size=(nq, d)).astype(np.float32)
per = n // world
lo, hi = rank * per, (rank + 1) * per if rank < world - 1 else n

def local_search(q, c, kk):                                      # oracle stands in for the GPU scan
  s, i = o_topk.brute_force(np.asarray(q), np.asarray(c), kk)
  return torch.from_numpy(s), torch.from_numpy(i.astype(np.int32))

def merge(all_s, all_i, kk):                                     # oracle stands in for tfrs_topk_merge
  w, b, _ = all_s.shape
  fs = all_s.permute(1, 0, 2).reshape(b, -1).numpy()
  fi = all_i.permute(1, 0, 2).reshape(b, -1).numpy()
  out_s, out_i = np.empty((b, kk), np.float32), np.empty((b, kk), np.int32)
  for r in range(b):
    order = np.lexsort((fi[r], -fs[r]))[:kk]
    out_s[r], out_i[r] = fs[r][order], fi[r][order]
  return torch.from_numpy(out_s), torch.from_numpy(out_i)

layer = ftk.ShardedBruteForce(k=k, local_search=local_search, merge=merge).index(cand[lo:hi], base_row=lo)
s, i = layer(qry)
es, ei = o_topk.brute_force(qry, cand, k)
assert np.array_equal(i.numpy(), ei), "sharded indices differ from the single-shard oracle"
assert np.array_equal(s.numpy(), es)
# global rows beyond int32 (SURVEY 8e): the shards sit at rows 5e9 + ..., rank order REVERSED with respect
# to row order (rank 0 owns the higher rows): int64 rows, same scores, ties still by ascending global row
big = 5_000_000_000
base = big + (n - hi)                     # rank 0 -> the LAST rows of the corpus
wide = ftk.ShardedBruteForce(k=k, local_search=local_search, merge=merge).index(cand[lo:hi], base_row=base)
assert wide._wide
s2, i2 = wide(qry)
assert i2.dtype == torch.int64
# the same corpus in global-row order: rank 1's rows first
order = np.concatenate([np.arange(per * (world - 1 - r), n if r == 0 else per * (world - r)) for r in range(world)])
es2, ei2 = o_topk.brute_force(qry, cand[order], k)
assert np.array_equal(s2.numpy(), es2)
assert np.array_equal(i2.numpy(), ei2.astype(np.int64) + big), "int64 global rows differ"
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharded_topk_two_ranks_gloo(tmp_path):
  """world_size-2 gloo run of ShardedBruteForce: shard -> local top-K -> all_gather ->
  merge gives exactly the single-shard answer on every rank (ties included)."""
  script = tmp_path / "worker.py"
  script.write_text(_WORKER.format(root=ROOT))
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  port = s.getsockname()[1]
  s.close()
  env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2",
             OMP_NUM_THREADS="2")
  procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
           for r in range(2)]
  outs = [p.communicate(timeout=240)[0].decode() for p in procs]
  for r, (p, o) in enumerate(zip(procs, outs)):
    assert p.returncode == 0, f"rank {r} failed:\n{o}"
    assert f"rank {r} ok" in o


_EMB_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from oracle import embedding as o_emb
from recommenders_amd.layers.sharded_embedding import ShardedEmbedding

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
V, D, B = 103, 8, 57                                  # V not divisible by the world size
rng = np.random.default_rng(7)
full = rng.normal(size=(V, D)).astype(np.float32)     # the same full table on every rank

def gather(shard, ids):                               # oracle stands in for the HIP gather
  return torch.from_numpy(o_emb.gather(shard.detach().numpy(), ids.numpy()))
def scatter(g, ids, vocab):                           # ... and for the scatter-add
  return torch.from_numpy(o_emb.scatter_add_grad(g.numpy(), ids.numpy(), vocab))

layer = ShardedEmbedding(V, D, device=torch.device("cpu"), local_gather=gather, local_scatter=scatter)
lo, hi = layer.row_range
with torch.no_grad():
  layer.embeddings.copy_(torch.from_numpy(full[lo:hi]))
ids = np.random.default_rng(100 + rank).integers(0, V, size=(B,))   # each rank: its own batch
w = np.random.default_rng(200 + rank).normal(size=(B, D)).astype(np.float32)
out = layer(torch.from_numpy(ids))
assert np.array_equal(out.detach().numpy(), o_emb.gather(full, ids)), "sharded lookup != full-table gather"
(out * torch.from_numpy(w)).sum().backward()
# expected shard gradient: every rank's (ids, w) contributions that fall into my row range
all_ids = [np.random.default_rng(100 + r).integers(0, V, size=(B,)) for r in range(world)]
all_w = [np.random.default_rng(200 + r).normal(size=(B, D)).astype(np.float32) for r in range(world)]
ref = o_emb.scatter_add_grad(np.concatenate(all_w), np.concatenate(all_ids), V)[lo:hi]
np.testing.assert_allclose(layer.embeddings.grad.numpy(), ref, rtol=1e-6, atol=1e-6)
# split sizes one batch ahead: stage(next ids) now, look them up later -- same result; looking up a batch that
# was written to after staging, or another tensor, RAISES (a per-rank fallback to the inline collective could
# leave the ranks in different collectives) and drops the staged batch; unstage() drops it explicitly
nxt = torch.from_numpy(np.random.default_rng(300 + rank).integers(0, V, size=(B,)))
layer.stage(nxt)
assert layer._staged is not None and layer._staged.ids is nxt
out2 = layer(nxt)
assert layer._staged is None
assert np.array_equal(out2.detach().numpy(), o_emb.gather(full, nxt.numpy())), "staged lookup differs"
layer.stage(nxt)
nxt[0] = (int(nxt[0]) + 1) % V                          # in-place write: the staged routing is stale
try:
  layer(nxt)
  raise AssertionError("stale staging was accepted")
except RuntimeError as e:
  assert "in-place write" in str(e) and layer._staged is None
out3 = layer(nxt)                                      # (every rank raised: all are back on the inline path)
assert np.array_equal(out3.detach().numpy(), o_emb.gather(full, nxt.numpy()))
layer.stage(nxt)
other = nxt.clone()
try:
  layer(other)                                         # not the staged object
  raise AssertionError("a lookup of another tensor took the staged routing")
except RuntimeError as e:
  assert "another tensor" in str(e)
layer.stage(nxt)
layer.unstage()
out4 = layer(other)
assert np.array_equal(out4.detach().numpy(), o_emb.gather(full, other.numpy()))
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharded_embedding_two_ranks_gloo(tmp_path):
  """world_size-2 gloo run of ShardedEmbedding: ids -> owners -> rows back, and gradient rows ->
  owners, equal the full-table gather / scatter-add on every rank."""
  script = tmp_path / "emb_worker.py"
  script.write_text(_EMB_WORKER.format(root=ROOT))
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  port = s.getsockname()[1]
  s.close()
  env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2",
             OMP_NUM_THREADS="2")
  procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
           for r in range(2)]
  outs = [p.communicate(timeout=240)[0].decode() for p in procs]
  for r, (p, o) in enumerate(zip(procs, outs)):
    assert p.returncode == 0, f"rank {r} failed:\n{o}"
    assert f"rank {r} ok" in o


_XREP_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from recommenders_amd.tasks.retrieval import cross_replica_concat

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
n, d = 5, 3
rng = np.random.default_rng(11)
blocks = [torch.tensor(rng.normal(size=(n, d)), dtype=torch.float32) for _ in range(world)]
coef = [torch.tensor(rng.normal(size=(world * n, d)), dtype=torch.float32) for _ in range(world)]

mine = blocks[rank].clone().requires_grad_(True)
cat = cross_replica_concat(mine)
# own block first, then the following ranks' blocks (reference docstring :239-292)
want = torch.cat([blocks[(rank + s) % world] for s in range(world)], dim=0)
assert torch.equal(cat.detach(), want), "wrong concatenation order"
loss = (cat * coef[rank]).sum() + (cat ** 2).sum()        # this rank's loss
loss.backward()

# single-process reference: total loss = sum of the ranks' losses, as a function of all blocks
ref = [b.clone().requires_grad_(True) for b in blocks]
total = 0.0
for r in range(world):
  cat_r = torch.cat([ref[(r + s) % world] for s in range(world)], dim=0)
  total = total + (cat_r * coef[r]).sum() + (cat_r ** 2).sum()
total.backward()
assert torch.allclose(mine.grad, ref[rank].grad, rtol=1e-6, atol=1e-6), "wrong gradient"

ids = cross_replica_concat(torch.arange(n) + 100 * rank)   # integer side inputs ride along
assert ids.tolist() == [100 * ((rank + s) % world) + i for s in range(world) for i in range(n)]
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_cross_replica_concat_two_ranks_gloo(tmp_path):
  """world_size-2 gloo run of the cross-replica negatives exchange (tasks/retrieval.py:238-321):
  own block first, and the gradient of a rank's block is the sum of all ranks' gradients for it."""
  script = tmp_path / "xrep_worker.py"
  script.write_text(_XREP_WORKER.format(root=ROOT))
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  port = s.getsockname()[1]
  s.close()
  env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2",
             OMP_NUM_THREADS="2")
  procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
           for r in range(2)]
  outs = [p.communicate(timeout=240)[0].decode() for p in procs]
  for r, (p, o) in enumerate(zip(procs, outs)):
    assert p.returncode == 0, f"rank {r} failed:\n{o}"
    assert f"rank {r} ok" in o


def test_cross_replica_concat_is_identity_without_process_group():
  from recommenders_amd.tasks.retrieval import cross_replica_concat
  x = torch.arange(6.0).reshape(3, 2)
  assert cross_replica_concat(x) is x



def _run_two_ranks(tmp_path, name, source):
  script = tmp_path / name
  script.write_text(source.format(root=ROOT))
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  port = s.getsockname()[1]
  s.close()
  env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2",
             OMP_NUM_THREADS="2")
  procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
           for r in range(2)]
  outs = [p.communicate(timeout=240)[0].decode() for p in procs]
  for r, (p, o) in enumerate(zip(procs, outs)):
    assert p.returncode == 0, f"rank {r} failed:\n{o}"
    assert f"rank {r} ok" in o


_DP_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import recommenders_amd as tfrs

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
torch.manual_seed(5)
B, D, H = 64, 12, 7                               # global batch; every rank trains on B / world rows
x = torch.randn(B, D); y = torch.randn(B)

class Net(tfrs.Model):
  def __init__(self):
    super().__init__()
    self.a = torch.nn.Linear(D, H); self.b = torch.nn.Linear(H, 1)
  def compute_loss(self, inputs, training=False):
    xb, yb = inputs
    per_example = (self.b(torch.tanh(self.a(xb))).reshape(-1) - yb) ** 2
    replicas = dist.get_world_size() if self._sync_world() > 1 else 1
    return per_example.mean() / replicas          # experimental/models/ranking.py:198-201

def make(sync):
  torch.manual_seed(9)
  m = Net()
  m.compile(optimizer=tfrs.optimizers.Adagrad(m.parameters(), learning_rate=0.3),
            sync_gradients=sync, bucket_bytes=128)   # tiny buckets: several reduce-scatter rounds
  return m

dp = make(None)                                   # default: synchronise (2 ranks are up)
single = make(False)                              # one "rank" on the full batch, no exchange
per = B // world
for step in range(4):
  dp.train_step((x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per]))
  single.train_step((x, y))
for (n1, p1), (n2, p2) in zip(dp.named_parameters(), single.named_parameters()):
  assert torch.allclose(p1, p2, rtol=1e-6, atol=1e-6), (n1, float((p1 - p2).abs().max()))
# replicas are bit-identical
for p in dp.parameters():
  both = [torch.empty_like(p) for _ in range(world)]
  dist.all_gather(both, p.detach())
  assert torch.equal(both[0], both[1])

# embedding slices: every rank ends with the rank-ordered concatenation of all ranks' (ids, rows)
class Slices(tfrs.Model):
  def __init__(self):
    super().__init__()
    self.table = torch.nn.Parameter(torch.zeros(10, 3))
m = Slices(); m.compile(optimizer=None)
m.table._tfrs_sparse_grad = True
n_mine = 3 + rank                                  # ragged counts across ranks
m.table._tfrs_slices = [(torch.arange(n_mine) + 10 * rank, torch.full((n_mine, 3), float(rank + 1)))]
m._all_reduce_gradients()
ids, rows = m.table._tfrs_slices[0]
assert ids.tolist() == [0, 1, 2, 10, 11, 12, 13], ids.tolist()
assert rows[:, 0].tolist() == [1.0] * 3 + [2.0] * 4
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_data_parallel_train_step_two_ranks_gloo(tmp_path):
  """world_size-2 gloo run of `Model.train_step` with the gradient exchange: two ranks on
  half-batches reach the parameters of one rank on the full batch (1e-6), replicas stay
  bit-identical, and embedding slices are all-gathered in rank order."""
  _run_two_ranks(tmp_path, "dp_worker.py", _DP_WORKER)


_SSTREAM_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from oracle import topk as o_topk
from recommenders_amd.layers import factorized_top_k as ftk

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
rng = np.random.default_rng(321)
n, nq, d, k = 1500, 19, 8, 40
cand = rng.integers(-3, 4, size=(n, d)).astype(np.float32)
qry = rng.integers(-3, 4, size=(nq, d)).astype(np.float32)
lo, hi = (0, 700) if rank == 0 else (700, n)          # uneven shards

class FakeStreaming(ftk.ShardedStreaming):            # oracle stands in for the GPU scan
  def _query_rows(self, queries, kk):
    blocks = list(self._candidates)
    s, i = o_topk.brute_force(np.asarray(queries), np.concatenate(blocks), kk)
    return torch.from_numpy(s), torch.from_numpy(i.astype(np.int32)) + self._base_row

def merge(all_s, all_i, kk):
  w, b, _ = all_s.shape
  fs = all_s.permute(1, 0, 2).reshape(b, -1).numpy()
  fi = all_i.permute(1, 0, 2).reshape(b, -1).numpy()
  out_s, out_i = np.empty((b, kk), np.float32), np.empty((b, kk), np.int32)
  for r in range(b):
    order = np.lexsort((fi[r], -fs[r]))[:kk]
    out_s[r], out_i[r] = fs[r][order], fi[r][order]
  return torch.from_numpy(out_s), torch.from_numpy(out_i)

layer = FakeStreaming(k=k, merge=merge).index_from_dataset(
    [cand[lo:hi][j:j + 128] for j in range(0, hi - lo, 128)], base_row=lo)
s, i = layer(qry)
es, ei = o_topk.brute_force(qry, cand, k)
assert np.array_equal(i.numpy(), ei) and np.array_equal(s.numpy(), es)

# ShardedBruteForce: global rows beyond int32 switch every rank to the int64 exchange (round 4; refused
# before); a negative base row is still an error
wide = ftk.ShardedBruteForce(k=5, local_search=lambda *a: None).index(cand[:10], base_row=2**31 - 5)
assert wide._wide
try:
  ftk.ShardedBruteForce(k=5, local_search=lambda *a: None).index(cand[:10], base_row=-1)
  raise SystemExit("expected ValueError")
except ValueError as e:
  assert "base_row" in str(e)
# integer identifiers are resolved by their owners
def local_search(q, c, kk):
  s_, i_ = o_topk.brute_force(np.asarray(q), np.asarray(c), kk)
  return torch.from_numpy(s_), torch.from_numpy(i_.astype(np.int32))
ids_all = (np.arange(n) * 7 + 3).astype(np.int64)
sb = ftk.ShardedBruteForce(k=k, local_search=local_search, merge=merge).index(
    cand[lo:hi], identifiers=ids_all[lo:hi], base_row=lo)
s2, id2 = sb(qry)
assert np.array_equal(id2.numpy(), ids_all[ei])
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharded_streaming_and_identifiers_two_ranks_gloo(tmp_path):
  """world_size-2 gloo run: ShardedStreaming (uneven shards, global row numbers from base_row)
  and ShardedBruteForce with integer identifiers equal the single-shard oracle; a base_row
  that would overflow int32 row numbers raises."""
  _run_two_ranks(tmp_path, "sstream_worker.py", _SSTREAM_WORKER)


def test_sharded_embedding_routes_out_of_range_ids_consistently():
  """ids outside [0, input_dim) must not desynchronise the exchange: they become row -1 on
  rank 0 (a zero row, no gradient)."""
  from oracle import embedding as o_emb
  from recommenders_amd.layers.sharded_embedding import ShardedEmbedding
  def gather(shard, ids):
    out = np.zeros((ids.numel(), shard.shape[1]), np.float32)
    ok = (ids.numpy() >= 0) & (ids.numpy() < shard.shape[0])
    out[ok] = shard.detach().numpy()[ids.numpy()[ok]]
    return torch.from_numpy(out)
  def scatter(g, ids, vocab):
    ok = (ids.numpy() >= 0) & (ids.numpy() < vocab)
    return torch.from_numpy(o_emb.scatter_add_grad(g.numpy()[ok], ids.numpy()[ok], vocab))
  layer = ShardedEmbedding(10, 4, device=torch.device("cpu"), local_gather=gather, local_scatter=scatter)
  ids = torch.tensor([3, -2, 10, 9, 123456])
  out = layer(ids)
  want = layer.embeddings.detach()[torch.tensor([3, 0, 0, 9, 0])].clone()
  want[[1, 2, 4]] = 0.0
  assert torch.equal(out.detach(), want)
  out.sum().backward()
  g = layer.embeddings.grad
  assert g[3].eq(1).all() and g[9].eq(1).all() and g.sum() == 8.0


def test_threshold_pass_plan_statistics(lib, monkeypatch):
  """csrc/topk_api.hip plan_sample / stat_rank (host code): the statistical rank m of a shuffled
  index is the smallest with P(Binomial(K - 1, 1 / stride) >= m) <= 1e-7, the survivor list it
  implies fits the list kernel (P(1.35 T > 1024) <= 1e-8 with T the negative-binomial position
  of the m-th sampled row), and the guaranteed plan keeps rank = K."""
  from scipy.stats import binom
  for key in ("TFRS_TOPK_STAT", "TFRS_TOPK_SAMPLE", "TFRS_TOPK_SAMPLE_STAT", "TFRS_TOPK_STAT_PFAIL"):
    monkeypatch.delenv(key, raising=False)
  plan = (ctypes.c_int64 * 5)()
  for n in (1_000_000, 12_500_000, 100_000_000):
    for k in (1, 10, 100, 256, 512):
      assert lib.tfrs_debug_topk_plan(n, k, 0, plan) == 0
      stride, stages, _, rank, stat = list(plan)
      assert rank == k and stat == 0 and stages > 0 and 1 <= stride <= 4
      assert lib.tfrs_debug_topk_plan(n, k, 1, plan) == 0
      stride, stages, _, rank, stat = list(plan)
      assert 1 <= stride <= 16 and stages == (n // 128) // stride and 1 <= rank <= k
      assert stat == (1 if rank < k else 0)
      if stride > 1 and k > 1:
        f = 1.0 / stride
        assert binom.sf(rank - 1, k - 1, f) <= 1e-7                 # P(bound too high)
        assert rank == 1 or binom.sf(rank - 2, k - 1, f) > 1e-7     # ... and no larger than needed
        assert binom.cdf(rank - 1, 758, f) <= 1e-8                  # the list overflows its 1024 slots
      assert 2 * stages >= 8 * rank                                 # enough bins for the rank
  # a corpus too small for the prefilter, and the statistical plan switched off
  assert lib.tfrs_debug_topk_plan(20_000, 100, 1, plan) == 0 and plan[1] == 0
  monkeypatch.setenv("TFRS_TOPK_STAT", "0")
  assert lib.tfrs_debug_topk_plan(1_000_000, 100, 1, plan) == 0
  assert plan[3] == 100 and plan[4] == 0


def test_embedding_dict_host_logic():
  """experimental.models.ranking.EmbeddingDict: the tables are row ranges of one parameter
  (host-side bookkeeping only; the lookups themselves are GPU tests)."""
  import torch
  from recommenders_amd.experimental.models import ranking as rk
  emb = rk.EmbeddingDict({"b": 7, "a": 5, 3: 2}, 4, device=torch.device("cpu"))
  assert [n for n, _ in emb.named_parameters()] == ["embeddings"]
  assert tuple(emb.embeddings.shape) == (14, 4)
  assert float(emb.embeddings.detach().abs().max()) <= 0.05
  views = emb.tables
  assert list(views) == ["b", "a", "3"]
  assert views["a"].embeddings.data_ptr() == emb.embeddings[7:12].data_ptr()
  assert tuple(views["3"].embeddings.shape) == (2, 4)
  assert emb._start_rows.tolist() == [0, 7, 12]
  ids = lambda n: torch.zeros((n,), dtype=torch.int64)
  assert emb.can_stack({"a": ids(6), "b": ids(6), 3: ids(6)})
  assert not emb.can_stack({"a": ids(6), "b": ids(5)})               # different batch sizes
  assert not emb.can_stack({"a": ids(6)[:, None], "b": ids(6)[:, None]})   # [B, 1] ids: generic path
  assert not emb.can_stack({"a": ids(6), "zz": ids(6)})               # unknown feature
  with pytest.raises(KeyError, match="no table"):
    emb({"zz": ids(3)})
  with pytest.raises(ValueError, match="at least one row"):
    rk.EmbeddingDict({"a": 0}, 4, device=torch.device("cpu"))


_ASM_CACHE = {}


def _device_asm(source, defines=()):
  """gfx950 assembly of one kernel source (cross-compiled once per test session, no GPU needed)."""
  key = (source, tuple(defines))
  if key not in _ASM_CACHE:
    import subprocess, tempfile
    from recommenders_amd.csrc import build as csrc_build
    src = os.path.join(os.path.dirname(csrc_build.__file__), source)
    out = os.path.join(tempfile.mkdtemp(prefix="tfrs_asm_"), "kernel.s")
    cmd = [csrc_build.hipcc(), f"--offload-arch={csrc_build.ARCH}", "-O3", "-std=c++17",
           *csrc_build.EXTRA_FLAGS.get(source, []), *[f"-D{d}" for d in defines],
           "-S", "--cuda-device-only", "-o", out, src]
    subprocess.run(cmd, check=True, capture_output=True, cwd=os.path.dirname(src))
    with open(out) as f:
      _ASM_CACHE[key] = f.read()
  return _ASM_CACHE[key]


def _mfma_hazard_checker():
  import importlib.util
  path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_mfma_hazards.py")
  spec = importlib.util.spec_from_file_location("check_mfma_hazards", path)
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


@pytest.mark.parametrize("source", ["topk_scan16.hip", "topk_raw.hip", "topk_scan.hip", "gemm16.hip", "interaction.hip",
                                    "softmax.hip", "softmax16.hip", "metric_fused.hip"])
def test_no_mfma_result_is_read_before_the_matrix_pipe_delivers_it(source):
  """gfx950 has no interlock between an MFMA in flight and a VALU / LDS / memory instruction that touches its
  destination: the wait states (12 behind v_mfma_f32_32x32x16_f16) are software's job, normally the compiler's.
  Round 5 caught its hazard recognizer leaving them out in the fp16 filter kernel (below); every kernel of the
  library that issues MFMAs is therefore walked on the cross-compiled assembly."""
  res = _mfma_hazard_checker().check(_device_asm(source))
  assert res, source
  assert not {k: v[:3] for k, v in res.items() if v}


def test_mfma_hazard_checker_counts_wait_states_on_a_hand_written_listing():
  """The checker's own arithmetic on listings small enough to read: 12 wait states behind the 8-pass fp16 shape
  (an instruction is one, `s_nop N` is N + 1, comments and labels none), 18 behind the 16-pass f32 shape, the
  chain's next MFMA is not a reader, an unconditional branch ends the fall-through path, AGPR destinations are
  tracked separately from VGPRs of the same number."""
  chk = _mfma_hazard_checker()

  def listing(body):
    return "_Z1kv:\n" + "\n".join("\t" + l for l in body) + "\n\t.amdhsa_kernel _Z1kv\n"

  mf = "v_mfma_f32_32x32x16_f16 v[0:15], v[20:23], v[24:27], v[0:15]"
  ok = [mf, "s_nop 7", "; a comment", ".LBB0_1:", "s_nop 3", "v_max_f32_e32 v40, v0, v1"]          # 8 + 4 = 12
  assert chk.check(listing(ok)) == {"_Z1kv": []}
  short = [mf, "s_nop 7", "s_nop 2", "v_max_f32_e32 v40, v0, v1"]                                    # 8 + 3 = 11
  (bad,) = chk.check(listing(short))["_Z1kv"]
  assert bad[1].startswith("v_max_f32") and bad[3] == 1
  chain = [mf, mf, "s_nop 7", "s_nop 3", "ds_write_b128 v50, v[12:15]"]                              # counted from the LAST link
  assert chk.check(listing(chain)) == {"_Z1kv": []}
  assert chk.check(listing([mf, mf, "s_nop 7", "s_nop 2", "ds_write_b128 v50, v[12:15]"]))["_Z1kv"]
  other = [mf, "v_add_u32_e32 v40, v41, v42", "v_accvgpr_read_b32 v43, a0"]                          # a0 is not v0
  assert chk.check(listing(other)) == {"_Z1kv": []}
  branch = [mf, "s_branch .LBB0_9", "v_max_f32_e32 v40, v0, v1"]                                     # not the fall-through
  assert chk.check(listing(branch)) == {"_Z1kv": []}
  # ... but the TAKEN path is followed too: a side block that reads the accumulator, a loop back edge to a reader
  side = [mf, "s_cbranch_vccnz .LBB0_7", "s_nop 7", "s_nop 7", "s_endpgm", ".LBB0_7:", "ds_write_b128 v50, v[0:3]"]
  (bad,) = chk.check(listing(side))["_Z1kv"]
  assert bad[1].startswith("ds_write_b128") and bad[3] == 11
  loop = [".LBB0_2:", "v_max_f32_e32 v40, v0, v1", "s_nop 7", "s_nop 7", mf, "s_cbranch_scc1 .LBB0_2"]
  assert [b[1][:9] for b in chk.check(listing(loop))["_Z1kv"]] == ["v_max_f32"]
  f32 = "v_mfma_f32_32x32x2_f32 a[0:15], v20, v21, a[0:15]"
  assert chk.check(listing([f32, "s_nop 15", "s_nop 1", "v_accvgpr_read_b32 v0, a3"])) == {"_Z1kv": []}   # 16 + 2 = 18
  assert chk.check(listing([f32, "s_nop 15", "s_nop 0", "v_accvgpr_read_b32 v0, a3"]))["_Z1kv"]


def test_the_mfma_hazard_check_finds_the_peeled_filter_kernel_without_hand_wait_states():
  """The build that lost survivors on the GPU (TFRS_SCAN16_PEEL=1: the queue append as a short side branch, wait
  states left to the compiler; profiles/r05_scan16f_peel.txt) is the checker's known positive: the max tree of
  every scan16f instantiation reads an accumulator 1-11 wait states early there, and nowhere else in the file."""
  res = _mfma_hazard_checker().check(_device_asm("topk_scan16.hip", ("TFRS_SCAN16_PEEL=1",)))
  bad = {k for k, v in res.items() if v}
  assert bad and all("scan16f_kernel" in k for k in bad)
  assert {k for k in res if "scan16f_kernelILi64ELi16ELi2ELi2E" in k} <= bad


@pytest.mark.parametrize("source,patterns,max_vgprs,agpr_spills_ok", [
    # the fp16 filter kernel of the headline path: four waves per SIMD
    ("topk_scan16.hip", ("scan16f_kernelILi64ELi8ELi2E", "scan16f_kernelILi32ELi8ELi2E"), 128, False),
    # ... and its 16-wave form (two query tiles on one stage buffer, two stages per barrier period: the
    # default for batches of at least two tiles up to dim 64) -- one workgroup per CU, still four waves per SIMD
    ("topk_scan16.hip", ("scan16f_kernelILi64ELi16ELi2ELi2E", "scan16f_kernelILi32ELi16ELi2ELi2E",
                         "scan16f_kernelILi16ELi16ELi2ELi2E"), 128, False),
    # the 256 x 256 split-fp16 GEMM (one 8-wave workgroup per CU: two waves per SIMD), all epilogues
    # (and the two instantiations that apply an activation in the epilogue)
    ("gemm16.hip", ("gemm16_big_kernelILi0ELb0E", "gemm16_big_kernelILi1ELb0E", "gemm16_big_kernelILi2ELb0E",
                    "gemm16_big_kernelILi3ELb0E", "gemm16_big_kernelILi0ELb1E", "gemm16_big_kernelILi1ELb1E"),
     256, False),
    # the DotInteraction producer / consumer kernels: backward one 8-wave workgroup per CU, forward two
    ("interaction.hip", ("dot_interaction_bwd_h16_kernelILi7ELi5ELi4E", "dot_interaction_bwd_h16_kernelILi7ELi6ELi4E"), 256, False),
    ("interaction.hip", ("dot_interaction_fwd_pc_kernelILi4ELi4E",), 128, False),
    # the block-fed fp16 filter of Streaming groups (one 4-wave workgroup per CU: one wave per SIMD): every
    # instantiation the launcher can pick -- four resident query groups at dim 128, eight up to dim 64 (there
    # 16 registers live in the accumulator file: no scratch memory, which is what the stage prefetch cares about)
    ("topk_raw.hip", ("rawscan16_kernelILi128ELi1E", "rawscan16_kernelILi128ELi2E", "rawscan16_kernelILi128ELi4E",
                      "rawscan16_kernelILi64ELi8E", "rawscan16_kernelILi32ELi8E", "rawscan16_kernelILi8ELi8E"), 512, True),
    # the wide block-fed filter, lock-step form (round 5; kept for A/B runs: one 8-wave workgroup per CU, two waves per
    # SIMD): the stage of rows in flight, the two query groups' B operands and one A-fragment set fill the 256 registers
    ("topk_raw.hip", ("rawscan16w_kernelILi128E", "rawscan16w_kernelILi64E", "rawscan16w_kernelILi32E"), 256, False),
    # ... and its producer / consumer form (round 6, default for 257-1024 queries): 12 waves = three per SIMD, 168 registers
    ("topk_raw.hip", ("rawscan16pc_kernelILi128ELi8E", "rawscan16pc_kernelILi64ELi8E", "rawscan16pc_kernelILi32ELi8E"), 168, False),
    # the fp16 image packer that keeps a stage's parity planes in registers
    ("topk_pack.hip", ("pack16_stage_regs_kernelILi128E", "pack16_stage_regs_kernelILi16E"), 128, False),
])
def test_hot_kernel_register_budgets(source, patterns, max_vgprs, agpr_spills_ok):
  """Hot kernels must keep their register budget and use no scratch -- a spill in the filter kernel
  puts `s_waitcnt vmcnt(0)` behind every stage prefetch.  Checked on the
  cross-compiled ISA metadata, no GPU needed."""
  found = set()
  for block in _device_asm(source).split("- .agpr_count:")[1:]:      # one metadata entry per kernel
    name = re.search(r"\.name:\s+(\S+)", block).group(1)
    hit = [p for p in patterns if p in name]
    if not hit:
      continue
    found.add(hit[0])
    assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", block).group(1)) == 0, name
    if not agpr_spills_ok:
      assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", block).group(1)) == 0, name
    assert int(re.search(r"\.vgpr_count:\s+(\d+)", block).group(1)) <= max_vgprs, name
  assert found == set(patterns)


@pytest.mark.parametrize("source,pattern,min_in_flight", [
    # producers: convert n + 1, load n + 5, reduce n + 2 -- three samples of 9 loads stay in flight
    ("interaction.hip", "dot_interaction_bwd_h16_kernelILi7ELi5ELi4E", 27),
    ("interaction.hip", "dot_interaction_fwd_pc_kernelILi4ELi4E", 12),
    # the Cross epilogue issues the 24 loads of eight rows of an accumulator tile before it waits
    ("gemm16.hip", "gemm16_big_kernelILi1ELb0E", 16),
    # single-pass row images: a wave's share of four rows (14 + 14 16-byte loads) in flight
    ("gemm16.hip", "g16_prep_rows_ksm1_kernelILi8ELb0ELb0E", 8),
    # ... and its form for 8-byte aligned rows (k % 4 == 2, eight waves per four rows): first built with a nested ternary per
    # element that compiled to a branch + vmcnt(0) behind every load (2.85 TB/s); the selects keep 4-5 loads in flight
    ("gemm16.hip", "g16_prep_rows_ksm1_kernelILi8ELb0ELb1ELi8E", 4),
    # round 6, last session -- latency chains of short kernels that the listing showed with ONE load in flight:
    # the in-batch softmax's record packer (one 4-byte load per round trip, eight in a row at dim 64)
    ("softmax16.hip", "sm16_prep_kernelILi64E", 4),
    # the row-scan Adagrad of small tables (ids, gradient rows of a row's hits, accumulator / table rows)
    ("embedding.hip", "scatter_rowscan_multi_kernel", 16),
    # the query prologue of the fp16 filter / threshold kernels: a wave's 16 x 16-byte query loads at once
    ("topk_scan16.hip", "scan16f_kernelILi64ELi16ELi2ELi2E", 8),
    ("topk_scan16.hip", "scan16_kernelILi64ELi2E", 8),
])
def test_pipelined_kernels_keep_loads_in_flight(source, pattern, min_in_flight):
  """gfx950 tracks a wave's outstanding loads with ONE in-order counter; a load inside a branch makes the
  count unknown to the compiler, which then waits with `s_waitcnt vmcnt(0)` everywhere and the software
  pipeline in the source does not exist in the ISA (found on four kernels in round
  3).  The kernels that rely on loads in flight must contain a wait that leaves at least
  `min_in_flight` loads outstanding."""
  asm = _device_asm(source)
  m = re.search(r"^(_Z\w*" + pattern + r"\w*):[^\n]*\n(.*?)^\s*\.amdhsa_kernel", asm, re.S | re.M)
  assert m, pattern
  waits = [int(v) for v in re.findall(r"s_waitcnt[^\n]*vmcnt\((\d+)\)", m.group(2))]
  assert waits and max(waits) >= min_in_flight, (pattern, sorted(set(waits))[-5:])


def test_every_library_switch_is_documented():
  """The `TFRS_*` switches read by the library (tfrs::option / env_int; settable through
  tfrs_set_option) form its configuration plane: each one appears in INTEGRATION.md's table."""
  import glob
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  names = set()
  for f in glob.glob(os.path.join(root, "recommenders_amd", "csrc", "*")):
    if f.endswith((".hip", ".cpp", ".h")):
      names.update(re.findall(r'"(TFRS_[A-Z0-9_]+)"', open(f).read()))
  doc = open(os.path.join(root, "INTEGRATION.md")).read()
  assert names and not [n for n in sorted(names) if n not in doc]
