"""A few cases of each tools/fuzz_*.py fuzzer per test run (fixed seeds; the tools run more):
random shapes and awkward data through the default paths against the all-f32 scan / float64."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")


def _load(name):
  spec = importlib.util.spec_from_file_location(name, os.path.join(TOOLS, name + ".py"))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


@pytest.mark.parametrize("name,seed,cases", [("fuzz_topk", 11, 5), ("fuzz_streaming", 12, 3),
                                             ("fuzz_gemm", 13, 5), ("fuzz_embedding", 14, 5)])
def test_fuzzers_find_nothing(name, seed, cases, monkeypatch):
  for key in ("TFRS_TOPK_FILTER", "TFRS_TOPK_STAT", "TFRS_GEMM_MODE"):
    monkeypatch.delenv(key, raising=False)
  mod = _load(name)
  assert (mod.main(seed, cases, light=True) if name == "fuzz_embedding" else mod.main(seed, cases)) == 0


# Every instantiation of the fp16 filter kernel (tfrs::scan16f_kernel<dim, waves, groups, stages per period>) on whole
# batches: the exactness of BruteForce rests on this kernel never losing a survivor (its per-wave LDS queue between
# check() and drain(), csrc/topk_scan16.hip), so each shape is held to the all-f32 search on 40 random cases per dim --
# 200 cases x up to 5 shapes -- with the data that stresses the queue: bursts of hot tiles (clusters, duplicates),
# row scales, few and many queries, k = 1 (entries that sit in the queue for many stages) .. 300.
@pytest.mark.parametrize("d", [16, 32, 64, 100, 128])
def test_scan16f_instantiations_keep_every_survivor(d, monkeypatch):
  import numpy as np
  import torch
  from recommenders_amd import _lib
  from recommenders_amd.layers import factorized_top_k as ftk
  for key in ("TFRS_TOPK_FILTER", "TFRS_TOPK_STAT", "TFRS_SCAN16_SHAPE", "TFRS_SCAN16_V"):
    monkeypatch.delenv(key, raising=False)
  rng = np.random.default_rng(500 + d)
  dev = torch.device("cuda", 0)
  shapes = ["8x2", "16x2", "4x4", "8x4", None] if d <= 64 else ["8x2", None]   # None = the default choice
  bad = []
  for case in range(40):
    n = int(rng.integers(66_000, 260_000))
    k = int(rng.choice([1, 10, 100, 300]))
    nq = int(rng.choice([1, 64, 500, 513, 1024, 1500, 2048, 3000]))
    kind = str(rng.choice(["gauss", "row_scales", "clustered", "dups", "hot_queries"]))
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
    c = torch.randn((n, d), generator=g, device=dev) / d ** 0.5
    q = torch.randn((nq, d), generator=g, device=dev) / d ** 0.5
    if kind == "row_scales":
      c *= torch.exp(1.5 * torch.randn((n, 1), generator=g, device=dev))
    elif kind == "clustered":
      cen = torch.randn((32, d), generator=g, device=dev) / d ** 0.5
      c = cen[torch.arange(n, device=dev) * 32 // n] + 0.2 * c
      q = cen[torch.randint(0, 32, (nq,), generator=g, device=dev)] + 0.2 * q
    elif kind == "dups":
      c[n // 3:2 * (n // 3)] = c[:n // 3].clone()
    elif kind == "hot_queries":          # identical queries: every lane of a tile is hot at once (queue bursts)
      q[:] = q[0]
    layer = ftk.BruteForce(k=k, dedup=(kind != "dups") and "auto").index(c)
    _lib.set_option("TFRS_TOPK_FILTER", "f32")
    try:
      s32, i32 = layer(q)
    finally:
      _lib.set_option("TFRS_TOPK_FILTER", None)
    for shape in shapes:
      _lib.set_option("TFRS_SCAN16_SHAPE", shape)
      try:
        s, i = layer(q)
      finally:
        _lib.set_option("TFRS_SCAN16_SHAPE", None)
      if not (torch.equal(s, s32) and torch.equal(i, i32)):
        bad.append((case, n, k, nq, kind, shape))
  assert not bad, bad
