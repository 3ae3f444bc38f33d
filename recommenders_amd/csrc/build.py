"""Builds recommenders_amd/libtfrs_hip.so with hipcc for gfx950 (in-tree).

    python -m recommenders_amd.csrc.build [--force]

One object per source (compiled in parallel), linked into a single shared library
that exports the C ABI declared in include/tfrs_hip.h.  No torch headers are used:
the boundary is plain pointers + sizes.
"""

import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(PKG, "libtfrs_hip.so")
ARCH = "gfx950"

SOURCES = [
    "api.cpp",
    "calibrate.hip",
    "topk_pack.hip",
    "topk_scan.hip",
    "topk_scan16.hip",
    "topk_select.hip",
    "topk_select16.hip",
    "topk_raw.hip",
    "topk_api.hip",
    "metric_fused.hip",
    "dedup.hip",
    "embedding.hip",
    "shard_route.hip",
    "hashing.hip",
    "softmax.hip",
    "softmax16.hip",
    "logits_ce.hip",
    "interaction.hip",
    "gemm16.hip",
]


# topk_scan16.hip: its max trees must compile to bare v_max3_f32 (no sNaN-quieting
# canonicalisation of the MFMA results); NaN inputs are outside the top-K contract anyway.
EXTRA_FLAGS = {"topk_scan16.hip": ["-fno-honor-nans"],
               # (the max trees over MFMA accumulators of the block-fed filters: v_max3_f32 without canonicalisation)
               "topk_raw.hip": ["-fno-honor-nans"],
               # MFMA accumulators in VGPRs: the softmax arithmetic reads them directly
               "softmax16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def hipcc() -> str:
  for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
    if cand and os.path.exists(cand):
      return cand
  raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _newer(src_paths, target) -> bool:
  if not os.path.exists(target):
    return True
  t = os.path.getmtime(target)
  return any(os.path.getmtime(p) > t for p in src_paths)


def _compile(src: str, force: bool) -> str:
  obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
  # every header of this directory (common.h, mfma_tile.h, ...): a stale gemm16.o / interaction.o after an edit of
  # mfma_tile.h was possible while only common.h was listed
  headers = sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h"))
  deps = [os.path.join(HERE, src), os.path.join(PKG, "..", "include", "tfrs_hip.h")] + headers
  if force or _newer(deps, obj):
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC"] + EXTRA_FLAGS.get(src, []) + [
           "-x", "hip", "-c", os.path.join(HERE, src), "-o", obj]
    subprocess.check_call(cmd)
  return obj


def build(force: bool = False, verbose: bool = False) -> str:
  os.makedirs(OBJ, exist_ok=True)
  srcs = [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
  with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
    objs = list(ex.map(lambda s: _compile(s, force), srcs))
  if force or _newer(objs, LIB):
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    if verbose:
      print("linked", LIB)
  return LIB


if __name__ == "__main__":
  print(build(force="--force" in sys.argv, verbose=True))
