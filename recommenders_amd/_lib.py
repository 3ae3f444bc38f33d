"""ctypes binding of libtfrs_hip.so (the C ABI in include/tfrs_hip.h).

The product path has no CPU or eager fallback: if the shared library is missing or a
call fails, the caller gets an exception -- never a silently different code path.
"""

import ctypes
import os
from typing import Optional

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libtfrs_hip.so")

TFRS_OK, TFRS_EINVAL, TFRS_ENOTIMPL, TFRS_EHIP, TFRS_ENOMEM, TFRS_ESTATE = 0, -1, -2, -3, -4, -5

c_void_p, c_int, c_i64, c_size_t, c_float = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                             ctypes.c_size_t, ctypes.c_float)
P = c_void_p
c_u64 = ctypes.c_uint64

# name -> (restype, argtypes); mirrors include/tfrs_hip.h one to one.
SIGNATURES = {
    "tfrs_version": (c_int, []),
    "tfrs_last_error": (ctypes.c_char_p, []),
    "tfrs_device_info": (c_int, [c_int, P, P, P, c_int]),
    "tfrs_set_option": (c_int, [ctypes.c_char_p, ctypes.c_char_p]),
    "tfrs_get_option": (c_int, [ctypes.c_char_p, ctypes.c_char_p, c_int]),
    "tfrs_profile_enable": (c_int, [c_int]),
    "tfrs_profile_read": (c_int, [P, P, P]),
    "tfrs_profile_read_kind": (c_int, [c_int, P, P, P]),
    "tfrs_calibrate_workspace_bytes": (c_size_t, []),
    "tfrs_calibrate_mfma_f16": (c_int, [P, c_size_t, c_int, P, P, P]),
    "tfrs_calibrate_copy": (c_int, [P, c_size_t, c_int, P, P]),
    "tfrs_index_create": (c_int, [P]),
    "tfrs_index_destroy": (c_int, [P]),
    "tfrs_index_set": (c_int, [P, P, c_i64, c_int, P]),
    "tfrs_index_reserve": (c_int, [P, c_i64, c_int, P]),
    "tfrs_index_append": (c_int, [P, P, c_i64, P]),
    "tfrs_index_nonfinite": (c_int, [P, c_int, P]),
    "tfrs_index_note_nonfinite": (c_int, [P, P, c_i64, P, c_i64, c_int, P]),
    "tfrs_index_size": (c_i64, [P]),
    "tfrs_index_dim": (c_int, [P]),
    "tfrs_index_unpack": (c_int, [P, P, P]),
    "tfrs_bruteforce_topk_workspace_bytes": (c_size_t, [c_i64, c_i64, c_int, c_int]),
    "tfrs_bruteforce_topk": (c_int, [P, P, c_i64, c_int, P, P, P, c_size_t, P]),
    "tfrs_row_hash64": (c_int, [P, c_i64, c_int, P, P]),
    "tfrs_topk_expand_duplicates": (c_int, [P, P, c_i64, c_int, P, P, c_int, P, P, P]),
    "tfrs_bruteforce_topk_below_workspace_bytes": (c_size_t, [c_i64, c_i64, c_int, c_int]),
    "tfrs_bruteforce_topk_below": (c_int, [P, P, c_i64, c_int, P, P, c_i64, P, P, P, c_size_t, P]),
    "tfrs_bruteforce_topk_redo_count": (c_int, [P, c_i64, c_i64, c_int, P, P]),
    "tfrs_bruteforce_topk_redo_reasons": (c_int, [P, c_i64, c_i64, c_int, P, P]),
    "tfrs_debug_topk_plan": (c_int, [c_i64, c_int, c_int, P]),
    "tfrs_debug_fp16_scores": (c_int, [P, P, c_i64, c_i64, c_i64, P, P, P]),
    "tfrs_streaming_topk_workspace_bytes": (c_size_t, [c_i64, c_i64, c_int, c_int]),
    "tfrs_streaming_topk_update": (c_int, [P, c_i64, c_int, P, c_i64, c_i64, c_int, P, P,
                                           ctypes.c_int32, P, P, c_size_t, P]),
    "tfrs_streaming_topk_blocks_workspace_bytes": (c_size_t, [c_i64, c_i64, c_int, c_int]),
    "tfrs_streaming_topk_update_blocks": (c_int, [P, c_i64, c_int, P, P, c_int, c_i64, c_i64, c_int, P, P,
                                                  ctypes.c_int32, P, P, c_size_t, P]),
    "tfrs_compute_scores": (c_int, [P, P, c_i64, c_int, c_int, P, c_int, P, c_size_t, P]),
    "tfrs_topk_update_from_scores": (c_int, [P, c_i64, c_i64, c_i64, c_i64, c_int, P, P, ctypes.c_int32,
                                             P, P]),
    "tfrs_topk_merge_workspace_bytes": (c_size_t, [c_i64, c_int, c_int, c_int]),
    "tfrs_topk_merge": (c_int, [P, P, c_int, c_i64, c_int, c_int, P, P, P, c_size_t, P]),
    "tfrs_topk_merge_strided": (c_int, [P, P, c_int, c_i64, c_i64, c_int, c_int, P, P, P]),
    "tfrs_topk_exclude": (c_int, [P, P, c_i64, c_int, P, c_int, c_int, P, P, P]),
    "tfrs_rank_of_positive": (c_int, [P, P, c_i64, c_int, P, c_int, P, c_int, P, P]),
    "tfrs_id_match_topk": (c_int, [P, P, c_i64, c_int, P, c_int, P, P]),
    "tfrs_rank_count_accumulate": (c_int, [P, P, c_i64, c_int, P, P, c_int, c_i64, c_i64, P, c_int, P]),
    "tfrs_topk_hits_update": (c_int, [P, c_i64, P, c_int, P, P, P, P, P]),
    "tfrs_shard_route_workspace_bytes": (c_size_t, [c_i64, c_int]),
    "tfrs_shard_route_ids": (c_int, [P, c_int, c_i64, c_i64, c_i64, c_int, P, P, P, P, P, c_size_t, P]),
    "tfrs_embedding_gather_fwd": (c_int, [P, c_i64, c_int, P, c_int, c_i64, P, P, P]),
    "tfrs_embedding_segment_reduce_fwd": (c_int, [P, c_i64, c_int, P, P, c_int, P, c_i64,
                                                  c_int, P, P, P]),
    "tfrs_embedding_segment_reduce_bwd": (c_int, [P, c_int, P, c_int, P, c_i64, c_int, P, P]),
    "tfrs_hash_bucket_strong_ids": (c_int, [P, c_int, c_i64, c_i64, c_u64, c_u64, P, P]),
    "tfrs_hash_bucket_strong_bytes": (c_int, [P, P, c_i64, c_i64, c_u64, c_u64, P, P]),
    "tfrs_unified_embedding_fwd": (c_int, [P, c_int, P, P, c_i64, c_int, P, P, P, c_i64, c_int, P,
                                           P, P]),
    "tfrs_unified_embedding_fwd_multi": (c_int, [c_int, P, c_int, c_i64, P, P, P, c_i64, c_int, P, P, P, P, P]),
    "tfrs_embedding_scatter_add_bwd": (c_int, [P, P, P, c_i64, c_int, P, P, c_float, c_float,
                                               c_int, P]),
    "tfrs_embedding_scatter_add_workspace_bytes": (c_size_t, [c_i64]),
    "tfrs_embedding_scatter_add_unsorted": (c_int, [P, P, c_int, c_i64, c_int, c_i64, P, P, c_float,
                                                    c_float, c_int, P, c_size_t, P]),
    "tfrs_embedding_scatter_add_rowscan_multi": (c_int, [c_int, P, P, P, P, P, P, P, P, c_float, c_float,
                                                         c_int, P]),
    "tfrs_adagrad_dense_multi": (c_int, [c_int, P, P, P, P, c_float, c_float, c_int, P]),
    "tfrs_copy_multi": (c_int, [c_int, P, P, P, P]),
    "tfrs_embedding_scatter_add_rowscan": (c_int, [P, P, c_int, c_i64, c_int, c_i64, P, P, c_float,
                                                   c_float, c_int, P]),
    "tfrs_inbatch_softmax_workspace_bytes": (c_size_t, [c_i64, c_i64, c_int]),
    "tfrs_inbatch_softmax_ce_fwd": (c_int, [P, P, c_i64, c_i64, c_int, P, c_float, P, P, P,
                                            P, P, P, P, c_size_t, P]),
    "tfrs_inbatch_softmax_ce_bwd": (c_int, [P, P, c_i64, c_i64, c_int, P, c_float, P, P, P,
                                            P, P, P, P, P, c_size_t, c_int, P]),
    "tfrs_logits_ce_fwd": (c_int, [P, P, c_i64, c_i64, P, P, P, P, P]),
    "tfrs_logits_ce_bwd": (c_int, [P, P, c_i64, c_i64, P, P, P, P, P, P]),
    "tfrs_cross_fwd": (c_int, [P, P, P, P, c_float, c_i64, c_int, P, P]),
    "tfrs_cross_fwd_ex": (c_int, [P, P, P, c_int, P, P, c_float, c_i64, c_int, P, P]),
    "tfrs_cross_bwd_workspace_bytes": (c_size_t, [c_i64, c_int, c_int]),
    "tfrs_cross_bwd": (c_int, [P, P, P, P, c_float, P, c_i64, c_int, P, P, P, P, P, c_size_t, P]),
    "tfrs_cross_bwd_f16": (c_int, [P, P, P, P, c_float, P, c_i64, c_int, P, P, P, P, P, c_size_t, P]),
    "tfrs_dense_bwd_workspace_bytes": (c_size_t, [c_i64, c_int, c_int, c_int]),
    "tfrs_dense_bwd": (c_int, [P, P, P, c_i64, c_int, c_int, P, P, P, c_int, P, c_size_t, P]),
    "tfrs_dense_fwd": (c_int, [P, P, P, c_i64, c_int, c_int, P, P]),
    "tfrs_dense_fwd_act": (c_int, [P, P, P, c_i64, c_int, c_int, c_int, P, P, c_int, P, c_size_t, P]),
    "tfrs_cross_fwd_act": (c_int, [P, P, P, c_int, P, P, c_float, c_int, c_i64, c_int, P, P, c_int, P, c_size_t, P]),
    "tfrs_act_pointwise_bwd": (c_int, [c_int, c_int, P, P, P, P, c_float, c_i64, P, P, P, P]),
    "tfrs_dense_bwd_add": (c_int, [P, P, P, P, c_i64, c_int, c_int, P, P, P, c_int, P, c_size_t, P]),
    "tfrs_gemm_f16_workspace_bytes": (c_size_t, [c_i64, c_int, c_int]),
    "tfrs_dense_fwd_f16": (c_int, [P, P, P, c_i64, c_int, c_int, P, P, c_size_t, P]),
    "tfrs_cross_fwd_f16": (c_int, [P, P, P, P, c_float, c_i64, c_int, P, P, c_size_t, P]),
    "tfrs_cross_fwd_f16_train": (c_int, [P, P, P, P, c_float, c_i64, c_int, P, P, P, c_size_t, P]),
    "tfrs_cross_bwd_f16_saved": (c_int, [P, P, P, P, c_float, P, c_i64, c_int, P, P, P, P, P, c_size_t, P]),
    "tfrs_cross_bwd_f16_saved_acc": (c_int, [P, P, P, P, c_float, P, c_i64, c_int, P, c_int, P, P, P, P, P, c_size_t,
                                             P]),
    "tfrs_dot_interaction_fwd": (c_int, [P, c_i64, c_int, c_int, c_int, c_int, P, P]),
    "tfrs_dot_interaction_bwd": (c_int, [P, P, c_i64, c_int, c_int, c_int, c_int, P, P]),
    "tfrs_dot_interaction_fwd_strided": (c_int, [P, c_i64, c_int, c_int, c_int, P, c_i64, P]),
    "tfrs_dot_interaction_bwd_strided": (c_int, [P, P, c_i64, c_i64, c_int, c_int, c_int, P, P]),
    "tfrs_dot_interaction_strided_supported": (c_int, [c_i64, c_int, c_int, c_int]),
}

_lib: Optional[ctypes.CDLL] = None


class HipExtensionMissing(RuntimeError):
  pass


def load() -> ctypes.CDLL:
  """Loads the library (once) and declares every prototype."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise HipExtensionMissing(
        f"{LIB_PATH} is missing: build it with `python -m recommenders_amd.csrc.build` "
        "(or __graft_entry__.build()).  There is no CPU fallback.")
  lib = ctypes.CDLL(LIB_PATH)
  missing = [name for name in SIGNATURES if not hasattr(lib, name)]
  if missing:
    raise HipExtensionMissing(
        f"{LIB_PATH} does not export {missing}: rebuild it with "
        "`python -m recommenders_amd.csrc.build --force`.")
  ablated = [n for n in ("tfrs_ablation_build_scan16", "tfrs_ablation_build_raww", "tfrs_ablation_build_list16",
             "tfrs_ablation_build_g16")
             if hasattr(lib, n)]
  if ablated and os.environ.get("TFRS_ALLOW_ABLATION") != "1":
    raise HipExtensionMissing(
        f"{LIB_PATH} is an ABLATION build ({ablated}: kernels with parts switched off, wrong results by design); "
        "rebuild without the *_ABLATE macros or set TFRS_ALLOW_ABLATION=1 for a measurement run.")
  for name, (res, args) in SIGNATURES.items():
    fn = getattr(lib, name)
    fn.restype = res
    fn.argtypes = args
  _lib = lib
  return lib


def set_option(name: str, value: Optional[str]) -> None:
  """One of the library's TFRS_* switches, overriding the environment; ``None``
  removes the override."""
  check(load().tfrs_set_option(name.encode(), None if value is None else str(value).encode()))


def get_option(name: str) -> Optional[str]:
  buf = ctypes.create_string_buffer(256)
  return buf.value.decode() if load().tfrs_get_option(name.encode(), buf, 256) == 1 else None


def last_error() -> str:
  return load().tfrs_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
  """Maps a status code to the exception type the reference raises for that case."""
  if rc == TFRS_OK:
    return
  msg = last_error()
  if rc in (TFRS_EINVAL, TFRS_ESTATE):
    raise ValueError(msg)
  if rc == TFRS_ENOTIMPL:
    raise NotImplementedError(msg)
  raise RuntimeError(f"libtfrs_hip error {rc}: {msg}")


def ptr(t) -> c_void_p:
  """Device pointer of a torch tensor (or None)."""
  if t is None:
    return c_void_p(0)
  return c_void_p(t.data_ptr())


def current_stream() -> c_void_p:
  import torch
  return c_void_p(torch.cuda.current_stream().cuda_stream)
