"""What a wrong result of a TFRS_SCAN16_PEEL build looks like: the fp16-filtered top-100 over 1 M x 64 against the f32
path on the same batch -- how many queries differ, whether entries are missing or foreign, where in the list."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recommenders_amd import _lib
from recommenders_amd.layers import factorized_top_k as ftk

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(42)
n = int(os.environ.get("N", 1_000_000))
corpus = torch.randn((n, 64), generator=g, device=dev) / 8.0
queries = torch.randn((int(os.environ.get("NQ", 8192)), 64), generator=g, device=dev) / 8.0
index = ftk.BruteForce(k=100).index(corpus)
_lib.set_option("TFRS_TOPK_FILTER", "f32")
s32, i32 = index(queries)
_lib.set_option("TFRS_TOPK_FILTER", None)
for shape in sys.argv[1:] or [None]:
  _lib.set_option("TFRS_SCAN16_SHAPE", shape)
  s, i = index(queries)
  s2, i2 = index(queries)
  _lib.set_option("TFRS_SCAN16_SHAPE", None)
  torch.cuda.synchronize()
  badq = ((i != i32).any(1) | (s != s32).any(1)).nonzero().flatten()
  out = {"shape": shape, "bad_queries": int(badq.numel()), "repeatable": bool(torch.equal(i, i2)),
         "nan_scores": int(torch.isnan(s).sum()), "neg_idx": int((i < 0).sum()), "idx_ge_n": int((i >= n).sum())}
  if badq.numel():
    out["bad_query_ids_head"] = badq[:24].tolist()
    out["bad_mod_512_hist"] = torch.bincount((badq % 512) // 32, minlength=16).tolist()
    det = []
    for qi in badq[:6].tolist():
      exp, got = set(i32[qi].tolist()), set(i[qi].tolist())
      first = int((i[qi] != i32[qi]).nonzero()[0])
      missing = sorted(exp - got)
      det.append({"q": qi, "first_diff_pos": first, "n_missing": len(missing), "n_foreign": len(got - exp),
                  "missing_head": missing[:4], "missing_mod128": [m % 128 for m in missing[:8]],
                  "missing_stage_mod": [(m // 128) % 4 for m in missing[:8]],
                  "got_at": [int(i[qi][first]), float(s[qi][first])], "exp_at": [int(i32[qi][first]), float(s32[qi][first])]})
    out["detail"] = det
  print(json.dumps(out), flush=True)
