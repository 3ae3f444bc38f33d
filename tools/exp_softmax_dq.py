"""softmax_f16.sizes.dq at (nq, nc, d, scale) = (200, 16500, 32, 0.3): which entries carry the largest error relative to
their own sum of |terms|, for 4 / 8 waves per workgroup and the all-f32 kernels, and how that error splits between the
recomputed logits (first GEMM) and the gradient product (second GEMM) -- VERDICT round 5, weak 2."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import retrieval as o_ret
from recommenders_amd.tasks.retrieval import in_batch_softmax_loss

nq, nc, d, scale = 200, 16500, 32, 0.3
rng = np.random.default_rng(nq + d)
cscale = 2.0 / (scale * np.sqrt(d))
q = (rng.normal(size=(nq, d)) * scale * np.exp(0.5 * rng.normal(size=(nq, 1)))).astype(np.float32)
c = (rng.normal(size=(nc, d)) * cscale * np.exp(0.5 * rng.normal(size=(nc, 1)))).astype(np.float32)
q[5] = 0.0
w = (10.0 ** rng.uniform(-2, 2, size=nq)).astype(np.float32)
t = lambda a: torch.as_tensor(a).cuda()
for kw in (dict(), dict(sample_weight=w, temperature=0.5)):
  dq_ref, dc_ref, dq_y, dc_y = o_ret.loss_grads(q, c, return_yardsticks=True, **kw)
  # float64 logits: how large are they, and their sum of |terms|
  s64 = q.astype(np.float64) @ c.astype(np.float64).T * (1.0 / kw.get("temperature", 1.0))
  sabs = np.abs(q.astype(np.float64)) @ np.abs(c.astype(np.float64)).T * (1.0 / kw.get("temperature", 1.0))
  for mode in ("f16/4", "f16/8", "f32"):
    os.environ.pop("TFRS_SOFTMAX_NW", None); os.environ.pop("TFRS_SOFTMAX_MODE", None)
    if mode.startswith("f16"):
      os.environ["TFRS_SOFTMAX_NW"] = mode[-1]
    else:
      os.environ["TFRS_SOFTMAX_MODE"] = "f32"
    tq, tc = t(q).requires_grad_(True), t(c).requires_grad_(True)
    loss = in_batch_softmax_loss(tq, tc, sample_weight=None if not kw else t(w), temperature=kw.get("temperature"))
    loss.backward()
    g = tq.grad.cpu().numpy().astype(np.float64)
    ratio = np.abs(g - dq_ref) / np.maximum(np.abs(dq_y), 1e-30)
    i, k = np.unravel_index(np.argmax(ratio), ratio.shape)
    rows = np.sort(ratio.max(axis=1))[::-1]
    print(json.dumps({"kw": sorted(kw), "mode": mode, "dq_max_ratio": float(ratio.max()), "at": [int(i), int(k)],
                      "row_weight": float(w[i]) if kw else 1.0, "row_norm": float(np.linalg.norm(q[i])),
                      "abs_logit_max_row": float(np.abs(s64[i]).max()), "sum_abs_terms_logit_max_row": float(sabs[i].max()),
                      "ratio_p50_p90_p99": [float(np.quantile(ratio, x)) for x in (0.5, 0.9, 0.99)],
                      "worst_rows": [float(x) for x in rows[:4]],
                      "err_model_2^-22*(3+sum|S terms|)": float(2.0 ** -22 * (3.0 + sabs[i].max()))}), flush=True)
