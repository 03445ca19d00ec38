# coding=utf-8
"""One variant of the headline launch (GCN propagation, weighted segment-sum, F = 100) on the products-sized R-MAT graph — or
on the uniform graph — run K times, for rocprofv3 passes (why is R-MAT not faster than uniform?).

    python tools/rmat_pmc.py <variant> [calls]
variants: uniform | rmat (the plan's default hub policy) | rmat_thr1024 | rmat_thr4096 | rmat_nohub | rmat_row_order
Prints one JSON line: variant, calls, event-timed ms per call, graph statistics."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tf_geometric_amd as tfg                                  # noqa: E402
from tf_geometric_amd import synthetic, _lib as L, plan as P    # noqa: E402
from tf_geometric_amd.nn.conv.gcn import gcn_norm_adj           # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "rmat"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n, e, f = synthetic.WORKLOADS["products"]
n, e, f = int(os.environ.get("RMAT_N", n)), int(os.environ.get("RMAT_E", e)), int(os.environ.get("RMAT_F", f))   # A/B sweeps
dev = torch.device("cuda")
if variant.startswith("rmat_thr"):
    P.HUB_THRESHOLD = int(variant[len("rmat_thr"):])
elif variant == "rmat_nohub":
    P.HUB_THRESHOLD = 1 << 30
elif variant == "rmat_row_order":
    P.ROW_ORDER_MAX_F = 256
ei = L.as_i32(synthetic.synthetic_edge_stripe(n, e, seed=0)) if variant == "uniform" else synthetic.rmat_edges(n, e, 7, dev)
x = L.as_f32(synthetic.synthetic_feature_rows(n, f, seed=1))
_pad = int(os.environ.get("RMAT_LD_PAD", "0"))       # rows (F + pad) floats apart: is it the power-of-two row stride?
if _pad:
    _wide = torch.empty((n, f + _pad), dtype=torch.float32, device=dev)
    _wide[:, :f] = x
    x = _wide[:, :f]
normed = gcn_norm_adj(tfg.SparseMatrix(ei, None, [n, n]), sym=True)
plan = normed.plan
hub = plan.hub_info()
out = torch.empty((n, f), dtype=torch.float32, device=dev)
fn = lambda: P.segment_reduce(plan, x, L.SUM, w_csr=normed.w_csr, self_coef=normed.self_coef, out=out)   # noqa: E731
fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(calls):
    fn()
e1.record()
torch.cuda.synchronize()
deg = plan.in_degree()
# how concentrated the SOURCES are: share of edges whose source is among the top 1 % / 10 % most referenced nodes
src_cnt = torch.bincount(plan.col.long(), minlength=n).sort(descending=True).values.double()
tot = float(src_cnt.sum())
print(json.dumps({"variant": variant, "N": n, "F": f, "calls_timed": calls, "calls_total": calls + 1, "ms_per_call_events": e0.elapsed_time(e1) / calls,
                  "edges": int(ei.shape[1]), "max_in_degree": int(deg.max()), "empty_rows": int((deg == 0).sum()),
                  "hub_rows": 0 if hub is None else int(hub[0].shape[0]), "hub_chunks": 0 if hub is None else int(hub[2].shape[0]),
                  "hub_threshold": int(getattr(plan, "hub_threshold", 0)),
                  "row_order_applied": bool(P.USE_ROW_ORDER and f <= P.ROW_ORDER_MAX_F and plan.row_order() is not None),
                  "edge_share_of_top_1pct_sources": float(src_cnt[: n // 100].sum()) / tot,
                  "edge_share_of_top_10pct_sources": float(src_cnt[: n // 10].sum()) / tot,
                  "bytes_of_top_10pct_source_rows": (n // 10) * f * 4}))
