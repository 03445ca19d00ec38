# coding=utf-8
"""Weighted segment-sum at products shape for an odd class width (47) against padded layouts of the same table:
F = 47 rows 188 B apart (dword loads, rows straddle 2-3 lines), the same 47 columns in rows 192 / 256 B apart, and F = 48 / 64
tables (16-byte loads).  One JSON line per layout."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tf_geometric_amd as tfg                                   # noqa: E402
from tf_geometric_amd import synthetic, _lib as L, plan as P     # noqa: E402
from tf_geometric_amd.nn.conv.gcn import gcn_norm_adj            # noqa: E402
import bench                                                     # noqa: E402

n, e, f = synthetic.WORKLOADS["products"]
ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=0))
adj = tfg.SparseMatrix(ei, None, [n, n])
normed = gcn_norm_adj(adj, cache={})
widths = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "47,40,41")]
for fw in widths:
    base = torch.randn(n, 64, device="cuda")
    for name, f_use, ld_in, ld_out in [("dense", fw, fw, fw), ("in_ld48", fw, (fw + 3) // 4 * 4, fw), ("in_ld64", fw, 64, fw),
                                       ("in_out_ld64", fw, 64, 64), ("F_pad4", (fw + 3) // 4 * 4, (fw + 3) // 4 * 4, (fw + 3) // 4 * 4),
                                       ("F_pad4_ld64", (fw + 3) // 4 * 4, 64, 64), ("F64", 64, 64, 64)]:
        xs = torch.zeros(n, ld_in, device="cuda")
        xs[:, :fw] = base[:, :fw]
        x = xs[:, :f_use]
        outs = torch.empty(n, ld_out, device="cuda")
        out = outs[:, :f_use]
        fn = lambda: P.segment_reduce(normed.plan, x, L.SUM, w_csr=normed.w_csr, self_coef=normed.self_coef, out=out)  # noqa: E731
        ms = bench._time(fn, steps=10, warmup=3)
        print(json.dumps({"width": fw, "layout": name, "F": f_use, "ld_in": ld_in, "ld_out": ld_out, "ms": ms,
                          "kernel": P.segment_reduce(normed.plan, x, L.SUM, w_csr=normed.w_csr, self_coef=normed.self_coef, out=out,
                                                     describe=True)}), flush=True)
