# coding=utf-8
"""A/B: the row-local self projection of a GraphSAGE layer (x @ W_self) on a SECOND HIP stream while the neighbour aggregation
(memory-bound gather) runs on the first, against the same two kernels back to back — products shape."""
import json
import sys

import torch

sys.path.insert(0, ".")
import bench                                                          # noqa: E402
from tf_geometric_amd import synthetic, _lib as L                     # noqa: E402
from tf_geometric_amd.plan import CsrPlan, segment_reduce, gemm_bias_act   # noqa: E402

n, e, f = synthetic.WORKLOADS["products"]
ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=0))
plan = CsrPlan.build(ei, n, n)
x = torch.randn(n, f, device="cuda")
res = {}
for units in (128, 32, 256):
    w_self = torch.randn(f, units, device="cuda") * 0.1
    a_out = torch.empty(n, units, device="cuda")
    r_out = torch.empty(n, f, device="cuda")
    side = torch.cuda.Stream()

    def seq():
        gemm_bias_act(x, w_self, out=a_out)
        segment_reduce(plan, x, L.MEAN, out=r_out)

    def par():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            gemm_bias_act(x, w_self, out=a_out)
        segment_reduce(plan, x, L.MEAN, out=r_out)
        cur.wait_stream(side)

    def agg_only():
        segment_reduce(plan, x, L.MEAN, out=r_out)

    t = {"seq": [], "par": [], "agg_only": []}
    for rnd in range(4):
        for name, fn in (("seq", seq), ("par", par), ("agg_only", agg_only)):
            t[name].append(round(bench._time(fn, steps=10, warmup=2), 3))
    res["units%d" % units] = t
print(json.dumps(res))
