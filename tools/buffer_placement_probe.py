# coding=utf-8
"""Does the gather rate depend on WHICH allocation holds the source table?  The same products-shaped pass over several
equal-sized tables allocated one after another (with other allocations in between), each timed alternately."""
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
tables, junk = [], []
for i in range(6):
    tables.append(torch.randn(n, f, device="cuda"))
    junk.append(torch.empty((i + 1) * 37_000_000, dtype=torch.uint8, device="cuda"))     # odd-sized neighbours
k = torch.randn(f, f, device="cuda") * 0.1
tables.append(gemm_bias_act(tables[0], k, act=L.ACT_RELU))                               # a GEMM output (ReLU: half zeros)
tables.append(torch.relu(tables[1]))                                                     # same value pattern, elementwise
out = torch.empty(n, f, device="cuda")
res = {"table_ptr_mod_2MiB": [int(t.data_ptr() % (2 << 20)) for t in tables], "max_ms": [], "sum_ms": []}
for rnd in range(3):
    res["max_ms"].append([round(bench._time(lambda t=t: segment_reduce(plan, t, L.MAX, out=out), steps=8, warmup=2), 3) for t in tables])
    res["sum_ms"].append([round(bench._time(lambda t=t: segment_reduce(plan, t, L.SUM, out=out), steps=8, warmup=2), 3) for t in tables])
print(json.dumps(res))
