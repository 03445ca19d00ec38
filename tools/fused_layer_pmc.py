# coding=utf-8
"""The GCN layer 100 -> 256 (+ bias, ReLU) at products shape, three times as ONE fused launch (agg_gemm_kernel) and three
times as the two launches it replaces (seg_reduce_kernel + gemm_rows_kernel) — run under `rocprofv3 --pmc FETCH_SIZE` /
`--pmc WRITE_SIZE` (separate passes) to see the [N, F] round trip of the aggregate disappear from the HBM-side counters
(tools/profile_round.sh -> profiles/r03_rocprof.md)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tf_geometric_amd as tfg                                   # noqa: E402
from tf_geometric_amd import synthetic, _lib as L, plan as P     # noqa: E402

n, e, f = synthetic.WORKLOADS["products"]
if len(sys.argv) > 1:                      # feature width override (128: half of the kernel streamed from L2, round 4)
    f = int(sys.argv[1])
P.AUTO_STATIC_LAYOUT = False               # the plain table in every call (no promotion on the layer's second call)
ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=0))
x = torch.randn(n, f, device="cuda")
cache = {}
gcn = tfg.layers.GCN(256, activation=tfg.relu)
gcn([x, ei], cache=cache)
torch.cuda.synchronize()
for fuse in (True, False):
    P.FUSE_AGGREGATE_GEMM = fuse
    for _ in range(3):
        gcn([x, ei], cache=cache)
    torch.cuda.synchronize()
print("done")
