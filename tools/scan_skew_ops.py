# coding=utf-8
"""R-MAT vs uniform graph (2^20 nodes, 30 M edges): the row-walking operators around the path — standalone segment
softmax, d/dw of the weighted sum (SDDMM), the neighbour sampler — must not serialise on a hub row."""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tf_geometric_amd as tfg
from tf_geometric_amd import synthetic, _lib as L

n, e = 1 << 20, 30000000


def t(fn, k=3):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / k


graphs = {"uniform": L.as_i32(synthetic.synthetic_edges(n, e, seed=0)), "rmat": synthetic.rmat_edges(n, e, 0, torch.device("cuda"))}
for gname, ei in graphs.items():
    E = int(ei.shape[1])
    w = torch.rand(E, device="cuda") + 0.5
    att = torch.randn(E, 8, device="cuda")
    row = {"graph": gname}
    row["segment_softmax[E,8]"] = round(t(lambda: tfg.nn.segment_softmax(att, ei[0], n)), 3)
    x = torch.randn(n, 64, device="cuda")
    wg = w.clone().requires_grad_(True)

    def step():
        wg.grad = None
        tfg.nn.aggregate_neighbors(x, ei, wg, tfg.nn.gcn_mapper, tfg.nn.sum_reducer, tfg.nn.identity_updater).sum().backward()
    row["sum d/dw (sddmm)"] = round(t(step), 3)
    s = tfg.utils.RandomNeighborSampler(ei, w)
    row["sampler k=10"] = round(t(lambda: s.sample(k=10, seed=1)), 3)
    row["sampler ratio=0.5"] = round(t(lambda: s.sample(ratio=0.5, seed=1)), 3)
    row["sampler k=2000 (keep-all rows)"] = round(t(lambda: s.sample(k=2000, seed=1)), 3)
    print(json.dumps(row), flush=True)
