# coding=utf-8
"""Scan aggregation widths for performance cliffs: forward and forward + backward (d/dx and d/dw) of sum / mean / max
over a 100 k-node / 10 M-edge graph, ns per edge per 32 columns (so equal numbers mean equal efficiency)."""
import gc
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tf_geometric_amd as tfg
from tf_geometric_amd import synthetic, _lib as L

n, e = 100000, 10000000
ei_np = synthetic.synthetic_edges(n, e, seed=0)
ei = L.as_i32(ei_np)
E = int(ei.shape[1])
w = torch.rand(E, device="cuda") + 0.5
cache = {}


def t(fn, k=3):
    for _ in range(2):
        fn()
    gc.collect()
    gc.disable()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k):
        fn()
    b.record()
    torch.cuda.synchronize()
    gc.enable()
    return a.elapsed_time(b) / k


for f in [1, 3, 7, 8, 20, 32, 41, 47, 64, 100, 101, 130, 256, 300]:
    row = {"F": f}
    lines = max(1.0, 4.0 * f / 128.0)
    for name, red in [("sum", tfg.nn.sum_reducer), ("mean", tfg.nn.mean_reducer), ("max", tfg.nn.max_reducer)]:
        x = torch.randn(n, f, device="cuda")
        fwd = t(lambda: tfg.nn.aggregate_neighbors(x, ei, w, tfg.nn.gcn_mapper, red, tfg.nn.identity_updater))
        xg = x.clone().requires_grad_(True)
        wg = w.clone().requires_grad_(name != "max" or True)

        def step(weights):
            xg.grad = None
            if weights.requires_grad:
                weights.grad = None
            tfg.nn.aggregate_neighbors(xg, ei, weights, tfg.nn.gcn_mapper, red, tfg.nn.identity_updater).sum().backward()
        bx = t(lambda: step(w), k=2)
        bxw = t(lambda: step(wg), k=2)
        row[name] = [round(fwd * 1e6 / E / lines, 3), round(bx * 1e6 / E / lines, 3), round(bxw * 1e6 / E / lines, 3)]
    print(json.dumps(row), flush=True)
