# coding=utf-8
"""Same edge count, very different densities: 30 M edges over 10^4 .. 10^7 nodes (average in-degree 3000 .. 3) — ns per
edge of the layers' forward and forward + backward must stay flat-ish; a density regime that falls off is a cliff."""
import gc
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tf_geometric_amd as tfg
from tf_geometric_amd import synthetic, _lib as L

e, f = 30000000, 64


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


for n in (10000, 100000, 1000000, 10000000):
    ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=0))
    E = int(ei.shape[1])
    w = torch.rand(E, device="cuda") + 0.5
    x = torch.randn(n, f, device="cuda")
    row = {"n": n, "avg_degree": round(E / n, 1)}
    for lname, make in [("GCN", lambda: tfg.layers.GCN(64, activation=tfg.relu)),
                        ("GAT", lambda: tfg.layers.GAT(64, num_heads=8, attention_units=8)),
                        ("MaxPoolSage", lambda: tfg.layers.MaxPoolGraphSage(32, activation=tfg.relu))]:
        cache = {}
        layer = make()
        fwd = t(lambda: layer([x, ei, w], cache=cache))
        layer.trainable(True)
        xg = x.clone().requires_grad_(True)

        def step():
            for p_ in layer.parameters():
                p_.grad = None
            xg.grad = None
            layer([xg, ei, w], cache=cache).sum().backward()
        tr = t(step, k=2)
        row[lname] = [round(fwd, 3), round(tr, 3)]
    print(json.dumps(row), flush=True)
    del ei, w, x
