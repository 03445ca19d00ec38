# coding=utf-8
"""Power-law (R-MAT) vs uniform graph of the same size: forward and forward + backward of the layers whose kernels walk
one destination / source row per lane group — a hub row of 10^5 edges must not serialise a launch."""
import gc
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tf_geometric_amd as tfg
from tf_geometric_amd import synthetic, _lib as L

n, e, f = 1 << 20, 30000000, 64


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


graphs = {"uniform": L.as_i32(synthetic.synthetic_edges(n, e, seed=0)), "rmat": synthetic.rmat_edges(n, e, 0, torch.device("cuda"))}
for gname, ei in graphs.items():
    E = int(ei.shape[1])
    deg = torch.bincount(ei[0].long(), minlength=n)
    w = torch.rand(E, device="cuda") + 0.5
    x = torch.randn(n, f, device="cuda")
    for lname, make in [("GCN(64)", lambda: tfg.layers.GCN(64, activation=tfg.relu)),
                        ("GAT(64,H8,A8)", lambda: tfg.layers.GAT(64, num_heads=8, attention_units=8)),
                        ("MaxPoolGraphSage(32)", lambda: tfg.layers.MaxPoolGraphSage(32, activation=tfg.relu)),
                        ("MeanGraphSage(64)", lambda: tfg.layers.MeanGraphSage(64, activation=tfg.relu))]:
        cache = {}
        layer = make()
        xin = x.clone().requires_grad_(False)
        fwd = t(lambda: layer([xin, ei, w], cache=cache))
        layer.trainable(True)
        xg = x.clone().requires_grad_(True)

        def step():
            for p_ in layer.parameters():
                p_.grad = None
            xg.grad = None
            layer([xg, ei, w], cache=cache).sum().backward()
        tr = t(step, k=2)
        print(json.dumps({"graph": gname, "max_in_degree": int(deg.max()), "E": E, "layer": lname, "fwd_ms": round(fwd, 3),
                          "fwd_bwd_ms": round(tr, 3)}), flush=True)
