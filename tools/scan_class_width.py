# coding=utf-8
"""2-layer GCN full-batch training step at products shape for several class counts (the width of the LAST layer): odd widths
(ogbn-products has 47 classes) against their neighbours.  One JSON line per width."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tf_geometric_amd as tfg                                   # noqa: E402
from tf_geometric_amd import synthetic, _lib as L               # noqa: E402
import bench                                                     # noqa: E402

n, e, f = synthetic.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "products"]
ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=0))
x = L.as_f32(synthetic.synthetic_features(n, f, seed=1))
idx = torch.arange(0, n, 10, device="cuda")
kind = sys.argv[3] if len(sys.argv) > 3 else "gcn"
w1 = torch.ones(int(ei.shape[1]), device="cuda")
for classes in [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "40,47,48,41,7")]:
    cache = {}
    if kind == "gcn":
        l0, l1 = tfg.layers.GCN(256, activation=tfg.relu), tfg.layers.GCN(classes)
        g0, g1 = (lambda inp, cache: l0(inp, cache=cache)), (lambda inp, cache: l1(inp, cache=cache))
    elif kind == "sage":
        l0, l1 = tfg.layers.MeanGraphSage(256, activation=tfg.relu), tfg.layers.MeanGraphSage(classes, activation=None, concat=False)
        g0, g1 = (lambda inp, cache: l0(inp + [w1], cache=cache)), (lambda inp, cache: l1(inp + [w1], cache=cache))
    elif kind == "maxpool":
        l0, l1 = tfg.layers.MaxPoolGraphSage(64, activation=tfg.relu), tfg.layers.MaxPoolGraphSage(classes, activation=None, concat=False)
        g0, g1 = (lambda inp, cache: l0(inp + [w1], cache=cache)), (lambda inp, cache: l1(inp + [w1], cache=cache))
    else:
        l0 = tfg.layers.GAT(64, attention_units=8, num_heads=8, activation=tfg.relu)
        l1 = tfg.layers.GAT(classes, attention_units=1, num_heads=1)
        g0, g1 = (lambda inp, cache: l0(inp, cache=cache)), (lambda inp, cache: l1(inp, cache=cache))
    with torch.no_grad():
        g1([g0([x, ei], cache=cache), ei], cache=cache)
    l0.trainable(True)
    l1.trainable(True)
    opt = torch.optim.Adam(l0.parameters() + l1.parameters(), lr=1e-2)
    labels = torch.randint(0, classes, (int(idx.shape[0]),), device="cuda")

    def step():
        opt.zero_grad(set_to_none=True)
        logits = g1([g0([x, ei], cache=cache), ei], cache=cache)
        torch.nn.functional.cross_entropy(logits[idx], labels).backward()
        opt.step()

    def fwd():
        with torch.no_grad():
            return g1([g0([x, ei], cache=cache), ei], cache=cache)

    print(json.dumps({"kind": kind, "classes": classes, "train_step_ms": bench._time(step, steps=5, warmup=3), "forward_ms": bench._time(fwd, steps=5, warmup=2)}), flush=True)
