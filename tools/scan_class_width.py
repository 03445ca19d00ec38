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
for classes in [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "40,47,48,41,7")]:
    cache = {}
    g0, g1 = tfg.layers.GCN(256, activation=tfg.relu), tfg.layers.GCN(classes)
    with torch.no_grad():
        g1([g0([x, ei], cache=cache), ei], cache=cache)
    g0.trainable(True)
    g1.trainable(True)
    opt = torch.optim.Adam(g0.parameters() + g1.parameters(), lr=1e-2)
    labels = torch.randint(0, classes, (int(idx.shape[0]),), device="cuda")

    def step():
        opt.zero_grad(set_to_none=True)
        logits = g1([g0([x, ei], cache=cache), ei], cache=cache)
        torch.nn.functional.cross_entropy(logits[idx], labels).backward()
        opt.step()

    def fwd():
        with torch.no_grad():
            return g1([g0([x, ei], cache=cache), ei], cache=cache)

    print(json.dumps({"classes": classes, "train_step_ms": bench._time(step, steps=5, warmup=3), "forward_ms": bench._time(fwd, steps=5, warmup=2)}), flush=True)
