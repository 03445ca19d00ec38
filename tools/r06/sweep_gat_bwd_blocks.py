# coding=utf-8
"""Reddit-shape GAT(64, H8, A) forward + backward by the number of destination blocks of the source pass (dK, dV)."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import tf_geometric_amd as tfg
from tf_geometric_amd import synthetic, _lib as L
from tf_geometric_amd.nn.conv import gat as G_

def ev(fn, steps=6, warmup=2):
    for _ in range(warmup):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); a.record()
        for _ in range(steps):
            fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / steps)
    return min(ts)

n, e, f = synthetic.WORKLOADS["reddit"]
ei = L.as_i32(synthetic.synthetic_edge_stripe(n, e, seed=3))
x = torch.randn(n, f, device="cuda")
cache = {}
for A in (8, 64):
    tl = tfg.layers.GAT(64, attention_units=A, num_heads=8, activation=tfg.relu)
    tl._maybe_build([x]); tl.trainable(True)
    def fb():
        for p_ in tl.parameters():
            p_.grad = None
        tl([x, ei], cache=cache).sum().backward()
    row = {"A": A}
    for kb in (None, 6, 8, 10, 12, 14, 16, 20, 24):
        G_.DESTINATION_BLOCKS = kb
        row["policy" if kb is None else str(kb)] = round(ev(fb), 3)
    G_.DESTINATION_BLOCKS = None
    print(json.dumps(row))
