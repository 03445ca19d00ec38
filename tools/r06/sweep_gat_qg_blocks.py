# coding=utf-8
"""Reddit-shaped GAT(64, H 8, A 8) layer forward + backward by the number of source blocks of the TRAINING forward (the kernel
that also accumulates the query gradient's sums), the backward's source pass pinned to its 14 destination blocks.  JSON lines."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import tf_geometric_amd as tfg
from tf_geometric_amd import synthetic, _lib as L
from tf_geometric_amd.nn.conv import gat as G

def ev(fn, steps=8, warmup=3):
    for _ in range(warmup):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); a.record()
        for _ in range(steps):
            fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / steps)
    return min(ts)

n, e, f = synthetic.WORKLOADS["reddit"]
ei = L.as_i32(synthetic.synthetic_edge_stripe(n, e, seed=3))
x = torch.randn(n, f, device="cuda")
cache = {}
tl = tfg.layers.GAT(64, attention_units=8, num_heads=8, activation=tfg.relu)
tl._maybe_build([x]); tl.trainable(True)
g = torch.randn(n, 64, device="cuda")
def fb():
    for p_ in tl.parameters():
        p_.grad = None
    tl([x, ei], cache=cache).backward(g)
G.DESTINATION_BLOCKS = 14
for kb in (None, 7, 8, 9, 10, 11, 12, 13):
    G.SOURCE_BLOCKS = kb
    print(json.dumps({"forward_source_blocks": kb if kb is not None else "policy", "fwd_bwd_ms": round(ev(fb), 3)}), flush=True)
