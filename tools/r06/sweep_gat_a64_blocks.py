# coding=utf-8
"""Reddit shape, attention_units = 64 (d_head = 8): attention alone and layer forward + backward by the number of source blocks."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import tf_geometric_amd as tfg
from tf_geometric_amd import synthetic, _lib as L
from tf_geometric_amd.plan import CsrPlan
from tf_geometric_amd.nn.conv import gat as G_
from tf_geometric_amd.nn.conv.gat import gat_attention

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
plan = CsrPlan.from_cache(ei, n, n, cache)
for A in (64, 8):
    Q, K, V = torch.randn(n, A, device="cuda"), torch.randn(n, A, device="cuda"), torch.randn(n, 64, device="cuda")
    tl = tfg.layers.GAT(64, attention_units=A, num_heads=8, activation=tfg.relu)
    tl._maybe_build([x]); tl.trainable(True)
    def fb():
        for p_ in tl.parameters():
            p_.grad = None
        tl([x, ei], cache=cache).sum().backward()
    for kb in (None, 8, 11, 13, 15, 18, 20, 24, 28):
        G_.SOURCE_BLOCKS = kb
        G_.DESTINATION_BLOCKS = None
        row = {"A": A, "source_blocks": "policy" if kb is None else kb,
               "policy_blocks": G_.source_block_count(plan, A, 64) if kb is None else None,
               "attention_ms": round(ev(lambda: gat_attention(plan, Q, K, V, 8)), 3), "fwd_bwd_ms": round(ev(fb, steps=4), 3)}
        print(json.dumps(row), flush=True)
    G_.SOURCE_BLOCKS = None
