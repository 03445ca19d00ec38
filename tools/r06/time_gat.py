# coding=utf-8
"""Attention alone / GAT layer forward / forward + backward at Reddit shape (A = 8 and 64), and the one-pass d_head = 8 walk on
a products-density graph.  One JSON line.  usage: python tools/r06/time_gat.py [tag]"""
import sys, os, json, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import tf_geometric_amd as tfg
from tf_geometric_amd import synthetic, _lib as L
from tf_geometric_amd.plan import CsrPlan
from tf_geometric_amd.nn.conv.gat import gat_attention

def ev(fn, steps=10, warmup=3):
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

res = {"tag": sys.argv[1] if len(sys.argv) > 1 else ""}
n, e, f = synthetic.WORKLOADS["reddit"]
ei = L.as_i32(synthetic.synthetic_edge_stripe(n, e, seed=3))
x = torch.randn(n, f, device="cuda")
cache = {}
plan = CsrPlan.from_cache(ei, n, n, cache)
for A in (8, 64):
    Q, K, V = torch.randn(n, A, device="cuda"), torch.randn(n, A, device="cuda"), torch.randn(n, 64, device="cuda")
    res["reddit_A{}_attention_ms".format(A)] = ev(lambda: gat_attention(plan, Q, K, V, 8))
    lay = tfg.layers.GAT(64, attention_units=A, num_heads=8, activation=tfg.relu)
    res["reddit_A{}_layer_forward_ms".format(A)] = ev(lambda: lay([x, ei], cache=cache), steps=6)
    tl = tfg.layers.GAT(64, attention_units=A, num_heads=8, activation=tfg.relu)
    tl._maybe_build([x]); tl.trainable(True)
    def fb():
        for p_ in tl.parameters():
            p_.grad = None
        tl([x, ei], cache=cache).sum().backward()
    res["reddit_A{}_fwd_bwd_ms".format(A)] = ev(fb, steps=4, warmup=2)
del x, plan, cache
n2, e2, _ = synthetic.WORKLOADS["products"]
ei2 = L.as_i32(synthetic.synthetic_edge_stripe(n2, e2, seed=0))
plan2 = CsrPlan.build(ei2, n2, n2)
for A in (8, 64):
    Q, K, V = torch.randn(n2, A, device="cuda"), torch.randn(n2, A, device="cuda"), torch.randn(n2, 64, device="cuda")
    res["products_density_A{}_attention_one_pass_ms".format(A)] = ev(lambda: gat_attention(plan2, Q, K, V, 8), steps=5)
print(json.dumps(res))
