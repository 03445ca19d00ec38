# coding=utf-8
"""MaxPoolGraphSage(256, concat) at products shape: layer forward + backward (layer 0 form) and the pooling MLP's weight-gradient
launch alone.  One JSON line.  usage: python tools/r06/time_pool.py [tag]"""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import tf_geometric_amd as tfg
from tf_geometric_amd import synthetic, _lib as L, plan as P

def ev(fn, steps=3, warmup=2):
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

P.AUTO_STATIC_LAYOUT = False
n, e, f = synthetic.WORKLOADS["products"]
ei = L.as_i32(synthetic.synthetic_edge_stripe(n, e, seed=0))
x = torch.randn(n, f, device="cuda")
w1 = torch.ones(int(ei.shape[1]), device="cuda")
layer = tfg.layers.MaxPoolGraphSage(256, activation=tfg.relu)
layer._maybe_build([x]); layer.trainable(True)
cache = {}
g = torch.randn(n, 256, device="cuda")
def fb():
    for p_ in layer.parameters():
        p_.grad = None
    layer([x, ei, w1], cache=cache).backward(g)
res = {"tag": sys.argv[1] if len(sys.argv) > 1 else "", "maxpool_fwd_bwd_ms": ev(fb)}
with torch.no_grad():
    res["maxpool_forward_ms"] = ev(lambda: layer([x, ei, w1], cache=cache))
if len(sys.argv) > 2 and sys.argv[2] == "hidden":        # the hidden-layer form: x carries a gradient too (winner masks, source-major apply)
    xg = x.clone().requires_grad_(True)
    def fb2():
        for p_ in layer.parameters():
            p_.grad = None
        xg.grad = None
        layer([xg, ei, w1], cache=cache).backward(g)
    res["maxpool_hidden_layer_fwd_bwd_ms"] = ev(fb2)
print(json.dumps(res))
