import sys, json, torch
sys.path.insert(0, ".")
import tf_geometric_amd as tfg
from tf_geometric_amd import synthetic, _lib as L
n, e, f = 233000, 114000000, 64
ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=0))
x = torch.randn(n, f, device="cuda")
cache = {}
def t(fn, k=5):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k
for units, heads, att in [(41, 1, 1), (40, 1, 1), (44, 1, 1), (41, 1, 4), (64, 8, 8), (47, 1, 1), (7, 1, 1), (16, 1, 1)]:
    layer = tfg.layers.GAT(units, num_heads=heads, attention_units=att)
    ms = t(lambda: layer([x, ei], cache=cache))
    print(json.dumps({"units": units, "heads": heads, "attention_units": att, "layer_ms": round(ms, 3)}))
