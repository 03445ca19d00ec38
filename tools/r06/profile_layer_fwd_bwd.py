# coding=utf-8
"""forward + backward of ONE layer in the form bench.py times it (layer 0: only the weights carry gradients), a few repeats —
run under `rocprofv3 --kernel-trace --stats`.  usage: profile_layer_fwd_bwd.py gat8|gat64|maxpool|meansage [repeats]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import tf_geometric_amd as tfg
from tf_geometric_amd import synthetic, _lib as L, plan as P

which = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
P.AUTO_STATIC_LAYOUT = False
if which.startswith("gat"):
    n, e, f = synthetic.WORKLOADS["reddit"]
    ei = L.as_i32(synthetic.synthetic_edge_stripe(n, e, seed=3))
    x = torch.randn(n, f, device="cuda")
    layer = tfg.layers.GAT(64, attention_units=int(which[3:]), num_heads=8, activation=tfg.relu)
    inputs = [x, ei]
else:
    n, e, f = synthetic.WORKLOADS["products"]
    ei = L.as_i32(synthetic.synthetic_edge_stripe(n, e, seed=0))
    x = torch.randn(n, f, device="cuda")
    w1 = torch.ones(int(ei.shape[1]), device="cuda")
    layer = (tfg.layers.MaxPoolGraphSage if which == "maxpool" else tfg.layers.MeanGraphSage)(256, activation=tfg.relu)
    inputs = [x, ei, w1]
layer._maybe_build([x])
layer.trainable(True)
cache = {}
for _ in range(reps + 2):
    for p_ in layer.parameters():
        p_.grad = None
    layer(inputs, cache=cache).sum().backward()
torch.cuda.synchronize()
print("done", which)
