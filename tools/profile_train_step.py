# coding=utf-8
"""One full-batch training step of the 2-layer GCN / mean-SAGE models at products shape, repeated a few times — run under
`rocprofv3 --kernel-trace --stats` to see where a step's time goes (tools/rocpd_summary.py over the result)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tf_geometric_amd as tfg
from tf_geometric_amd import synthetic, _lib as L

which = sys.argv[1] if len(sys.argv) > 1 else "gcn"
n, e, f = synthetic.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "products"]
ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=0))
x = torch.randn(n, f, device="cuda")
w1 = torch.ones(int(ei.shape[1]), device="cuda")
cache = {}
if which == "gcn":
    l0, l1 = tfg.layers.GCN(256, activation=tfg.relu), tfg.layers.GCN(40)
    fwd = lambda: l1([l0([x, ei], cache=cache), ei], cache=cache)      # noqa: E731
else:
    l0, l1 = tfg.layers.MeanGraphSage(256, activation=tfg.relu), tfg.layers.MeanGraphSage(40, activation=None)
    fwd = lambda: l1([l0([x, ei, w1], cache=cache), ei, w1], cache=cache)      # noqa: E731
l0.trainable(True)
l1.trainable(True)
with torch.no_grad():
    fwd()
opt = torch.optim.Adam(l0.parameters() + l1.parameters(), lr=1e-2)
idx = torch.arange(0, n, 10, device="cuda")
labels = torch.randint(0, 40, (int(idx.shape[0]),), device="cuda")
for _ in range(6):
    opt.zero_grad(set_to_none=True)
    torch.nn.functional.cross_entropy(fwd()[idx], labels).backward()
    opt.step()
torch.cuda.synchronize()
print("done", which)
