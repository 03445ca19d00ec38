# coding=utf-8
"""The C2 training step bench.py times (GCN(256, relu) -> GCN(40), cross-entropy on 10 % of the nodes, Adam) a few times — run under
`rocprofv3 --kernel-trace --stats`; tools/rocpd_sequence.py lists the last step's launches."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import tf_geometric_amd as tfg
from tf_geometric_amd import synthetic, _lib as L

na, ea, fa = synthetic.WORKLOADS["arxiv"]
eia = L.as_i32(synthetic.synthetic_edge_stripe(na, ea, seed=0))
xa = L.as_f32(synthetic.synthetic_feature_rows(na, fa, seed=1))
ca = {}
g0, g1 = tfg.layers.GCN(256, activation=tfg.relu), tfg.layers.GCN(40)
fwd = lambda: g1([g0([xa, eia], cache=ca), eia], cache=ca)
fwd()
g0.trainable(True); g1.trainable(True)
opt = torch.optim.Adam(g0.parameters() + g1.parameters(), lr=1e-2)
idx = torch.arange(0, na, 10, device="cuda")
labels = torch.randint(0, 40, (int(idx.shape[0]),), device="cuda")
for _ in range(8):
    opt.zero_grad(set_to_none=True)
    torch.nn.functional.cross_entropy(fwd()[idx], labels).backward()
    opt.step()
torch.cuda.synchronize()
print("done")
