# coding=utf-8
"""products-shape GCN(256): product d/dx vs float64 autograd vs the explicit float64 formula; which rows differ and by what."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
import torch
import tf_geometric_amd as tfg
from tf_geometric_amd import synthetic, _lib as L, plan as P, autograd as AG
from tf_geometric_amd.plan import CsrPlan
from oracle import tfg_oracle as oracle
import f64_layers as R

n, e, f = synthetic.WORKLOADS["products"]
ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=0))
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn(n, f, generator=g, device="cuda")
w = torch.rand(int(ei.shape[1]), generator=g, device="cuda") + 0.5
plan = CsrPlan.build(ei, n, n)
rng = np.random.Generator(np.random.PCG64(163))
ws = dict(kernel=oracle.glorot_uniform(rng, f, 256), bias=(rng.standard_normal(256) * 0.1).astype(np.float32))
layer = tfg.layers.GCN(256, activation=tfg.relu)
layer._maybe_build([x]); layer.set_weights(**ws); layer.trainable(True)
Gup = torch.randn(n, 256, generator=g, device="cuda")
cache = {"tfgx_csr_plan": plan}
xt = x.clone().requires_grad_()
out = layer([xt, ei, w], cache=cache)
out.backward(Gup)
ref_out, ref = R.gcn_layer(x, ei, w, ws["kernel"], ws["bias"], Gup)
d = (xt.grad.double() - ref["x"]).abs()
bad = (d > 1e-4).any(1)
print("bad rows", int(bad.sum()), "of", n, "first", torch.nonzero(bad)[:10].flatten().tolist(), "max", float(d.max()))
deg_out = torch.bincount(ei[1].long(), minlength=n)
deg_in = torch.bincount(ei[0].long(), minlength=n)
bi = torch.nonzero(bad).flatten()
print("out-degree of bad rows (first 10)", deg_out[bi[:10]].tolist(), "in-degree", deg_in[bi[:10]].tolist())
print("bad columns per bad row (first 10)", (d[bi[:10]] > 1e-4).sum(1).tolist())
# relu mask agreement
m_prod = out.detach() > 0
m_ref = ref_out > 0
dm = (m_prod != m_ref)
print("relu mask differs at", int(dm.sum()), "elements in", int(dm.any(1).sum()), "rows; |ref pre-activation| there max",
      float(ref_out[dm].abs().max()) if dm.any() else 0.0, float(out.detach()[dm].abs().max()) if dm.any() else 0.0)
