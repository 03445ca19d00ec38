# coding=utf-8
"""Where does d/dx of the products-shape GCN(256) layer leave the float64 reference?  Stage by stage."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
import torch
import tf_geometric_amd as tfg
from tf_geometric_amd import synthetic, _lib as L, plan as P, autograd as AG
from tf_geometric_amd.plan import CsrPlan, segment_reduce, gemm_bias_act, transpose
import f64_layers as R

n, e, f = synthetic.WORKLOADS["products"]
ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=0))
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn(n, f, generator=g, device="cuda")
w = torch.rand(int(ei.shape[1]), generator=g, device="cuda") + 0.5
plan = CsrPlan.build(ei, n, n)

def stat(name, got, ref, tol=1e-5):
    d = (got.double() - ref.double()).abs() - tol * ref.double().abs()
    bad_rows = (d > tol).any(1)
    print("{:<40s} worst/tol {:10.3f}  bad rows {} of {}  first bad {}".format(
        name, float(d.max()) / tol, int(bad_rows.sum()), d.shape[0], torch.nonzero(bad_rows)[:8].flatten().tolist()))
    return bad_rows

# stage A: the GEMM g @ W^T at M = 2.4M, K = 256, N = 100
G = torch.randn(n, 256, generator=g, device="cuda")
W = torch.randn(100, 256, generator=g, device="cuda") * 0.1
ga = gemm_bias_act(G, transpose(W))
ref = (G.double() @ W.double().t())
stat("gemm 2.4M x 256 -> 100", ga, ref)
ga2 = torch.matmul(G, W.t())
stat("torch.matmul same", ga2, ref)
# stage B: transposed aggregation of a [n,100] table with weights + self coef
sc = torch.rand(n, device="cuda")
w_csr = plan.edge_attr_to_csr(w)
pt, t2d = AG._transposed(plan)
w_t = AG._transposed_weights(plan, w_csr, t2d)
gg = P.gather_friendly_copy(ga)
gx = segment_reduce(pt, gg, L.SUM, w_csr=w_t, self_coef=sc)
row, col = ei[0].long(), ei[1].long()
refx = R.segment_sum_columns(ga.double(), col, row, w.double(), n) + sc.double()[:, None] * ga.double()
stat("transposed aggregate", gx, refx)
# forward aggregate for comparison
fx = segment_reduce(plan, x, L.SUM, w_csr=w_csr, self_coef=sc)
reff = R.segment_sum_columns(x.double(), row, col, w.double(), n) + sc.double()[:, None] * x.double()
stat("forward aggregate", fx, reff)
