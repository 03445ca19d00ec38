# coding=utf-8
"""Reddit shape, H = 8, d_head = 1 (A = 8), dv = 8: dQ / dK / dV of the fused attention vs float64 autograd and vs float32
op-for-op autograd of the reference formulation; where is the product less accurate than plain float32?"""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch
from tf_geometric_amd import synthetic, _lib as L, autograd as AG
from tf_geometric_amd.plan import CsrPlan
from tf_geometric_amd.nn.conv import gat as G_
import f64_layers as R

n, e, f = synthetic.WORKLOADS["reddit"]
A = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, U = 8, 64
ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=3))
plan = CsrPlan.build(ei, n, n)
g = torch.Generator(device="cuda"); g.manual_seed(5)
Q = torch.relu(torch.randn(n, A, generator=g, device="cuda")) * 1.5
K = torch.relu(torch.randn(n, A, generator=g, device="cuda")) * 1.5
V = torch.randn(n, U, generator=g, device="cuda")
Gup = torch.randn(n, U, generator=g, device="cuda")
sg = R.SortedEdges(ei, n, self_loops=True)

def reference(dtype):
    q, k, v = (t.detach().clone().to(dtype).requires_grad_() for t in (Q, K, V))
    d, dv = A // H, U // H
    outs = []
    for h in range(H):
        s = (q[:, h * d:(h + 1) * d][sg.row] * k[:, h * d:(h + 1) * d][sg.col]).sum(-1) / math.sqrt(d)
        m = sg.seg_max(s.detach(), -math.inf)
        p = torch.exp(s - m[sg.row])
        a = p / (sg.seg_sum(p) + 1e-8)[sg.row]
        o = sg.seg_sum(a[:, None] * v[:, h * dv:(h + 1) * dv][sg.col])
        o.backward(Gup[:, h * dv:(h + 1) * dv].to(dtype))
        outs.append(o.detach())
    return torch.cat(outs, 1), q.grad, k.grad, v.grad

o64, q64, k64, v64 = reference(torch.float64)
o32, q32, k32, v32 = reference(torch.float32)
qt, kt, vt = (t.clone().requires_grad_() for t in (Q, K, V))
out = AG.gat_attention(plan, qt, kt, vt, H)
out.backward(Gup)
def err(a, r):
    return float(((a.double() - r).abs() / (1 + r.abs().amax(1, keepdim=True) + r.abs())).max()), float((a.double() - r).abs().max()), float(r.abs().max())
for name, a, b, r in (("out", out.detach(), o32, o64), ("dQ", qt.grad, q32, q64), ("dK", kt.grad, k32, k64), ("dV", vt.grad, v32, v64)):
    print("{:<4s} ours: row-scaled {:.3e} abs {:.3e} | float32 autograd: row-scaled {:.3e} abs {:.3e} | max |ref| {:.3e}".format(
        name, *err(a, r)[:2], *err(b, r)))
print("block stats", G_.SOURCE_BLOCK_STATS)
for kb in (1,):
    G_.SOURCE_BLOCKS = kb
    qt, kt, vt = (t.clone().requires_grad_() for t in (Q, K, V))
    out = AG.gat_attention(plan, qt, kt, vt, H)
    out.backward(Gup)
    for name, a, r in (("out", out.detach(), o64), ("dQ", qt.grad, q64), ("dK", kt.grad, k64), ("dV", vt.grad, v64)):
        print("  one pass (no blocks) {:<4s} row-scaled {:.3e} abs {:.3e}".format(name, *err(a, r)[:2]))
