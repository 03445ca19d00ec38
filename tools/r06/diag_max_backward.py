# coding=utf-8
"""Max aggregation at products shape, 512 columns: product d/dh vs float64 segment_reduce max on THE SAME float32 h
(identical decisions): isolates the tracked-max forward + mask backward from decision ambiguity."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch
from tf_geometric_amd import synthetic, _lib as L, plan as P, autograd as AG
from tf_geometric_amd.plan import CsrPlan
import f64_layers as R

n, e, f = synthetic.WORKLOADS["products"]
which = sys.argv[1] if len(sys.argv) > 1 else "uniform"
width = int(sys.argv[2]) if len(sys.argv) > 2 else 512
ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=0)) if which == "uniform" else synthetic.rmat_edges(n, e, 7, torch.device("cuda"))
g = torch.Generator(device="cuda"); g.manual_seed(1)
plan = CsrPlan.build(ei, n, n)
h = P.gather_friendly_empty(n, width, torch.device("cuda"))
h.copy_(torch.relu(torch.randn(n, width, generator=g, device="cuda")))
Gr = torch.randn(n, width, generator=g, device="cuda") * (plan.in_degree() > 0).float()[:, None]
ht = h.detach().requires_grad_()
red = AG.aggregate(plan, ht, L.MAX)
red.backward(Gr)
sg = R.SortedEdges(ei, n)
ref_dh = torch.zeros(n, width, dtype=torch.float64, device="cuda")
ref_red = torch.empty(n, width, dtype=torch.float64, device="cuda")
for c0 in range(0, width, 16):
    hc = h[:, c0:c0 + 16].double().requires_grad_()
    r = sg.seg_max_tf(hc[sg.col], R.F32_LOWEST)
    r.backward(Gr[:, c0:c0 + 16].double())
    ref_dh[:, c0:c0 + 16] = hc.grad
    ref_red[:, c0:c0 + 16] = r.detach()
print(which, "forward equal:", bool((red.detach().double() == ref_red).all()))
d = (ht.grad.double() - ref_dh).abs()
bad = d > 1e-5 * (1 + ref_dh.abs())
print("bad elements", int(bad.sum()), "in rows", int(bad.any(1).sum()), "max |d|", float(d.max()), "max |ref|", float(ref_dh.abs().max()))
br = torch.nonzero(bad.any(1)).flatten()
dout = torch.bincount(ei[1].long(), minlength=n)
print("first bad rows", br[:10].tolist(), "their out-degrees", dout[br[:10]].tolist(), "max out-degree", int(dout.max()))
if br.numel():
    i = int(br[0]); j = int(torch.nonzero(bad[i]).flatten()[0])
    print("row", i, "col", j, "got", float(ht.grad[i, j]), "ref", float(ref_dh[i, j]), "h", float(h[i, j]))
if br.numel():
    # by hand: destinations of source i, which of them it wins in column j, tie counts there
    dst = ei[0][ei[1] == i].long()
    print("source", i, "out-edges", dst.numel(), "distinct destinations", torch.unique(dst).numel())
    total = 0.0
    for r_ in dst.tolist():
        nb = ei[1][ei[0] == r_].long()
        vals = h[nb, j]
        mx = float(vals.max())
        if float(h[i, j]) == mx:
            cnt = int((vals == mx).sum())
            total += float(Gr[r_, j]) / cnt
            print("   wins at destination", r_, "deg", nb.numel(), "ties", cnt, "g", float(Gr[r_, j]), "red", float(red[r_, j]), "ref_red", float(ref_red[r_, j]))
    print("by hand", total)
