# coding=utf-8
"""Which stage of the max gradient breaks at products shape with tied maxima: packed tracking vs the arg kernel, mask form vs pull form."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from tf_geometric_amd import synthetic, _lib as L, plan as P, autograd as AG
from tf_geometric_amd.plan import CsrPlan, segment_reduce, can_track

n, e, f = synthetic.WORKLOADS["products"]
ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=0))
g = torch.Generator(device="cuda"); g.manual_seed(1)
plan = CsrPlan.build(ei, n, n)
lib = L.require_gpu()
for width, pad in ((64, False), (512, False), (512, True), (100, False)):
    h = P.gather_friendly_empty(n, width, torch.device("cuda")) if pad else torch.empty(n, width, device="cuda")
    h.copy_(torch.relu(torch.randn(n, width, generator=g, device="cuda")))
    Gr = torch.randn(n, width, generator=g, device="cuda")
    x2, ldx = L.row_major_2d(h)
    print("width", width, "ld", ldx, "can_track", can_track(plan, x2, ldx))
    out0, cnt0 = torch.empty((n, width), device="cuda"), torch.empty((n, width), device="cuda")
    arg0 = torch.empty((n, width), dtype=torch.int32, device="cuda")
    L.check(lib.tfgx_segment_max_with_arg_f32(L.ptr(plan.row_ptr), L.ptr(plan.col), None, n, L.ptr(x2), ldx, width,
                                              L.ptr(out0), width, L.ptr(cnt0), width, L.ptr(arg0), width, L.stream_ptr()), "with_arg")
    if can_track(plan, x2, ldx):
        out1 = torch.empty((n, width), device="cuda")
        packed = torch.empty((n, width), dtype=torch.int32, device="cuda")
        segment_reduce(plan, x2, L.MAX, out=out1, track=packed)
        pk = packed.to(torch.int64) & 0xFFFFFFFF
        cnt1 = (pk >> 16).float()
        pos1 = torch.where(cnt1 > 0, plan.row_ptr[:-1].long().unsqueeze(1) + (pk & 0xFFFF), torch.full_like(pk, -1))
        print("   packed: out equal", bool(torch.equal(out0, out1)), "count mismatches", int((cnt0 != cnt1).sum()),
              "pos mismatches", int((arg0.long() != pos1).sum()), "max count", float(cnt0.max()), "elements with ties", int((cnt0 > 1).sum()))
    # pull form (reference kernel, needs count) vs autograd path
    ht = h.detach().requires_grad_()
    AG.aggregate(plan, ht, L.MAX).backward(Gr)
    pt, t2d = AG._transposed(plan)
    gx = torch.empty((n, width), device="cuda")
    gn = torch.empty((n, width), device="cuda")
    L.check(lib.tfgx_segment_max_backward_hub_f32(L.ptr(pt.row_ptr), L.ptr(pt.col), None, pt.n_dst, L.ptr(x2), ldx, width,
                                                  L.ptr(out0), width, L.ptr(Gr), width, L.ptr(cnt0), width, L.ptr(gx), width,
                                                  plan.n_dst, L.ptr(gn), None, None, L.stream_ptr()), "pull")
    d = (ht.grad - gx).abs()
    print("   autograd (mask form) vs pull form: max |d|", float(d.max()), "bad elements", int((d > 1e-4).sum()))
    del h, Gr, out0, cnt0, arg0, ht, gx, gn
