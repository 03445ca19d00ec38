# coding=utf-8
"""Backward kernels as RATIOS to the forward pass measured alternately in one process (box-to-box and minute-to-minute
clock drift is +-5 %, which is as large as the effects being tuned): forward weighted sum, d/dw (SDDMM), max training
forward (tracked), max gradient (mask form), at products shape.

    python tools/ab_backward_ratios.py > gpurun_out/r03/backward_ratios.json
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tf_geometric_amd as tfg                                     # noqa: E402
from tf_geometric_amd import synthetic, autograd as AG, _lib as L   # noqa: E402
from tf_geometric_amd.plan import CsrPlan, segment_reduce           # noqa: E402
import bench                                                       # noqa: E402

n, e, f = synthetic.WORKLOADS["products"]
ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=0))
g = torch.Generator(device="cuda")
g.manual_seed(1)
x = torch.randn(n, f, generator=g, device="cuda")
w = torch.rand(int(ei.shape[1]), generator=g, device="cuda") + 0.5
plan = CsrPlan.build(ei, n, n)
w_csr = plan.edge_attr_to_csr(w)
gout = torch.randn(n, f, generator=g, device="cuda")
lib = L.require_gpu()
out = torch.empty((n, f), device="cuda")
dw = torch.empty(plan.num_edges, device="cuda")


def fwd():
    segment_reduce(plan, x, L.SUM, w_csr=w_csr, out=out)


def fwd_max():
    segment_reduce(plan, x, L.MAX, w_csr=w_csr, out=out)


def sddmm():
    L.check(lib.tfgx_sddmm_f32(L.ptr(plan.row_ptr), L.ptr(plan.col), n, L.ptr(gout), f, L.ptr(x), f, f, L.ptr(dw),
                               L.stream_ptr()), "sddmm")


xt = x.clone().requires_grad_(True)


def max_train_fwd():
    return AG.aggregate(plan, xt, L.MAX, w_csr=w_csr)


y = max_train_fwd()


def max_bwd():
    xt.grad = None
    y.backward(gout, retain_graph=True)


fns = {"forward_sum_w": fwd, "forward_max_w": fwd_max, "sddmm": sddmm, "max_training_forward": max_train_fwd, "max_gradient_mask": max_bwd}
times = {k: [] for k in fns}
for rnd in range(4):
    for k, fn in fns.items():
        times[k].append(bench._time(fn, steps=10, warmup=3 if rnd == 0 else 1))
med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
print(json.dumps({"shape": "products", "ms_median_of_4_alternating_rounds": med,
                  "ratio_to_forward_sum_w": {k: round(v / med["forward_sum_w"], 4) for k, v in med.items()},
                  "all_rounds_ms": times}))
