# coding=utf-8
"""Same-box A/B of the product segment-reduce kernel over feature widths, one setting per process (the switches are read
once): TFGX_REDUCE_WIDE_BLOCKS=0|1 (column blocks on grid.y for wide line-aligned rows, round 5) and TFGX_LIB_PATH=<variant
library> (e.g. lib/variants/serial_tail: the round 1-4 one-load-per-edge remainder).  One JSON line per (graph, F, op).

    python tools/ab_wide_blocks.py uniform|rmat [F,F,...]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tf_geometric_amd import synthetic, _lib as L, plan as P    # noqa: E402

graph = sys.argv[1] if len(sys.argv) > 1 else "uniform"
widths = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [64, 96, 100, 128, 160, 192, 224, 256, 320, 384, 512, 1024]
n, e, _ = synthetic.WORKLOADS["products"]
dev = torch.device("cuda")
ei = L.as_i32(synthetic.synthetic_edge_stripe(n, e, seed=0)) if graph == "uniform" else synthetic.rmat_edges(n, e, 7, dev)
E = int(ei.shape[1])
plan = P.CsrPlan.build(ei, n, n)
plan.hub_info()
torch.manual_seed(11)                      # the same operands in every process: the checksums of two settings must agree
w = torch.rand(E, device=dev) + 0.5
sc = torch.rand(n, device=dev)
tag = {"wide_blocks": os.environ.get("TFGX_REDUCE_WIDE_BLOCKS", "2"), "g256": os.environ.get("TFGX_REDUCE_WIDE_G256", "32"), "lib": os.path.basename(os.path.dirname(os.environ.get("TFGX_LIB_PATH", "lib/x")))}


def timeit(fn, steps=6, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


pad = int(os.environ.get("AB_LD_PAD", "0"))          # AB_LD_PAD=32: the table's rows start (F + 32) floats apart (a view of a wider array)
tag["ld_pad"] = pad
for f in widths:
    x = torch.randn(n, f + pad, device=dev)[:, :f] if pad else torch.randn(n, f, device=dev)
    out = torch.empty((n, f), dtype=torch.float32, device=dev)
    for name, op, ww, s in (("sum_w_self", L.SUM, w, sc), ("max", L.MAX, None, None)):
        if name == "max" and f not in (100, 256, 512):
            continue
        ms = timeit(lambda: P.segment_reduce(plan, x, op, w_csr=ww, self_coef=s, out=out))
        e_agg = E + (n if s is not None else 0)
        balg = e_agg * (4 * f + 4 + (4 if ww is not None else 0)) + n * 4 * f + 4 * (n + 1)
        print(json.dumps(dict(tag, graph=graph, F=f, op=name, ms=round(ms, 4), frac_of_8TBps_alg=round(balg / ms / 1e6 / 8000, 4),
                              kernel=P.segment_reduce(plan, x, op, w_csr=ww, self_coef=s, out=out, describe=True),
                              checksum=float(out.double().abs().sum().item()))), flush=True)
    del x, out
