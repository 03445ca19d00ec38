# coding=utf-8
"""Evidence for the parity band of the Reddit-shape GAT test (VERDICT r2 weak #1-ii): on the SAME sampled sub-problem
compare (a) the HIP layer and (b) the oracle in op-for-op fp32 (`acc=np.float32`: what a TF-CPU fp32 run computes) with
the float64-accumulated oracle, and report how far each is from the plain 1e-5 + 1e-5*|ref| band.  Runs on the GPU box:

    python tools/reddit_gat_band.py > gpurun_out/r03_reddit_gat_band.jsonl
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tf_geometric_amd as tfg            # noqa: E402
from tf_geometric_amd import synthetic    # noqa: E402
from oracle import tfg_oracle as oracle   # noqa: E402  (checker only)

n, e, f = synthetic.WORKLOADS["reddit"]
ei = tfg._lib.as_i32(synthetic.synthetic_edges(n, e, seed=3))
g = torch.Generator(device="cuda")
g.manual_seed(9)
x = torch.randn(n, f, generator=g, device="cuda")
g2 = torch.Generator(device="cpu")
g2.manual_seed(6)
rows = torch.sort(torch.randperm(n, generator=g2)[:400]).values.cuda()      # the test's own 400 rows
mask = torch.isin(ei[0].long(), rows)
ei_sub = ei[:, mask].cpu().numpy()
x_np = x.cpu().numpy()
rows_np = rows.cpu().numpy()
for A in (8, 64):
    rng = np.random.Generator(np.random.PCG64(40 + A))
    U, H = 64, 8
    wq, wk, wv = oracle.glorot_uniform(rng, f, A), oracle.glorot_uniform(rng, f, A), oracle.glorot_uniform(rng, f, U)
    bq, bk = (rng.standard_normal(A) * 0.1).astype(np.float32), (rng.standard_normal(A) * 0.1).astype(np.float32)
    b = (rng.standard_normal(U) * 0.1).astype(np.float32)
    layer = tfg.layers.GAT(U, attention_units=A, num_heads=H, activation=tfg.relu)
    layer._maybe_build([x])
    layer.set_weights(query_kernel=wq, query_bias=bq, key_kernel=wk, key_bias=bk, kernel=wv, bias=b)
    got = layer([x, ei])[rows].cpu().numpy().astype(np.float64)
    args = (x_np, ei_sub, wq, bq, "relu", wk, bk, "relu", wv, b, "relu")
    r64 = oracle.gat(*args, num_heads=H)[rows_np].astype(np.float64)
    r32 = oracle.gat(*args, num_heads=H, acc=np.float32)[rows_np].astype(np.float64)
    band = 1e-5 + 1e-5 * np.abs(r64)
    rec = {"attention_units": A, "rows": int(rows_np.size), "edges": int(ei_sub.shape[1])}
    for name, v in (("hip", got), ("oracle_fp32_op_for_op", r32)):
        err = np.abs(v - r64)
        rec[name] = {"max_abs_err": float(err.max()), "max_excess_over_1e-5_band": float((err - band).max()),
                     "frac_outside_band": float((err > band).mean())}
    print(json.dumps(rec), flush=True)
