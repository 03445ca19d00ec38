# coding=utf-8
"""Where does the Reddit-shape GAT layer (demo-literal d_head = 1) lose accuracy against a float64 evaluation — in the
Q / K / V projections (602-term fp32 dot products on the MFMA GEMM) or in the fused attention kernel?  Evaluates the
attention of 400 sampled rows in float64 from (a) float64 projections and (b) the PRODUCT's own fp32 Q, K, V.

    python tools/reddit_gat_attribution.py > gpurun_out/r03/reddit_gat_attribution.json
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
from tf_geometric_amd.nn.conv import gat as G    # noqa: E402
from oracle import tfg_oracle as oracle   # noqa: E402  (checker only)

n, e, f = synthetic.WORKLOADS["reddit"]
ei = tfg._lib.as_i32(synthetic.synthetic_edges(n, e, seed=3))
g = torch.Generator(device="cuda")
g.manual_seed(9)
x = torch.randn(n, f, generator=g, device="cuda")
g2 = torch.Generator(device="cpu")
g2.manual_seed(6)
rows = torch.sort(torch.randperm(n, generator=g2)[:400]).values.cuda()
mask = torch.isin(ei[0].long(), rows)
ei_sub = ei[:, mask].cpu().numpy()
rows_np = rows.cpu().numpy()
A, U, H = 8, 64, 8
rng = np.random.Generator(np.random.PCG64(40 + A))
wq, wk, wv = oracle.glorot_uniform(rng, f, A), oracle.glorot_uniform(rng, f, A), oracle.glorot_uniform(rng, f, U)
bq, bk = (rng.standard_normal(A) * 0.1).astype(np.float32), (rng.standard_normal(A) * 0.1).astype(np.float32)
b = (rng.standard_normal(U) * 0.1).astype(np.float32)


def attention64(Q, K, V):
    """float64 evaluation of gat.py:73-120 on the sampled rows from given projections (numpy float64 [n, *])."""
    d, dv = A // H, U // H
    e_dst = np.concatenate([ei_sub[0].astype(np.int64), rows_np])
    e_src = np.concatenate([ei_sub[1].astype(np.int64), rows_np])
    pos = np.searchsorted(rows_np, e_dst)
    out = np.zeros((rows_np.size, U))
    for h in range(H):
        s = (Q[e_dst, h * d:(h + 1) * d] * K[e_src, h * d:(h + 1) * d]).sum(1) / np.sqrt(d)
        smax = np.full(rows_np.size, -np.inf)
        np.maximum.at(smax, pos, s)
        ex = np.exp(s - smax[pos])
        den = np.zeros(rows_np.size)
        np.add.at(den, pos, ex)
        np.add.at(out[:, h * dv:(h + 1) * dv], pos, (ex / (den[pos] + 1e-8))[:, None] * V[e_src, h * dv:(h + 1) * dv])
    return np.maximum(out + b, 0)


layer = tfg.layers.GAT(U, attention_units=A, num_heads=H, activation=tfg.relu)
layer._maybe_build([x])
layer.set_weights(query_kernel=wq, query_bias=bq, key_kernel=wk, key_bias=bk, kernel=wv, bias=b)
got = layer([x, ei])[rows].cpu().numpy().astype(np.float64)
L = tfg._lib
Qh, Kh, Vh = G._project_qkv(x, L.as_f32(wq), L.as_f32(bq), tfg.relu, L.as_f32(wk), L.as_f32(bk), tfg.relu, L.as_f32(wv))
xd = x.double()
Q64 = torch.relu(xd @ torch.tensor(wq, device="cuda").double() + torch.tensor(bq, device="cuda").double())
K64 = torch.relu(xd @ torch.tensor(wk, device="cuda").double() + torch.tensor(bk, device="cuda").double())
V64 = xd @ torch.tensor(wv, device="cuda").double()
Qt = torch.relu(x @ torch.tensor(wq, device="cuda") + torch.tensor(bq, device="cuda"))        # hipBLASLt fp32 for comparison
rec = {}
for name, a_, b_ in (("Q_hip_vs_f64", Qh, Q64), ("K_hip_vs_f64", Kh, K64), ("V_hip_vs_f64", Vh, V64), ("Q_hipblaslt_vs_f64", Qt, Q64)):
    d_ = (a_.double() - b_).abs()
    rec[name] = {"max_abs": float(d_.max()), "rms_abs": float(d_.pow(2).mean().sqrt()), "max_ref": float(b_.abs().max())}
ref_full = attention64(Q64.cpu().numpy(), K64.cpu().numpy(), V64.cpu().numpy())
ref_hipqkv = attention64(Qh.double().cpu().numpy(), Kh.double().cpu().numpy(), Vh.double().cpu().numpy())
band = 1e-5 + 1e-5 * np.abs(ref_full)
for name, a_, b_ in (("layer_vs_f64", got, ref_full), ("gemm_part(f64 attention of hip QKV vs f64)", ref_hipqkv, ref_full),
                     ("attention_part(layer vs f64 attention of hip QKV)", got, ref_hipqkv)):
    err = np.abs(a_ - b_)
    rec[name] = {"max_abs_err": float(err.max()), "max_excess_over_band": float((err - band).max()),
                 "n_outside": int((err > band).sum())}
print(json.dumps(rec, indent=1))
