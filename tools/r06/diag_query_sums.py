# coding=utf-8
"""dQ of the d == 1 attention from the forward's sums vs from the destination pass, both against float64 (small dense graph with
peaked softmaxes, and Reddit-density rows).  usage: python tools/r06/diag_query_sums.py"""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from tf_geometric_amd.nn.conv import gat as G
from tf_geometric_amd import autograd as AG
from tf_geometric_amd.plan import CsrPlan

def f64(ei, n, H, dv, Qn, Kn, Vn, gout):
    dev = "cuda"
    Q, K, V = (torch.tensor(t, dtype=torch.float64, device=dev, requires_grad=True) for t in (Qn, Kn, Vn))
    row = torch.cat([torch.tensor(ei[0], device=dev).long(), torch.arange(n, device=dev)])
    col = torch.cat([torch.tensor(ei[1], device=dev).long(), torch.arange(n, device=dev)])
    s = Q[row] * K[col]
    mx = torch.full((n, H), -float("inf"), dtype=torch.float64, device=dev).scatter_reduce(0, row[:, None].expand(-1, H), s.detach(), "amax")
    ex = torch.exp(s - mx[row])
    den = torch.zeros((n, H), dtype=torch.float64, device=dev).index_add(0, row, ex) + 1e-8
    alpha = ex / den[row]
    out = torch.zeros((n, H, dv), dtype=torch.float64, device=dev).index_add(0, row, alpha[:, :, None] * V[col].reshape(-1, H, dv))
    out.reshape(n, H * dv).backward(gout.double())
    return Q.grad

for n, e, std in ((500, 30000, 1.5), (500, 30000, 0.3), (20000, 4000000, 1.0), (20000, 10000000, 0.2)):
    rng = np.random.default_rng(77)
    H, dv = 8, 8
    ei = np.stack([rng.integers(0, n, e), rng.integers(0, n, e)]).astype(np.int32)
    plan = CsrPlan.from_cache(ei, n, n, {})
    Qn, Kn, Vn = (rng.standard_normal((n, H)).astype(np.float32) * std, rng.standard_normal((n, H)).astype(np.float32) * std + 0.7,
                  rng.standard_normal((n, H * dv)).astype(np.float32))
    gout = torch.tensor(rng.standard_normal((n, H * dv)).astype(np.float32), device="cuda")
    ref = f64(ei, n, H, dv, Qn, Kn, Vn, gout)
    res = {"n": n, "e": e, "std": std, "ref_absmax": float(ref.abs().max()), "ref_rms": float(ref.pow(2).mean().sqrt())}
    for on in (True, False):
        Q, K, V = (torch.tensor(t, device="cuda", requires_grad=True) for t in (Qn, Kn, Vn))
        G.QUERY_GRAD_SUMS = on
        out = AG.gat_attention(plan, Q, K, V, H)
        out.backward(gout)
        err = (Q.grad.double() - ref).abs()
        res["sums" if on else "dst_pass"] = {"max_abs_err": float(err.max()), "max_scaled": float((err / (1 + ref.abs())).max())}
    G.QUERY_GRAD_SUMS = True
    print(json.dumps(res))
