# coding=utf-8
"""The worst forward element of the fused attention at d_head = 1 with large scores: its row's (m, l) vs float64, by hand."""
import sys, os, math
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import torch
from tf_geometric_amd import synthetic, _lib as L
from tf_geometric_amd.plan import CsrPlan
from tf_geometric_amd.nn.conv import gat as G_
import f64_layers as R

n, e, f = synthetic.WORKLOADS["reddit"]
A, H, U = 8, 8, 64
ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=3))
plan = CsrPlan.build(ei, n, n)
g = torch.Generator(device="cuda"); g.manual_seed(5)
Q = torch.relu(torch.randn(n, A, generator=g, device="cuda")) * 1.5
K = torch.relu(torch.randn(n, A, generator=g, device="cuda")) * 1.5
V = torch.randn(n, U, generator=g, device="cuda")
sg = R.SortedEdges(ei, n, self_loops=True)
def reference(dtype):
    q, k, v = (t.to(dtype) for t in (Q, K, V))
    outs, ms, ls = [], [], []
    for h in range(H):
        s = (q[:, h:h + 1][sg.row] * k[:, h:h + 1][sg.col]).sum(-1)
        m = sg.seg_max(s, -math.inf)
        p = torch.exp(s - m[sg.row])
        l = sg.seg_sum(p)
        a = p / (l + 1e-8)[sg.row]
        outs.append(sg.seg_sum(a[:, None] * v[:, h * 8:(h + 1) * 8][sg.col])); ms.append(m); ls.append(l)
    return torch.cat(outs, 1), torch.stack(ms, 1), torch.stack(ls, 1)
o64, m64, l64 = reference(torch.float64)
o32, m32, l32 = reference(torch.float32)
for kb in (None, 1):
    G_.SOURCE_BLOCKS = kb
    stats = torch.empty(n, 2 * H, device="cuda")
    out = G_.gat_attention(plan, Q, K, V, H, True, stats_ml=stats)
    d = (out.double() - o64).abs()
    i = int(d.argmax()); r, c = i // U, i % U; h = c // 8
    print("blocks", kb, "worst |d|", float(d.max()), "row", r, "col", c, "deg", int(plan.in_degree()[r]),
          "ours", float(out[r, c]), "f64", float(o64[r, c]), "f32", float(o32[r, c]))
    print("   m ours", float(stats[r, 2 * h]), "f64", float(m64[r, h]), " l ours", float(stats[r, 2 * h + 1]), "f64", float(l64[r, h]), "f32", float(l32[r, h]))
    print("   rel err of l: ours {:.3e} f32 {:.3e}".format(abs(float(stats[r, 2 * h + 1]) - float(l64[r, h])) / float(l64[r, h]),
                                                           abs(float(l32[r, h]) - float(l64[r, h])) / float(l64[r, h])))
    relL = ((stats[:, 1::2].double() - l64).abs() / l64)
    relL32 = ((l32.double() - l64).abs() / l64)
    print("   over all rows: worst rel err of l ours {:.3e}, f32 {:.3e}; mean ours {:.3e}, f32 {:.3e}".format(
        float(relL.max()), float(relL32.max()), float(relL.mean()), float(relL32.mean())))
# emulate the kernel's online softmax for row r, head h in numpy float32 (one pass: CSR order, then the self-loop)
import numpy as np
rp = plan.row_ptr.cpu().numpy(); col = plan.col.cpu().numpy()
srcs = np.concatenate([col[rp[r]:rp[r + 1]], [r]])
qv = np.float32(Q[r, h].item()); kv = K[:, h].cpu().numpy()[srcs].astype(np.float32)
sc = (qv * kv).astype(np.float32)
m, l = np.float32(-3.4028234663852886e38), np.float32(0)
for s_ in sc:
    dlt = np.float32(s_ - m)
    if dlt > 0:
        ex = np.float32(np.exp(np.float64(-dlt))); corr, p, m = ex, np.float32(1), s_
    else:
        ex = np.float32(np.exp(np.float64(dlt))); corr, p = np.float32(1), ex
    l = np.float32(np.float64(l) * np.float64(corr) + np.float64(p))
l_true = np.sum(np.exp(sc.astype(np.float64) - np.float64(sc.max())))
print("emulated float32 online softmax: m", m, "l", l, " exact-on-float32-scores l", l_true, " kernel l", float(stats[r, 2 * h + 1]), " f64 l", float(l64[r, h]))
print("scores: max", sc.max(), "second", np.sort(sc)[-2], "count", sc.size)
