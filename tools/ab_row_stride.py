import sys, json, torch
sys.path.insert(0, ".")
import tf_geometric_amd as tfg  # noqa: F401 (loads the library)
from tf_geometric_amd import synthetic, _lib as L
from tf_geometric_amd.plan import CsrPlan, segment_reduce
n, e, f = synthetic.WORKLOADS["products"]
ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=0))
plan = CsrPlan.build(ei, n, n)
w = torch.rand(plan.num_edges, device="cuda")
def t(fn, k=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k
for F, lds in [(20, [20, 32]), (24, [24, 32]), (7, [7, 8]), (12, [12, 16]), (47, [47, 48, 64]), (41, [41, 48]), (40, [40, 48, 64]), (100, [100, 112, 128]), (172, [172, 176, 192])]:
    row = {"F": F}
    for ld in lds:
        buf = torch.randn(n, ld, device="cuda")
        x = buf[:, :F]
        out = torch.empty(n, F, device="cuda")
        row["ld%d" % ld] = round(t(lambda: segment_reduce(plan, x, L.SUM, w_csr=w, out=out)), 3)
    print(json.dumps(row))
