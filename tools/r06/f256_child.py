# coding=utf-8
"""One process: the headline launch at F = 256 on the products graph (layer layout, rows 288 floats apart), a few timed repeats.
Prints one JSON line {ms, x_ptr, out_ptr}.  The width is bimodal FROM PROCESS TO PROCESS (profiles/r06_ab_f256.jsonl): run under
`rocprofv3 --kernel-trace --pmc ...` by tools/r06/f256_modes.sh to see which counter moves with the mode."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from tf_geometric_amd import synthetic, _lib as L, plan as P
from tf_geometric_amd.plan import CsrPlan, segment_reduce
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n, e, _ = synthetic.WORKLOADS["products"]
ei = L.as_i32(synthetic.synthetic_edge_stripe(n, e, seed=0))
plan = CsrPlan.build(ei, n, n)
w = torch.rand(int(ei.shape[1]), device="cuda") + 0.5
sc = torch.rand(n, device="cuda")
x = P.gather_friendly_empty(n, F, torch.device("cuda")); x.normal_()
out = torch.empty(n, F, device="cuda")
fn = lambda: segment_reduce(plan, x, L.SUM, w_csr=w, self_coef=sc, out=out)
for _ in range(2):
    fn()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(4):
    torch.cuda.synchronize(); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
print(json.dumps({"F": F, "ms_min": round(min(ts), 3), "ms_all": [round(t, 3) for t in ts], "x_ptr": hex(x.data_ptr()), "out_ptr": hex(out.data_ptr()),
                  "x_ptr_mod_2MiB": x.data_ptr() % (2 << 20), "kernel": segment_reduce(plan, x, L.SUM, w_csr=w, self_coef=sc, out=out, describe=True)}))
