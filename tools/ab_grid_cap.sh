#!/bin/bash
# ON THE GPU BOX: A/B of the grid cap of seg_reduce_kernel (rows per lane group -> depth of the row-header pipeline).
OUT=gpurun_out/r02_ab_grid_cap.jsonl
: > $OUT
for cap in 1048576 65536 16384 8192 4096 2048; do
TFGX_REDUCE_GRID_CAP=$cap python - "$cap" >> $OUT <<'PY'
import sys, json, torch
sys.path.insert(0, '.')
from tf_geometric_amd import _lib as L, synthetic
from tf_geometric_amd.plan import CsrPlan, segment_reduce
n, e, _ = synthetic.WORKLOADS["products"]
ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=0))
E = int(ei.shape[1])
plan = CsrPlan.build(ei, n, n)
w = torch.rand(E, device="cuda") + 0.5
for f in (16, 32, 64, 100, 128, 256, 512):
    x = torch.randn(n, f, device="cuda"); out = torch.empty_like(x)
    for _ in range(3): segment_reduce(plan, x, L.SUM, w_csr=w, out=out)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(8): segment_reduce(plan, x, L.SUM, w_csr=w, out=out)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 8
    print(json.dumps({"grid_cap": int(sys.argv[1]), "F": f, "ms": round(ms, 4),
                      "frac_alg": (E * (4 * f + 8) + n * 4 * f + 4 * (n + 1)) / ms / 1e6 / 8000}))
    del x, out
PY
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open('gpurun_out/r02_ab_grid_cap.jsonl')]
caps=sorted({r["grid_cap"] for r in rows}, reverse=True)
fs=sorted({r["F"] for r in rows})
print("F      " + "  ".join("{:>9d}".format(c) for c in caps))
for f in fs:
    print("{:<6d} ".format(f) + "  ".join("{:9.3f}".format(next(r["ms"] for r in rows if r["F"]==f and r["grid_cap"]==c)) for c in caps))
PY
