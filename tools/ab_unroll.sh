#!/bin/bash
# ON THE GPU BOX: A/B of the edge-walk unroll depth of seg_reduce_kernel for wide rows (CH = 2: 128 < F <= 512 columns
# per lane group pass; CH = 4: wider).  Rebuilds only tfgx_reduce.hip with -DTFGX_UNROLL_CH2 / -DTFGX_UNROLL_CH4.
set -u
OUT=gpurun_out/r02_ab_unroll.jsonl
: > $OUT
for cfg in "4 2" "8 2" "6 2" "8 4"; do
  set -- $cfg
  touch tf_geometric_amd/csrc/tfgx_reduce.hip
  TFGX_EXTRA_HIPCC_FLAGS="-DTFGX_UNROLL_CH2=$1 -DTFGX_UNROLL_CH4=$2" python -c "import sys; sys.path.insert(0,'.'); from tf_geometric_amd import _build; _build.build(verbose=False)" 2>&1 | tail -2
  python - "$1" "$2" >> $OUT <<'PY'
import sys, json, torch
sys.path.insert(0, '.')
from tf_geometric_amd import _lib as L, synthetic
from tf_geometric_amd.plan import CsrPlan, segment_reduce
n, e, _ = synthetic.WORKLOADS["products"]
ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=0))
E = int(ei.shape[1])
plan = CsrPlan.build(ei, n, n)
w = torch.rand(E, device="cuda") + 0.5
for f in (192, 256, 384, 512, 600):
    x = torch.randn(n, f, device="cuda"); out = torch.empty_like(x)
    for _ in range(3): segment_reduce(plan, x, L.SUM, w_csr=w, out=out)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(8): segment_reduce(plan, x, L.SUM, w_csr=w, out=out)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 8
    lines = -(-4 * f // 128)
    print(json.dumps({"unroll_ch2": int(sys.argv[1]), "unroll_ch4": int(sys.argv[2]), "F": f, "ms": ms,
                      "G_row_lines_per_s": E * lines / ms / 1e6,
                      "kernel": segment_reduce(plan, x, L.SUM, w_csr=w, out=out, describe=True)}))
    del x, out
PY
done
cat $OUT
