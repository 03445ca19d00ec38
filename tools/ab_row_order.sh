#!/bin/bash
# Same-box A/B of the degree-ordered row walk at wide rows (F = 100 / 128 / 256) on R-MAT graphs of three sizes:
# natural order (variant rmat) vs walk order (variant rmat_row_order), alternating.  -> gpurun_out/ab_row_order.jsonl
OUT=gpurun_out/ab_row_order.jsonl
: > $OUT
for shape in "2400000 123000000" "2097152 61000000"; do
  set -- $shape
  for f in 100 128 256; do
    for v in rmat rmat_row_order rmat rmat_row_order; do
      RMAT_N=$1 RMAT_E=$2 RMAT_F=$f timeout 200 python tools/rmat_pmc.py $v 10 2>/dev/null | grep '^{' >> $OUT
    done
  done
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/ab_row_order.jsonl")]
agg={}
for r in rows: agg.setdefault((r["N"],r["edges"],r["F"],r["variant"]),[]).append(r["ms_per_call_events"])
for k,v in sorted(agg.items()): print(k, [round(x,3) for x in v])
PY
