#!/bin/bash
# tools/gemm_ab.py for the tree's libtfgx.so and every variant under tf_geometric_amd/lib/variants/ -> gpurun_out/gemm_ab.jsonl
OUT=gpurun_out/gemm_ab.jsonl
: > $OUT
timeout 300 python tools/gemm_ab.py tree 2>/dev/null | grep '^{' >> $OUT
for d in tf_geometric_amd/lib/variants/*/; do
  v=$(basename $d)
  TFGX_LIB_PATH=$d/libtfgx.so timeout 300 python tools/gemm_ab.py $v 2>/dev/null | grep '^{' >> $OUT
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/gemm_ab.jsonl")]
tags=[]
for r in rows:
    if r["tag"] not in tags: tags.append(r["tag"])
shapes=[]
for r in rows:
    k=(r["M"],r["K"],r["N"])
    if k not in shapes: shapes.append(k)
print("shape".ljust(24), "hipblaslt".rjust(9), *[t.rjust(8) for t in tags])
for s in shapes:
    rs={r["tag"]:r for r in rows if (r["M"],r["K"],r["N"])==s}
    lib=sorted(r["hipblaslt_ms"] for r in rs.values())[len(rs)//2]
    print(str(s).ljust(24), ("%.4f"%lib).rjust(9), *[("%.4f"%rs[t]["ms"]).rjust(8) if t in rs else "-".rjust(8) for t in tags])
PY
