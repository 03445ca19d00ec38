#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 37: fused aggregate -> project kernel with 32 x 32 -> 64-bit row offsets (main) vs int64 x int64 (fused0).
set -u
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r05_fused_off"; mkdir -p "$OUT"; cd "$ROOT"
timeout 500 python -m pytest tests/test_gpu_layers.py tests/test_gpu_reference_golden.py -m gpu -x -q -k "fused or gcn or sage or golden" 2>&1 | tail -1
for v in main fused0 main fused0; do
  if [ "$v" = "main" ]; then unset TFGX_LIB_PATH; else export TFGX_LIB_PATH="$ROOT/tf_geometric_amd/lib/variants/$v/libtfgx.so"; fi
  timeout 300 python tools/ab_fused_layer.py products uniform 2>/dev/null | sed "s/^{/{\"lib\": \"$v\", /" >> "$OUT/r05_fused_offsets_ab.jsonl"
  timeout 300 python tools/ab_fused_layer.py arxiv uniform 2>/dev/null | sed "s/^{/{\"lib\": \"$v\", /" >> "$OUT/r05_fused_offsets_ab.jsonl"
done
python - <<'PY'
import json
for ln in open("gpurun_out/r05_fused_off/r05_fused_offsets_ab.jsonl"):
    d=json.loads(ln)
    keep={k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if isinstance(v,(int,float,str)) and ("ms" in k or k in("lib","workload","which"))}
    print(json.dumps(keep)[:400])
PY
