#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 39: max_backward_mask_apply_kernel with 32 x 32 -> 64-bit offsets (main) vs the committed kernel (bwd0).
set -u
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r05_maxbwd"; mkdir -p "$OUT"; cd "$ROOT"
timeout 500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_fullsize.py -m gpu -x -q -k "max or pool" 2>&1 | tail -1
for v in main bwd0 main bwd0; do
  if [ "$v" = "main" ]; then unset TFGX_LIB_PATH; else export TFGX_LIB_PATH="$ROOT/tf_geometric_amd/lib/variants/$v/libtfgx.so"; fi
  timeout 300 python bench.py --no-cpu-baseline --no-rmat --steps 10 --warmup 3 2>/dev/null > "$OUT/b.json"
  python - "$v" <<'PY' | tee -a gpurun_out/r05_maxbwd/r05_maxpool_bwd_offsets_ab.jsonl
import json,sys
d=json.load(open("gpurun_out/r05_maxbwd/b.json"))
c=d["configs"]["C4_maxpool_sage_256_concat"]
print(json.dumps({"lib": sys.argv[1], "maxpool_forward_ms": round(c["forward_ms"],3), "maxpool_fwd_bwd_ms": round(c["fwd_bwd_ms"],3),
                  "mean_sage_fwd_bwd_ms": round(d["configs"]["C4_mean_sage_256_concat"]["fwd_bwd_ms"],3)}))
PY
done
