#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 38: seg_reduce_kernel with 32 x 32 -> 64-bit row offsets (variant red32) vs the tree's kernel.
set -u
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r05_red32"; mkdir -p "$OUT"; cd "$ROOT"
for v in main red32 main red32 main red32; do
  if [ "$v" = "main" ]; then unset TFGX_LIB_PATH; else export TFGX_LIB_PATH="$ROOT/tf_geometric_amd/lib/variants/$v/libtfgx.so"; fi
  timeout 300 python bench.py --no-cpu-baseline --no-configs --steps 20 --warmup 5 2>/dev/null > "$OUT/b.json"
  python - "$v" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r05_red32/b.json"))
r=d["roofline"]
st=d.get("static_layout",{})
print(sys.argv[1], "headline_ms", round(r["kernel_ms"],4), "frac", round(r["frac"],4), "rmat_ms", round(d["rmat"]["kernel_ms"],4), "static", {k:(round(v,4) if isinstance(v,float) else v) for k,v in st.items() if "ms" in k})
PY
done
