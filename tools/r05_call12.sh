#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 12: R-MAT after the burst policy for re-laid-out tables; fused launch vs two launches at F = 128.
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/r05_call12"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
: > "$OUT/r05_rmat_vs_uniform_final.jsonl"
for rep in 1 2; do
  for g in uniform rmat; do
    timeout 300 python tools/ab_wide_blocks.py $g 100,128,192,256,512 >> "$OUT/r05_rmat_vs_uniform_final.jsonl" 2>> "$OUT/err.log"
  done
done
timeout 300 python tools/ab_fused_layer.py products uniform 128 > "$OUT/r05_ab_fused_layer_F128.json" 2>> "$OUT/err.log"
timeout 300 python tools/ab_fused_layer.py products uniform > "$OUT/r05_ab_fused_layer.json" 2>> "$OUT/err.log"
timeout 300 python -m pytest tests/test_gpu_aggregate.py tests/test_gpu_fullsize.py -x -q > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log" >&2
grep -v amdgpu.ids "$OUT/err.log" | tail -3 >&2
cat "$OUT/r05_ab_fused_layer_F128.json" "$OUT/r05_ab_fused_layer.json"
