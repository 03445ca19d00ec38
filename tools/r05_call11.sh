#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 11: whole GPU suite on the final tree; R-MAT against uniform through the product path.
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/r05_call11"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > "$OUT/pytest_gpu.log" 2>&1
tail -12 "$OUT/pytest_gpu.log" >&2
: > "$OUT/r05_rmat_vs_uniform_final.jsonl"
for rep in 1 2; do
  for g in uniform rmat; do
    timeout 300 python tools/ab_wide_blocks.py $g 100,128,192,256,512 >> "$OUT/r05_rmat_vs_uniform_final.jsonl" 2>> "$OUT/err.log"
  done
done
python "$ROOT/__graft_entry__.py" --smoke > "$OUT/smoke.log" 2>&1
tail -3 "$OUT/smoke.log" >&2
cat "$OUT/r05_rmat_vs_uniform_final.jsonl"
