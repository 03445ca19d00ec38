#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 3: product-level A/B of the wide-row column blocks and of the masked remainder batch
# (uniform + R-MAT), then the whole GPU test suite with durations.
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/r05_call3"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
: > "$OUT/r05_ab_wide_blocks.jsonl"
for rep in 1 2; do
  TFGX_REDUCE_WIDE_BLOCKS=0 timeout 300 python tools/ab_wide_blocks.py uniform >> "$OUT/r05_ab_wide_blocks.jsonl" 2>> "$OUT/ab.err"
  TFGX_REDUCE_WIDE_BLOCKS=1 timeout 300 python tools/ab_wide_blocks.py uniform >> "$OUT/r05_ab_wide_blocks.jsonl" 2>> "$OUT/ab.err"
done
: > "$OUT/r05_ab_masked_tail.jsonl"
for rep in 1 2; do
  TFGX_LIB_PATH="$ROOT/tf_geometric_amd/lib/variants/serial_tail/libtfgx.so" timeout 300 python tools/ab_wide_blocks.py rmat 64,100,128,256 >> "$OUT/r05_ab_masked_tail.jsonl" 2>> "$OUT/ab.err"
  timeout 300 python tools/ab_wide_blocks.py rmat 64,100,128,256 >> "$OUT/r05_ab_masked_tail.jsonl" 2>> "$OUT/ab.err"
done
TFGX_LIB_PATH="$ROOT/tf_geometric_amd/lib/variants/serial_tail/libtfgx.so" timeout 300 python tools/ab_wide_blocks.py uniform 32,64,100,128 >> "$OUT/r05_ab_masked_tail.jsonl" 2>> "$OUT/ab.err"
timeout 300 python tools/ab_wide_blocks.py uniform 32,64,100,128 >> "$OUT/r05_ab_masked_tail.jsonl" 2>> "$OUT/ab.err"
tail -5 "$OUT/ab.err" >&2
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > "$OUT/pytest_gpu.log" 2>&1
tail -30 "$OUT/pytest_gpu.log" >&2
cat "$OUT/r05_ab_wide_blocks.jsonl" "$OUT/r05_ab_masked_tail.jsonl"
