#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 5: GAT in source blocks (tests + Reddit-shape timings), wide-block policy on R-MAT, kernel tests.
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/r05_call5"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_backward.py tests/test_gpu_fullsize.py tests/test_gpu_aggregate.py tests/test_gpu_reference_golden.py tests/test_gpu_regressions.py -x -q > "$OUT/pytest_kernels.log" 2>&1
tail -6 "$OUT/pytest_kernels.log" >&2
# Reddit-shape GAT: one pass vs source blocks, alternating in one process
timeout 600 python tools/bench_gat_blocks.py > "$OUT/r05_reddit_gat.jsonl" 2> "$OUT/gat.err"
grep -v amdgpu.ids "$OUT/gat.err" | tail -5 >&2
: > "$OUT/r05_ab_wide_blocks_rmat_policy.jsonl"
for rep in 1 2; do
  TFGX_REDUCE_WIDE_BLOCKS=0 timeout 300 python tools/ab_wide_blocks.py rmat 128,192,224,256 >> "$OUT/r05_ab_wide_blocks_rmat_policy.jsonl" 2>> "$OUT/ab.err"
  timeout 300 python tools/ab_wide_blocks.py rmat 128,192,224,256 >> "$OUT/r05_ab_wide_blocks_rmat_policy.jsonl" 2>> "$OUT/ab.err"
done
cat "$OUT/r05_reddit_gat.jsonl"
