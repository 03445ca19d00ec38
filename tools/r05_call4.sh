#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 4: wide-row column blocks in pass order vs rotated over the workgroups (uniform + R-MAT),
# the source-blocked GAT attention experiment, the kernel-facing GPU tests.
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/r05_call4"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
: > "$OUT/r05_ab_wide_blocks.jsonl"
W=128,192,224,256,320,384,512,1024
for rep in 1 2; do
  for mode in 0 1 2; do
    TFGX_REDUCE_WIDE_BLOCKS=$mode timeout 300 python tools/ab_wide_blocks.py uniform $W >> "$OUT/r05_ab_wide_blocks.jsonl" 2>> "$OUT/ab.err"
  done
  TFGX_REDUCE_WIDE_BLOCKS=2 TFGX_REDUCE_WIDE_G256=16 timeout 300 python tools/ab_wide_blocks.py uniform 256 >> "$OUT/r05_ab_wide_blocks.jsonl" 2>> "$OUT/ab.err"
done
: > "$OUT/r05_ab_wide_blocks_rmat.jsonl"
for rep in 1 2; do
  for mode in 0 1 2; do
    TFGX_REDUCE_WIDE_BLOCKS=$mode timeout 300 python tools/ab_wide_blocks.py rmat 100,128,192,224,256,512 >> "$OUT/r05_ab_wide_blocks_rmat.jsonl" 2>> "$OUT/ab.err"
  done
done
timeout 300 python tools/gat_source_blocks.py 8 4,8,16,32,64 > "$OUT/r05_gat_source_blocks.jsonl" 2>> "$OUT/ab.err"
timeout 300 python tools/gat_source_blocks.py 64 8,16,32 >> "$OUT/r05_gat_source_blocks.jsonl" 2>> "$OUT/ab.err"
grep -v amdgpu.ids "$OUT/ab.err" | tail -5 >&2
timeout 900 python -m pytest tests/test_gpu_aggregate.py tests/test_gpu_fullsize.py tests/test_gpu_layers.py tests/test_gpu_backward.py tests/test_gpu_reference_golden.py tests/test_gpu_fuzz.py -x -q > "$OUT/pytest_kernels.log" 2>&1
tail -4 "$OUT/pytest_kernels.log" >&2
cat "$OUT/r05_gat_source_blocks.jsonl"
