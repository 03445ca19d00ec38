#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 13: edges in flight per lane group in the GAT kernels (4 vs 8), Reddit shape, source blocks on.
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/r05_call13"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
: > "$OUT/r05_gat_unroll_ab.jsonl"
for rep in 1 2; do
  for v in main gat_u8_fwd gat_u8_both; do
    if [ "$v" = "main" ]; then unset TFGX_LIB_PATH; else export TFGX_LIB_PATH="$ROOT/tf_geometric_amd/lib/variants/$v/libtfgx.so"; fi
    timeout 300 python tools/bench_gat_blocks.py 2>> "$OUT/err.log" | sed "s/^{/{\"lib\": \"$v\", /" >> "$OUT/r05_gat_unroll_ab.jsonl"
  done
done
unset TFGX_LIB_PATH
grep -v amdgpu.ids "$OUT/err.log" | tail -3 >&2
cat "$OUT/r05_gat_unroll_ab.jsonl"
