#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 19: generic GEMM — double-buffered LDS + unguarded interior loads + operands a step ahead
# (main) vs each piece switched off (bufs1, int0) vs the round-4 loop (old).
set -u
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r05_gemm_lds"; mkdir -p "$OUT"; cd "$ROOT"
timeout 400 python -m pytest tests/test_gpu_layers.py -m gpu -x -q -k gemm 2>&1 | tail -3
for v in main old bufs1 int0 main old; do
  if [ "$v" = "main" ]; then unset TFGX_LIB_PATH; else export TFGX_LIB_PATH="$ROOT/tf_geometric_amd/lib/variants/$v/libtfgx.so"; fi
  timeout 200 python tools/gemm_generic_ab.py 2>/dev/null | sed "s/^{/{\"lib\": \"$v\", /" | tee -a "$OUT/ab2_wide.jsonl" | cut -c1-160
  timeout 200 python tools/gemm_skinny_ab.py 2>/dev/null | sed "s/^{/{\"lib\": \"$v\", /" >> "$OUT/ab2_narrow.jsonl"
done
