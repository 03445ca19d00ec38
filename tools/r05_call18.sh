#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 18: generic GEMM with LDS operands read a step ahead (main) vs per-step reads (lds0).
set -u
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r05_gemm_lds"; mkdir -p "$OUT"; cd "$ROOT"
for v in main lds0 main lds0; do
  if [ "$v" = "main" ]; then unset TFGX_LIB_PATH; else export TFGX_LIB_PATH="$ROOT/tf_geometric_amd/lib/variants/$v/libtfgx.so"; fi
  timeout 200 python tools/gemm_generic_ab.py 2>/dev/null | sed "s/^{/{\"lib\": \"$v\", /" | tee -a "$OUT/ab_wide.jsonl"
  timeout 200 python tools/gemm_skinny_ab.py 2>/dev/null | sed "s/^{/{\"lib\": \"$v\", /" >> "$OUT/ab_narrow.jsonl"
done
unset TFGX_LIB_PATH
timeout 400 python -m pytest tests/test_gpu_layers.py -m gpu -x -q -k gemm 2>&1 | tail -3
