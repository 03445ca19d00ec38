#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 20: generic GEMM, LDS operands a step ahead for K >= 256 (default) vs off; GEMM + layer tests.
set -u
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r05_gemm_lds"; mkdir -p "$OUT"; cd "$ROOT"
timeout 600 python -m pytest tests/test_gpu_layers.py tests/test_gpu_backward.py tests/test_gpu_reference_golden.py -m gpu -x -q 2>&1 | tail -3
for v in 1 0 1 0; do
  TFGX_GEMM_LDS_AHEAD=$v timeout 200 python tools/gemm_generic_ab.py 2>/dev/null | tee -a "$OUT/r05_gemm_generic_ab.jsonl" | cut -c1-150
done
