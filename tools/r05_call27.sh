#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 27: gat_fused_kernel with one exponential per edge + multiply by the exact inverse of a
# power-of-two scale (main) vs two exponentials (exp2) vs always dividing (div).
set -u
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r05_gat_valu"; mkdir -p "$OUT"; cd "$ROOT"
timeout 500 python -m pytest tests/test_gpu_layers.py tests/test_gpu_backward.py tests/test_gpu_reference_golden.py tests/test_gpu_fullsize.py -m gpu -x -q -k "gat or GAT or attention or reddit" 2>&1 | tail -2
for v in main; do
  if [ "$v" = "main" ]; then unset TFGX_LIB_PATH; else export TFGX_LIB_PATH="$ROOT/tf_geometric_amd/lib/variants/$v/libtfgx.so"; fi
  timeout 300 python tools/bench_gat_blocks.py 2>/dev/null | sed "s/^{/{\"lib\": \"$v\", /" | tee -a "$OUT/r05_gat_valu_ab.jsonl" | cut -c1-215
done
