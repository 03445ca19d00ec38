#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 25: gat_fused_kernel with the next batch's source ids loaded ahead (main) vs not (col0; call 26: bcol0 = the backward kernel without it).
set -u
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r05_gat_col"; mkdir -p "$OUT"; cd "$ROOT"
for v in main bcol0 main bcol0; do
  if [ "$v" = "main" ]; then unset TFGX_LIB_PATH; else export TFGX_LIB_PATH="$ROOT/tf_geometric_amd/lib/variants/$v/libtfgx.so"; fi
  timeout 300 python tools/bench_gat_blocks.py 2>/dev/null | sed "s/^{/{\"lib\": \"$v\", /" | tee -a "$OUT/r05_gat_col_ahead_ab.jsonl" | cut -c1-230
done
unset TFGX_LIB_PATH
timeout 500 python -m pytest tests/test_gpu_layers.py tests/test_gpu_backward.py tests/test_gpu_reference_golden.py -m gpu -x -q -k "gat or GAT or attention" 2>&1 | tail -2
