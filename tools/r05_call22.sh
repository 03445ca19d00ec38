#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 22: generic GEMM long-K instantiation with one (main) vs two (pf2) tiles of global loads in flight.
set -u
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r05_gemm_lds"; mkdir -p "$OUT"; cd "$ROOT"
for v in main pf2 main pf2; do
  if [ "$v" = "main" ]; then unset TFGX_LIB_PATH; else export TFGX_LIB_PATH="$ROOT/tf_geometric_amd/lib/variants/$v/libtfgx.so"; fi
  timeout 200 python tools/gemm_generic_ab.py 2>/dev/null | sed "s/^{/{\"lib\": \"$v\", /" | tee -a "$OUT/r05_gemm_prefetch_ab.jsonl" | cut -c1-165
done
unset TFGX_LIB_PATH
TFGX_LIB_PATH="$ROOT/tf_geometric_amd/lib/variants/pf2/libtfgx.so" timeout 400 python -m pytest tests/test_gpu_layers.py -m gpu -x -q -k gemm 2>&1 | tail -2
