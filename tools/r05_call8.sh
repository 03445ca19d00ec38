#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 8: the long-K narrow-output GEMM (tests + A/B), max-pool layer after the gn stride change.
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/r05_call8"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
timeout 600 python -m pytest tests/test_gpu_layers.py tests/test_gpu_backward.py -x -q -k "gemm or max or pool or gat or Pool or sage" > "$OUT/pytest.log" 2>&1
tail -5 "$OUT/pytest.log" >&2
: > "$OUT/r05_gemm_skinny_ab.jsonl"
for rep in 1 2; do
  TFGX_GEMM_SKINNY=0 timeout 300 python tools/gemm_skinny_ab.py >> "$OUT/r05_gemm_skinny_ab.jsonl" 2>> "$OUT/err.log"
  TFGX_GEMM_SKINNY=1 timeout 300 python tools/gemm_skinny_ab.py >> "$OUT/r05_gemm_skinny_ab.jsonl" 2>> "$OUT/err.log"
done
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-rmat > "$OUT/bench_configs.json" 2>> "$OUT/err.log"
grep -v amdgpu.ids "$OUT/err.log" | tail -5 >&2
cat "$OUT/r05_gemm_skinny_ab.jsonl"
