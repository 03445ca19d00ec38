#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 15: the norm kernel's unrolled edge loop against the goldens and the whole GPU suite,
# then the default bench line (R-MAT plan build time included).
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/r05_call15"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
timeout 600 python -m pytest tests/test_gpu_reference_golden.py tests/test_gpu_aggregate.py -m gpu -x -q > "$OUT/pytest_norm.log" 2>&1
tail -3 "$OUT/pytest_norm.log" >&2
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > "$OUT/pytest_gpu.log" 2>&1
tail -8 "$OUT/pytest_gpu.log" >&2
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/r05_bench_products_final2.json" 2> "$OUT/bench.err" ) 2> "$OUT/bench.time"
cat "$OUT/bench.time" >&2
grep -v amdgpu.ids "$OUT/bench.err" | tail -3 >&2
head -c 700 "$OUT/r05_bench_products_final2.json"
