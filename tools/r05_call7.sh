#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 7: whole GPU suite + the default bench line on the current tree.
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/r05_call7"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > "$OUT/pytest_gpu.log" 2>&1
tail -25 "$OUT/pytest_gpu.log" >&2
( time timeout 600 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" ) 2> "$OUT/bench_default.time"
tail -3 "$OUT/bench_default.err" >&2
cat "$OUT/bench_default.time" >&2
