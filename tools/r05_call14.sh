#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 14: final validation — smoke, whole GPU suite, the default bench line.
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/r05_call14"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1
tail -2 "$OUT/smoke.log" >&2
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > "$OUT/pytest_gpu.log" 2>&1
tail -8 "$OUT/pytest_gpu.log" >&2
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/r05_bench_products_final.json" 2> "$OUT/bench.err" ) 2> "$OUT/bench.time"
cat "$OUT/bench.time" >&2
TFGX_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --workload tiny --steps 3 --warmup 1 > "$OUT/r05_bench_2rank_plumbing.json" 2>> "$OUT/bench.err"
grep -v amdgpu.ids "$OUT/bench.err" | tail -3 >&2
head -c 700 "$OUT/r05_bench_products_final.json"
