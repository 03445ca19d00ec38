#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 23: final tree — smoke, whole GPU suite, kernel trace of the default bench, the default line.
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/r05_call23"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1
tail -2 "$OUT/smoke.log" >&2
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > "$OUT/pytest_gpu.log" 2>&1
tail -4 "$OUT/pytest_gpu.log" >&2
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -- python $ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 > "$OUT/r05_bench_under_rocprof_final.json" 2> "$OUT/stats.err"
cd "$ROOT"
S=$(find "$OUT/stats" -name "*_results.db" | head -1)
python tools/rocpd_summary.py "$S" > "$OUT/r05_rocprof_final.md"
rm -rf "$OUT/stats"
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/r05_bench_products_final3.json" 2> "$OUT/bench.err" ) 2> "$OUT/bench.time"
cat "$OUT/bench.time" >&2
head -30 "$OUT/r05_rocprof_final.md" >&2
head -c 400 "$OUT/r05_bench_products_final3.json"
