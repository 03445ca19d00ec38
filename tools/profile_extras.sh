#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 --kernel-trace --stats of `bench.py --extras` (whole layers, training steps, hipGraph replays)
# at products shape and of the Reddit-shaped GAT sweep; per-dispatch rows of EVERY tfgx kernel -> small CSVs for profiles/.
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/prof_r04"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/extras" -- python "$ROOT/bench.py" --extras --no-cpu-baseline --no-rmat --steps 5 --warmup 2 > "$OUT/bench_extras_under_stats.json" 2> "$OUT/extras.err"
rocprofv3 --kernel-trace --stats -d "$OUT/reddit" -- python "$ROOT/tools/bench_sweep.py" --only=reddit > "$OUT/reddit.jsonl" 2> "$OUT/reddit.err"
cd "$ROOT"
E=$(find "$OUT/extras" -name "*_results.db" | head -1)
R=$(find "$OUT/reddit" -name "*_results.db" | head -1)
python tools/rocpd_summary.py "$E" > "$OUT/summary_extras.md"
python tools/rocpd_summary.py "$R" > "$OUT/summary_reddit.md"
python tools/kernel_dispatch_csv.py "$E" "tfgx::" "$OUT/r04_extras_dispatches.csv"
python tools/kernel_dispatch_csv.py "$R" "tfgx::" "$OUT/r04_reddit_dispatches.csv"
rm -rf "$OUT/extras" "$OUT/reddit"
ls -la "$OUT" | tail -12
