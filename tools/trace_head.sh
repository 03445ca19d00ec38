#!/bin/bash
# Runs ON THE GPU BOX: kernel trace (rocprofv3 --kernel-trace --stats) of the default bench command on the round's last tree.
set -u
ROOT="$(pwd)"; TAG="${TFGX_ROUND:-r06}"; OUT="$ROOT/gpurun_out/${TAG}_trace_head"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
timeout 170 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -- python $ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 > "$OUT/${TAG}_bench_under_rocprof_head.json" 2> "$OUT/stats.err"
cd "$ROOT"
S=$(find "$OUT/stats" -name "*_results.db" -printf "%s %p\n" | sort -nr | head -1 | cut -d" " -f2-)
python tools/rocpd_summary.py "$S" > "$OUT/${TAG}_rocprof_head.md"
rm -rf "$OUT/stats"
grep -n "gat_\|seg_reduce_kernel<4, 32, 1, false, true, false, false, 0>" "$OUT/${TAG}_rocprof_head.md" | cut -c1-200
