#!/bin/bash
# Runs ON THE GPU BOX (via gpurun), round 5 call 2: column-block variants of the wide-row walk + piece probes; the default
# bench.py line with the new configs block; bench / dist GPU tests (watchdog, 8-rank plumbing at products shape).
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/r05_wide2"
mkdir -p "$OUT"
export TMPDIR=/tmp
H="$ROOT/tf_geometric_amd/lib/wide_row_ab"
export LD_LIBRARY_PATH="$ROOT/tf_geometric_amd/lib:${LD_LIBRARY_PATH:-}"
timeout 400 $H 2400000 123000000 64,100,128,192,256,384,512 "shipped|blk|piece probe|ysplit" 5 > "$OUT/r05_wide_row_ab_blocks.jsonl" 2> "$OUT/harness.err"
tail -3 "$OUT/harness.err" >&2
cd "$ROOT"
( time timeout 600 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" ) 2> "$OUT/bench_default.time"
tail -5 "$OUT/bench_default.err" >&2
cat "$OUT/bench_default.time" >&2
timeout 1500 python -m pytest tests/test_gpu_bench.py tests/test_gpu_dist.py tests/test_gpu_c_abi.py -x -q > "$OUT/pytest_bench_dist.log" 2>&1
tail -15 "$OUT/pytest_bench_dist.log" >&2
cat "$OUT/r05_wide_row_ab_blocks.jsonl"
