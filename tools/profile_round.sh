#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace/stats pass + separate PMC passes of bench.py, summaries into
# gpurun_out/prof_r04/.  --pmc is never combined with any trace domain other than --kernel-trace (pool rule).
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/prof_r04"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="python $ROOT/bench.py --no-cpu-baseline --no-rmat"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -- $B --steps 10 --warmup 3 > "$OUT/bench_under_stats.json" 2> "$OUT/stats.err"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -- $B --steps 3 --warmup 1 > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -- $B --steps 3 --warmup 1 > /dev/null 2> "$OUT/write.err"
cd "$ROOT"
S=$(find "$OUT/stats" -name "*_results.db" | head -1)
F=$(find "$OUT/fetch" -name "*_results.db" | head -1)
W=$(find "$OUT/write" -name "*_results.db" | head -1)
python tools/rocpd_summary.py "$S" > "$OUT/summary_stats.md"
# per-dispatch rows of the headline kernel: roofline.frac can be re-derived from this small CSV without the raw database
python tools/kernel_dispatch_csv.py "$S" "seg_reduce_kernel<4, 32, 1, false, true, false, false>" "$OUT/r04_products_headline_dispatches.csv"
python tools/rocpd_summary.py "$F" "$W" > "$OUT/summary_pmc.md"
python tools/make_pmc_json.py "$F" "$W" "$S" "seg_reduce_kernel<4, 32, 1, false, true, false, false>" "$OUT/r04_products_pmc.json" products
python tools/make_pmc_json.py "$F" "$W" "$S" "seg_reduce_kernel<4, 32, 1, false, true, true, false>" "$OUT/r04_products_edge_tail_pmc.json" products
# keep the merge-back small: the raw databases stay on the box
rm -rf "$OUT/stats" "$OUT/fetch" "$OUT/write"
ls -la "$OUT"
