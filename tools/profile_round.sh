#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace/stats pass + separate PMC passes of bench.py, summaries into
# gpurun_out/prof_<TAG>/ (TFGX_ROUND, default r06).  --pmc is never combined with any trace domain other than --kernel-trace (pool rule).
# (bench.py starts tools/line_rate_probe as child processes: rocprofv3 writes one database per process — the LARGEST is the bench itself)
set -u
ROOT="$(pwd)"
TAG="${TFGX_ROUND:-r06}"          # names of the artefacts: profiles/<TAG>_*
OUT="$ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
HEAD="seg_reduce_kernel<4, 32, 1, false, true, false, false, 0>"
TAIL="seg_reduce_kernel<4, 32, 1, false, true, true, false, 0>"
# the whole default command (headline + static layout + R-MAT + configs block; CPU legs skipped: they launch nothing)
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -- python $ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 > "$OUT/bench_under_stats.json" 2> "$OUT/stats.err"
B="python $ROOT/bench.py --no-cpu-baseline --no-rmat --no-configs"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -- $B --steps 3 --warmup 1 > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -- $B --steps 3 --warmup 1 > /dev/null 2> "$OUT/write.err"
cd "$ROOT"
S=$(find "$OUT/stats" -name "*_results.db" -printf "%s %p\n" | sort -nr | head -1 | cut -d" " -f2-)
F=$(find "$OUT/fetch" -name "*_results.db" -printf "%s %p\n" | sort -nr | head -1 | cut -d" " -f2-)
W=$(find "$OUT/write" -name "*_results.db" -printf "%s %p\n" | sort -nr | head -1 | cut -d" " -f2-)
python tools/rocpd_summary.py "$S" > "$OUT/summary_stats.md"
python tools/kernel_dispatch_csv.py "$S" "$HEAD" "$OUT/${TAG}_products_headline_dispatches.csv"
python tools/rocpd_summary.py "$F" "$W" > "$OUT/summary_pmc.md"
# (the stats pass above also runs the R-MAT graph and the configs block through the SAME kernel symbol: kernel_ms of the pmc
#  json must come from a trace of the plain command — a second, short stats pass)
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats2" -- $B --steps 20 --warmup 5 > /dev/null 2> "$OUT/stats2.err"
cd "$ROOT"
S2=$(find "$OUT/stats2" -name "*_results.db" -printf "%s %p\n" | sort -nr | head -1 | cut -d" " -f2-)
python tools/make_pmc_json.py "$F" "$W" "$S2" "$HEAD" "$OUT/${TAG}_products_pmc.json" products
python tools/make_pmc_json.py "$F" "$W" "$S2" "$TAIL" "$OUT/${TAG}_products_edge_tail_pmc.json" products
rm -rf "$OUT/stats" "$OUT/stats2" "$OUT/fetch" "$OUT/write"
# the plain default line (what the driver runs) and the extras, unprofiled
timeout 600 python bench.py --steps 20 --warmup 5 > "$OUT/${TAG}_bench_products.json" 2> "$OUT/bench.err"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rmat --no-configs --extras > "$OUT/${TAG}_bench_extras.json" 2>> "$OUT/bench.err"
timeout 600 python bench.py --workload arxiv --steps 50 --warmup 10 --extras > "$OUT/${TAG}_bench_arxiv_extras.json" 2>> "$OUT/bench.err"
ls -la "$OUT"
tail -3 "$OUT/bench.err"
