#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace/stats pass + separate PMC passes of bench.py, summaries into
# gpurun_out/prof_r03/.  --pmc is never combined with any trace domain other than --kernel-trace (pool rule).
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/prof_r03"
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
python tools/kernel_dispatch_csv.py "$S" "seg_reduce_kernel<4, 32, 1, false, true, false, false>" "$OUT/r03_products_headline_dispatches.csv"
python tools/rocpd_summary.py "$F" "$W" > "$OUT/summary_pmc.md"
python tools/make_pmc_json.py "$F" "$W" "$S" "seg_reduce_kernel<4, 32, 1, false, true, false, false>" "$OUT/r03_products_pmc.json" products
python tools/make_pmc_json.py "$F" "$W" "$S" "seg_reduce_kernel<4, 32, 1, false, true, true, false>" "$OUT/r03_products_edge_tail_pmc.json" products
# the fused aggregate -> GEMM launch against the two launches it replaces: HBM-side bytes per kernel (separate PMC passes)
cd /tmp
rocprofv3 --pmc FETCH_SIZE -d "$OUT/fused_fetch" -- python "$ROOT/tools/fused_layer_pmc.py" > /dev/null 2> "$OUT/fused_fetch.err"
rocprofv3 --pmc WRITE_SIZE -d "$OUT/fused_write" -- python "$ROOT/tools/fused_layer_pmc.py" > /dev/null 2> "$OUT/fused_write.err"
cd "$ROOT"
python tools/rocpd_summary.py "$(find "$OUT/fused_fetch" -name "*_results.db" | head -1)" "$(find "$OUT/fused_write" -name "*_results.db" | head -1)" > "$OUT/summary_fused_pmc.md"
rm -rf "$OUT/fused_fetch" "$OUT/fused_write"
# the other BASELINE configs: Reddit-shaped GAT and the papers100M-shaped shard, kernel-trace only
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/reddit" -- python "$ROOT/tools/bench_sweep.py" --only=reddit > "$OUT/reddit.jsonl" 2> "$OUT/reddit.err"
rocprofv3 --kernel-trace --stats -d "$OUT/papers" -- python "$ROOT/tools/bench_sweep.py" --only=papers_shard > "$OUT/papers_shard.jsonl" 2> "$OUT/papers.err"
cd "$ROOT"
python tools/rocpd_summary.py "$(find "$OUT/reddit" -name "*_results.db" | head -1)" > "$OUT/summary_reddit.md"
python tools/rocpd_summary.py "$(find "$OUT/papers" -name "*_results.db" | head -1)" > "$OUT/summary_papers.md"
# keep the merge-back small: the raw databases stay on the box
rm -rf "$OUT/stats" "$OUT/fetch" "$OUT/write" "$OUT/reddit" "$OUT/papers"
ls -la "$OUT"
