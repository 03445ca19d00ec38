#!/bin/bash
# Runs ON THE GPU BOX: HBM-side bytes of the fused aggregate -> GEMM launch vs the two launches at F = 128 -> 256 and 100 -> 256
# (separate --pmc FETCH_SIZE / WRITE_SIZE passes) -> gpurun_out/prof_r04/summary_fused_pmc_F*.md
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/prof_r04"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for F in 128 100; do
  rocprofv3 --pmc FETCH_SIZE -d "$OUT/ff_$F" -- python "$ROOT/tools/fused_layer_pmc.py" $F > /dev/null 2> "$OUT/ff_$F.err"
  rocprofv3 --pmc WRITE_SIZE -d "$OUT/fw_$F" -- python "$ROOT/tools/fused_layer_pmc.py" $F > /dev/null 2>> "$OUT/ff_$F.err"
  python "$ROOT/tools/rocpd_summary.py" "$(find "$OUT/ff_$F" -name "*_results.db" | head -1)" "$(find "$OUT/fw_$F" -name "*_results.db" | head -1)" > "$OUT/summary_fused_pmc_F$F.md"
  rm -rf "$OUT/ff_$F" "$OUT/fw_$F"
done
cd "$ROOT"
grep -E "agg_gemm|seg_reduce|gemm_rows" "$OUT"/summary_fused_pmc_F*.md | grep -E "FETCH|WRITE" | cut -c1-260
