#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): per variant of tools/rmat_pmc.py one --kernel-trace pass and two separate --pmc passes
# (never combined with another trace domain), then tools/rmat_pmc_collect.py -> gpurun_out/rmat_pmc/r04_rmat_pmc.json
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/rmat_pmc"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for v in ${VARIANTS:-uniform rmat rmat_thr1024 rmat_thr4096 rmat_nohub rmat_row_order}; do
  timeout 240 rocprofv3 --kernel-trace --stats -d "$OUT/$v/stats" -- python "$ROOT/tools/rmat_pmc.py" $v 5 > "$OUT/$v.json" 2> "$OUT/$v.err"
  timeout 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/$v/hit" -- python "$ROOT/tools/rmat_pmc.py" $v 2 > "$OUT/$v.hit.meta" 2>> "$OUT/$v.err"
  timeout 240 rocprofv3 --pmc FETCH_SIZE -d "$OUT/$v/fetch" -- python "$ROOT/tools/rmat_pmc.py" $v 2 > "$OUT/$v.fetch.meta" 2>> "$OUT/$v.err"
done
cd "$ROOT"
python tools/rmat_pmc_collect.py "$OUT" "$OUT/r04_rmat_pmc.json" > "$OUT/collect.log" 2>&1
for v in $(ls -d "$OUT"/*/ ); do rm -rf "$v"; done
cat "$OUT/collect.log"
