#!/bin/bash
# Runs ON THE GPU BOX (via gpurun), round 5 call 1: wide-row A/B harness, counter passes of the shipped kernel at F = 128 / 256 / 512,
# occupancy probe at F = 256, R-MAT vs uniform at F = 128 / 256 with the per-kernel split.  --pmc passes are separate runs and are
# never combined with a trace domain other than --kernel-trace.
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/r05_wide"
mkdir -p "$OUT"
export TMPDIR=/tmp
H="$ROOT/tf_geometric_amd/lib/wide_row_ab"
export LD_LIBRARY_PATH="$ROOT/tf_geometric_amd/lib:${LD_LIBRARY_PATH:-}"

echo "== harness: all variants" >&2
timeout 420 $H 2400000 123000000 100,128,192,256,384,512 all 5 > "$OUT/r05_wide_row_ab.jsonl" 2> "$OUT/harness.err"
tail -3 "$OUT/harness.err" >&2

echo "== occupancy probe, shipped kernel, F = 256 / 128" >&2
: > "$OUT/r05_occupancy_wide.jsonl"
for lds in 0 27000 40000 54000 80000; do
  echo "{\"dummy_lds\": $lds}" >> "$OUT/r05_occupancy_wide.jsonl"
  TFGX_REDUCE_DUMMY_LDS=$lds timeout 120 $H 2400000 123000000 128,256,512 shipped 5 >> "$OUT/r05_occupancy_wide.jsonl" 2>> "$OUT/harness.err"
done

echo "== counter passes" >&2
SEL="shipped|G64 CH1 U8 mask early rot|G32 CH1 U8 mask early rot|G64 CH2 U4 mask early rot nocap|probe 1KB rows U8|probe 512B rows U8"
cd /tmp
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_avr" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set -d "$OUT/pmc$i" -- $H 2400000 123000000 128,256,512 "$SEL" 2 > "$OUT/pmc$i.out" 2> "$OUT/pmc$i.err"
  db=$(find "$OUT/pmc$i" -name "*_results.db" | head -1)
  if [ -n "$db" ]; then python "$ROOT/tools/rocpd_summary.py" "$db" > "$OUT/pmc$i.md" 2>> "$OUT/pmc$i.err"; fi
  rm -rf "$OUT/pmc$i"
done
# kernel trace of the same selection (durations to price the counters with)
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -- $H 2400000 123000000 128,256,512 "$SEL" 2 > "$OUT/kt.out" 2> "$OUT/kt.err"
db=$(find "$OUT/kt" -name "*_results.db" | head -1)
if [ -n "$db" ]; then python "$ROOT/tools/rocpd_summary.py" "$db" > "$OUT/kt.md"; fi
rm -rf "$OUT/kt"

echo "== R-MAT vs uniform at F = 128 / 256 (product path), per-kernel split" >&2
cd /tmp
for f in 128 256; do
  RMAT_F=$f timeout 200 python "$ROOT/tools/rmat_pmc.py" uniform 5 > "$OUT/rmat_uniform_F$f.json" 2> "$OUT/rmat.err"
  RMAT_F=$f timeout 240 rocprofv3 --kernel-trace --stats -d "$OUT/rk$f" -- python "$ROOT/tools/rmat_pmc.py" rmat 5 > "$OUT/rmat_rmat_F$f.json" 2>> "$OUT/rmat.err"
  db=$(find "$OUT/rk$f" -name "*_results.db" | head -1)
  if [ -n "$db" ]; then python "$ROOT/tools/rocpd_summary.py" "$db" > "$OUT/rmat_rmat_F$f.kernels.md"; fi
  rm -rf "$OUT/rk$f"
done
cd "$ROOT"
ls -la "$OUT" >&2
cat "$OUT/r05_wide_row_ab.jsonl"
