#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 6: is it the power-of-two row stride? (R-MAT at F = 128 / 256 with rows 32 floats further
# apart + L2 hit-rate passes), the papers100M-shaped shard with and without column blocks, the default bench line.
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/r05_call6"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
: > "$OUT/r05_ab_ld_pad.jsonl"
for rep in 1 2; do
  for g in rmat uniform; do
    AB_LD_PAD=0 timeout 300 python tools/ab_wide_blocks.py $g 128,256,512 >> "$OUT/r05_ab_ld_pad.jsonl" 2>> "$OUT/ab.err"
    AB_LD_PAD=32 timeout 300 python tools/ab_wide_blocks.py $g 128,256,512 >> "$OUT/r05_ab_ld_pad.jsonl" 2>> "$OUT/ab.err"
  done
done
cd /tmp
for v in "224 0" "256 0" "256 32"; do
  set -- $v
  RMAT_F=$1 RMAT_LD_PAD=$2 timeout 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d "$OUT/hit_$1_$2" -- python "$ROOT/tools/rmat_pmc.py" rmat 2 > "$OUT/rmat_hit_F$1_pad$2.json" 2>> "$OUT/pmc.err"
  db=$(find "$OUT/hit_$1_$2" -name "*_results.db" | head -1)
  if [ -n "$db" ]; then python "$ROOT/tools/rocpd_summary.py" "$db" | grep "seg_reduce\|hub_finalize\|counter" > "$OUT/rmat_hit_F$1_pad$2.md"; fi
  rm -rf "$OUT/hit_$1_$2"
done
cd "$ROOT"
: > "$OUT/r05_papers_shard.jsonl"
TFGX_REDUCE_WIDE_BLOCKS=0 timeout 400 python tools/bench_sweep.py --only=papers_shard >> "$OUT/r05_papers_shard.jsonl" 2>> "$OUT/ab.err"
timeout 400 python tools/bench_sweep.py --only=papers_shard >> "$OUT/r05_papers_shard.jsonl" 2>> "$OUT/ab.err"
( time timeout 600 python bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" ) 2> "$OUT/bench_default.time"
grep -v amdgpu.ids "$OUT/ab.err" | tail -3 >&2
tail -3 "$OUT/bench_default.err" >&2
cat "$OUT/r05_papers_shard.jsonl" "$OUT"/rmat_hit_*.md
