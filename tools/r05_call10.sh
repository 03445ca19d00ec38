#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 10: whole GPU suite, then the profile round (tools/profile_round.sh) on the same tree.
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/r05_call10"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > "$OUT/pytest_gpu.log" 2>&1
tail -12 "$OUT/pytest_gpu.log" >&2
bash tools/profile_round.sh > "$OUT/profile_round.log" 2>&1
tail -5 "$OUT/profile_round.log" >&2
