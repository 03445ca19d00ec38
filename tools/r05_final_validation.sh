#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 44: final tree — whole GPU suite, smoke, default bench line.
set -u
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r05_call44"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log" >&2
timeout 1500 python -m pytest tests -m gpu -x -q --durations=3 > "$OUT/pytest_gpu.log" 2>&1
grep -n "passed\|failed" "$OUT/pytest_gpu.log" | tail -2 >&2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/r05_bench_products_final6.json" 2> "$OUT/bench.err"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05_call44/r05_bench_products_final6.json"))
print(d["ms_per_step"], d["roofline"]["frac"], json.dumps(d["configs"]["C3_reddit_gat_H8_A8"])[:600])
PY
