#!/bin/bash
# Runs ON THE GPU BOX: the tree as it is — smoke, the whole GPU suite, the default bench line.
#   gpurun --timeout 2400 -- 'bash tools/final_validation.sh [name]'   -> gpurun_out/<name>/{smoke.log, pytest_gpu.log, bench.json}
set -u
NAME="${1:-final_validation}"
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/$NAME"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log" >&2
timeout 2000 python -m pytest tests -m gpu -x -q --durations=5 > "$OUT/pytest_gpu.log" 2>&1
grep -n "passed\|failed" "$OUT/pytest_gpu.log" | tail -2 >&2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["frac"])
for k, v in d.get("configs", {}).items():
    if isinstance(v, dict):
        print(k, {a: round(b, 3) for a, b in v.items() if a.endswith("_ms") and isinstance(b, (int, float))}, (v.get("roofline") or {}).get("bound"), (v.get("roofline") or {}).get("frac"))
PY
