#!/bin/bash
# Runs ON THE GPU BOX, round 5 call 9: narrow-output GEMM at N <= 48 (tests + three-way A/B), norm kernels on long rows (tests + R-MAT plan time)
set -u
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/r05_call9"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_layers.py tests/test_gpu_reference_golden.py tests/test_gpu_aggregate.py tests/test_gpu_fullsize.py tests/test_gpu_plan.py tests/test_gpu_dist.py -x -q > "$OUT/pytest.log" 2>&1
tail -5 "$OUT/pytest.log" >&2
: > "$OUT/r05_gemm_skinny_ab.jsonl"
for mode in 0 1 2 0 1 2; do
  TFGX_GEMM_SKINNY=$mode timeout 300 python tools/gemm_skinny_ab.py >> "$OUT/r05_gemm_skinny_ab.jsonl" 2>> "$OUT/err.log"
done
timeout 300 python tools/rmat_pmc.py rmat 3 > "$OUT/rmat_plan.json" 2>> "$OUT/err.log"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs > "$OUT/bench_rmat.json" 2>> "$OUT/err.log"
grep -v amdgpu.ids "$OUT/err.log" | tail -5 >&2
python - <<'PY'
import json
l=json.load(open("gpurun_out/r05_call9/bench_rmat.json"))
print("rmat plan_build_s", l["rmat"]["plan_build_s"], "kernel_ms", l["rmat"]["kernel_ms"])
PY
