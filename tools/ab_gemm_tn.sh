#!/bin/bash
# ON THE GPU BOX: A/B of gemm_tn_kernel's LDS-read batching (TFGX_TN_BATCH) and workgroup count (TFGX_TN_WGS).
OUT=gpurun_out/r02_ab_gemm_tn.jsonl
: > $OUT
for cfg in "1 512" "4 512" "8 512" "4 768" "4 1024" "2 512"; do
  set -- $cfg
  touch tf_geometric_amd/csrc/tfgx_gemm.hip
  TFGX_EXTRA_HIPCC_FLAGS="-DTFGX_TN_BATCH=$1 -DTFGX_TN_WGS=$2" python -c "import sys; sys.path.insert(0,'.'); from tf_geometric_amd import _build; _build.build(verbose=False)" 2>&1 | tail -2
  python - "$1" "$2" >> $OUT <<'PY'
import sys, json, torch
sys.path.insert(0, '.')
from tf_geometric_amd.plan import gemm_tn
n = 2400000
for (ka, nn) in [(100, 256), (256, 40), (128, 128), (256, 256), (100, 16)]:
    x = torch.randn(n, ka, device="cuda"); g = torch.randn(n, nn, device="cuda")
    for _ in range(3): gemm_tn(x, g, want_bias=True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): gemm_tn(x, g, want_bias=True)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(json.dumps({"batch": int(sys.argv[1]), "wgs": int(sys.argv[2]), "K": ka, "N": nn, "ms": round(ms, 4),
                      "TFLOPs": round(2.0 * n * ka * nn / ms / 1e9, 1), "GBps": round(4.0 * n * (ka + nn) / ms / 1e6, 0)}))
PY
done
cat $OUT
