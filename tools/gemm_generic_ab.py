# coding=utf-8
"""The generic (LDS-staged) GEMM beside torch.matmul (hipBLASLt) on shapes the row-streaming kernel cannot take (K % 4 != 0),
one setting of TFGX_GEMM_LDS_AHEAD per process (1 = operands of the next k step read from LDS before the current step's
MFMAs when K >= 256, 0 = per-step reads).  One JSON line per shape."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tf_geometric_amd.plan import gemm_bias_act          # noqa: E402


def timeit(fn, steps=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


torch.manual_seed(0)
SHAPES = [(170000, 1433, 256), (170000, 1433, 200), (233000, 602, 256), (233000, 602, 160), (100000, 301, 256),
          (2400000, 101, 256), (233000, 602, 64), (233000, 602, 128), (170000, 1433, 64), (2708, 1433, 64), (19717, 501, 64)]
for (m, k, n) in SHAPES:
    a = torch.randn(m, k, device="cuda")
    b = torch.randn(k, n, device="cuda") * 0.1
    c = torch.empty(m, n, device="cuda")
    rounds = []
    for _ in range(3):
        rounds.append((timeit(lambda: gemm_bias_act(a, b, out=c)), timeit(lambda: torch.matmul(a, b, out=c))))
    ours, lib = sorted(r[0] for r in rounds)[1], sorted(r[1] for r in rounds)[1]
    ref = torch.matmul(a[:20000].double(), b.double())
    err = float((gemm_bias_act(a, b)[:20000].double() - ref).abs().max())
    print(json.dumps({"lds_ahead": os.environ.get("TFGX_GEMM_LDS_AHEAD", "1"), "M": m, "K": k, "N": n, "ms": round(ours, 4),
                      "torch_matmul_ms": round(lib, 4), "ratio_ours_over_torch": round(ours / lib, 3),
                      "TFLOPs": round(2.0 * m * k * n / ours / 1e9, 1), "max_abs_err_vs_f64": err}), flush=True)
    del a, b, c
