# coding=utf-8
"""Narrow-output products beside torch.matmul (hipBLASLt), one setting of TFGX_GEMM_SKINNY per process (0 = the kernels of
rounds 1-4, 1 = gemm_skinny_kernel where the row-streaming kernel cannot go, 2 = gemm_skinny_kernel for every N <= 48).  One JSON line per shape."""
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
SHAPES = [(173312, 1433, 16), (233000, 602, 16), (233000, 602, 8), (2400000, 100, 16), (2400000, 100, 40), (2400000, 256, 40),
          (170000, 256, 40), (2400000, 128, 40), (233000, 602, 41), (2400000, 256, 47), (2400000, 100, 32)]
for (m, k, n) in SHAPES:
    a = torch.randn(m, k, device="cuda")
    b = torch.randn(k, n, device="cuda") * 0.1
    c = torch.empty(m, n, device="cuda")
    rounds = []
    for _ in range(3):
        rounds.append((timeit(lambda: gemm_bias_act(a, b, out=c)), timeit(lambda: torch.matmul(a, b, out=c))))
    ours, lib = sorted(r[0] for r in rounds)[1], sorted(r[1] for r in rounds)[1]
    ref = torch.matmul(a.double(), b.double())
    err = float((gemm_bias_act(a, b).double() - ref).abs().max())
    err_lib = float((torch.matmul(a, b).double() - ref).abs().max())
    print(json.dumps({"skinny": os.environ.get("TFGX_GEMM_SKINNY", "1"), "M": m, "K": k, "N": n, "ms": round(ours, 4),
                      "torch_matmul_ms": round(lib, 4), "ratio_ours_over_torch": round(ours / lib, 3),
                      "A_TBps": round(4.0 * m * k / ours / 1e9, 3), "max_abs_err_vs_f64": err, "torch_max_abs_err_vs_f64": err_lib}), flush=True)
    del a, b, c
