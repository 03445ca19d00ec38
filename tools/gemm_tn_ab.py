# coding=utf-8
"""Weight-gradient kernel (tfgx_gemm_tn_f32: dW = x^T g, db = column sums) at products shape: ms / TFLOP/s / GB/s per shape.
    [TFGX_LIB_PATH=...variants/<name>/libtfgx.so] python tools/gemm_tn_ab.py [tag]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_geometric_amd.plan import gemm_tn      # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "tree"
n = 2400000
for (ka, nn, gated) in [(100, 256, False), (256, 47, False), (256, 40, False), (128, 128, False), (256, 256, False), (100, 16, False), (100, 256, True)]:
    x = torch.randn(n, ka, device="cuda")
    g = torch.randn(n, nn, device="cuda")
    gate = torch.randn(n, nn, device="cuda") if gated else None
    fn = lambda: gemm_tn(x, g, want_bias=True, gate=gate)      # noqa: E731
    for _ in range(3):
        fn()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 10)
    ms = sorted(ts)[1]
    print(json.dumps({"tag": tag, "K": ka, "N": nn, "gated": gated, "ms": round(ms, 4), "TFLOPs": round(2.0 * n * ka * nn / ms / 1e9, 1),
                      "GBps": round(4.0 * n * (ka + nn * (2 if gated else 1)) / ms / 1e6, 0)}), flush=True)
