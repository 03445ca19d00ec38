# coding=utf-8
"""The dense x @ W of the layers (tfgx_gemm_bias_act_f32, fp32 MFMA) against torch.matmul (hipBLASLt / rocBLAS fp32) on the
BASELINE shapes, alternating in one process (same box, same clocks).  One JSON line per shape.

    [TFGX_LIB_PATH=.../variants/<name>/libtfgx.so] python tools/gemm_ab.py [tag]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tf_geometric_amd import _lib as L                  # noqa: E402
from tf_geometric_amd.plan import gemm_bias_act         # noqa: E402

SHAPES = [(2400000, 100, 256), (2400000, 100, 128), (2400000, 128, 256), (2400000, 100, 64), (2400000, 256, 128),
          (2400000, 100, 16), (2400000, 256, 256), (170000, 128, 256), (170000, 256, 40), (233000, 602, 64),
          (173312, 1433, 16), (233000, 602, 16), (170000, 1433, 256), (100000, 301, 40), (2400000, 256, 40)]
tag = sys.argv[1] if len(sys.argv) > 1 else "tree"
if os.environ.get("GEMM_AB_SHAPES"):      # "M,K,N;M,K,N;..."
    SHAPES = [tuple(int(v) for v in s.split(",")) for s in os.environ["GEMM_AB_SHAPES"].split(";")]


def t(fn, steps, warmup):
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


L.require_gpu()
g = torch.Generator(device="cuda")
g.manual_seed(0)
for M, K, N in SHAPES:
    a = torch.randn(M, K, generator=g, device="cuda")
    b = torch.randn(K, N, generator=g, device="cuda") * 0.1
    out, ref = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    f_mine = lambda: gemm_bias_act(a, b, out=out)          # noqa: E731
    f_lib = lambda: torch.matmul(a, b, out=ref)            # noqa: E731
    steps = 20 if M * K * N < 2e10 else 10
    mine, lib = [], []
    for r in range(3):
        mine.append(t(f_mine, steps, 3 if r == 0 else 1))
        lib.append(t(f_lib, steps, 3 if r == 0 else 1))
    mine.sort()
    lib.sort()
    err = float((out - ref).abs().max() / ref.abs().max())
    print(json.dumps({"tag": tag, "M": M, "K": K, "N": N, "ms": mine[1], "hipblaslt_ms": lib[1], "ratio": mine[1] / lib[1],
                      "TFLOPs": 2.0 * M * K * N / (mine[1] * 1e-3) / 1e12, "rel_diff_vs_lib": err}), flush=True)
    del a, b, out, ref
