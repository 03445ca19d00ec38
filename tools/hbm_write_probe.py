# coding=utf-8
"""What bounds the tall-skinny GEMM 2.4 M x 100 -> 256 (1.27 ms, 96 TFLOP/s: "at neither roof")?
It reads 0.96 GB and WRITES 2.46 GB.  This probe times plain streaming kernels over the same byte counts on the same box:
a fill (write only), a copy (read + write), a reduction (read only) — and the GEMM itself with narrower outputs.

    python tools/hbm_write_probe.py > gpurun_out/r03/hbm_write_probe.jsonl
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tf_geometric_amd as tfg                              # noqa: E402
from tf_geometric_amd.plan import gemm_bias_act             # noqa: E402
import bench                                                # noqa: E402

n = 2400000
out = torch.empty(n, 256, device="cuda")
src = torch.randn(n, 256, device="cuda")
a = torch.randn(n, 100, device="cuda")
res = {}
ms = bench._time(lambda: out.fill_(1.0), steps=20, warmup=5)
res["fill_2.46GB"] = {"ms": ms, "write_TBps": out.numel() * 4 / ms / 1e9}
ms = bench._time(lambda: out.copy_(src), steps=20, warmup=5)
res["copy_2.46GB"] = {"ms": ms, "read_plus_write_TBps": 2 * out.numel() * 4 / ms / 1e9}
ms = bench._time(lambda: src.sum(), steps=20, warmup=5)
res["sum_2.46GB"] = {"ms": ms, "read_TBps": src.numel() * 4 / ms / 1e9}
for N in (64, 128, 256):
    b = torch.randn(100, N, device="cuda")
    c = torch.empty(n, N, device="cuda")
    ms = bench._time(lambda: gemm_bias_act(a, b, out=c), steps=20, warmup=5)
    mt = bench._time(lambda: torch.matmul(a, b, out=c), steps=20, warmup=5)
    byt = 4 * n * (100 + N)
    res["gemm_100_to_{}".format(N)] = {"ms": ms, "TFLOPs": 2.0 * n * 100 * N / ms / 1e9, "bytes_TBps": byt / ms / 1e9,
                                       "write_GB": 4 * n * N / 1e9, "hipblaslt_ms": mt}
print(json.dumps(res, indent=1))
