# coding=utf-8
"""One GEMM shape a few times (for rocprofv3 --pmc passes):  python tools/gemm_once.py M K N [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_geometric_amd.plan import gemm_bias_act          # noqa: E402

M, K, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
a = torch.randn(M, K, device="cuda")
b = torch.randn(K, N, device="cuda") * 0.1
out = torch.empty(M, N, device="cuda")
for _ in range(reps):
    gemm_bias_act(a, b, out=out)
torch.cuda.synchronize()
print("done")
