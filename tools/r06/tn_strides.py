# coding=utf-8
"""red^T dh at the max-pool layer's shapes: dense operands vs the strides the layer hands over (red rows 544 floats apart, dh a column
half of a [N, 256] gradient)."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from tf_geometric_amd.plan import gemm_tn
from tf_geometric_amd import plan as P
M = 2449029
def t(fn):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); a.record()
        for _ in range(5): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 5)
    return round(min(ts), 3)
res = {}
xd = torch.randn(M, 512, device="cuda"); gd = torch.randn(M, 128, device="cuda")
res["dense_x_dense_g"] = t(lambda: gemm_tn(xd, gd))
xf = P.gather_friendly_empty(M, 512, torch.device("cuda")); xf.normal_()
res["ld544_x_dense_g"] = t(lambda: gemm_tn(xf, gd))
g2 = torch.randn(M, 256, device="cuda")
res["dense_x_half_of_256_g"] = t(lambda: gemm_tn(xd, g2[:, 128:]))
res["ld544_x_half_of_256_g"] = t(lambda: gemm_tn(xf, g2[:, 128:]))
res["ld544_x_half_of_256_g_bias"] = t(lambda: gemm_tn(xf, g2[:, 128:], want_bias=True))
print(json.dumps(res))
