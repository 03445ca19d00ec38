# coding=utf-8
"""Row STRIDE of the gathered table vs gather time at products shape: are line-aligned, power-of-two strides (F = 128: 512 B,
F = 256: 1024 B) slower per 128-byte line than odd strides, and does spreading the rows (stride 640 / 1152 B, still whole
lines per row) recover the random-line ceiling that tools/line_rate_probe.cpp measures on the same box?"""
import json
import sys

import torch

sys.path.insert(0, ".")
from tf_geometric_amd import synthetic, _lib as L                     # noqa: E402
from tf_geometric_amd.plan import CsrPlan, segment_reduce             # noqa: E402

n, e, _ = synthetic.WORKLOADS["products"]
ei = L.as_i32(synthetic.synthetic_edges(n, e, seed=0))
plan = CsrPlan.build(ei, n, n)
w = torch.rand(plan.num_edges, device="cuda")


def t(fn, k=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / k


cases = [(128, [128, 132, 136, 144, 160, 192]), (100, [100, 128, 160]), (64, [64, 96]), (96, [96, 128, 160]),
         (256, [256, 288, 320]), (192, [192, 224, 256])]
if len(sys.argv) > 1:
    cases = [(int(a.split(":")[0]), [int(v) for v in a.split(":")[1].split(",")]) for a in sys.argv[1:]]
for F, lds in cases:
    row = {"F": F}
    for ld in lds:
        buf = torch.randn(n, ld, device="cuda")
        x = buf[:, :F]
        out = torch.empty(n, F, device="cuda")
        ms = t(lambda: segment_reduce(plan, x, L.SUM, w_csr=w, out=out))
        row["ld%d_ms" % ld] = round(ms, 3)
        del buf, x, out
    print(json.dumps(row), flush=True)

# base offset of a line-aligned table: rows of 512 B starting at 0 / 16 / 32 / 64 / 128 bytes past a 512-byte boundary
if len(sys.argv) == 1:
    for F, ld in [(128, 128), (256, 256), (64, 64)]:
        row = {"F": F, "ld": ld, "what": "base offset in floats -> ms"}
        for shift in (0, 4, 8, 16, 32, 64):
            buf = torch.randn(n * ld + 256, device="cuda")
            x = buf[shift:shift + n * ld].view(n, ld)[:, :F]
            out = torch.empty(n, F, device="cuda")
            row["shift%d_ms" % shift] = round(t(lambda: segment_reduce(plan, x, L.SUM, w_csr=w, out=out)), 3)
            del buf, x, out
        print(json.dumps(row), flush=True)
