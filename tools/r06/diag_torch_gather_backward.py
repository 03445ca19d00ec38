# coding=utf-8
"""Is torch's float64 backward of x[idx] (index_put_ accumulate) exact at E = 123 M x 25 columns?  vs index_add."""
import torch
n, e, c = 2400000, 123000000, 25
g = torch.Generator(device="cuda"); g.manual_seed(1)
for e_, c_ in ((e, c), (e, 8), (20000000, 25), (114000000, 8)):
    idx = torch.randint(0, n, (e_,), generator=g, device="cuda")
    x = torch.randn(n, c_, generator=g, device="cuda", dtype=torch.float64).requires_grad_()
    up = torch.randn(e_, c_, generator=g, device="cuda", dtype=torch.float64)
    ref = torch.zeros(n, c_, dtype=torch.float64, device="cuda").index_add(0, idx, up)
    x[idx].backward(up)
    d1 = float((x.grad - ref).abs().max()); x.grad = None
    x.index_select(0, idx).backward(up)
    d2 = float((x.grad - ref).abs().max()); x.grad = None
    xs = torch.randn(n, 100, generator=g, device="cuda", dtype=torch.float64).requires_grad_()
    xs[:, 25:25 + c_][idx].backward(up)
    d3 = float((xs.grad[:, 25:25 + c_] - ref).abs().max())
    print("E={} cols={}: x[idx] backward max|d| {:.3e}; index_select backward {:.3e}; sliced x[:, a:b][idx] {:.3e}".format(e_, c_, d1, d2, d3))
    del idx, x, up, ref, xs
